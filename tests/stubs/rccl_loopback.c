/* Test infrastructure, not product code: an in-process loopback stand-in for the slice of librccl that
 * audio_analyzer_rs_amd/csrc/aa_comm.cpp binds (dlopen through AA_RCCL_PATH).  Every "rank" of a world lives in
 * ONE process; ncclSend / ncclRecv are logged (op, rank, peer, pointer, count, inside a group or not) and
 * matched first-in first-out per (source rank, destination rank); a matched pair is carried out with
 * hipMemcpy (resolved at run time: the real runtime on a GPU box, tests/stubs/hip_host.c on a CPU box).
 * The point-to-point semantics are NCCL's: counts of a matched pair must agree, a pair matches whenever
 * both sides have been posted (sends and receives are asynchronous with respect to the host).  The log lets
 * tests/test_comm_stub.py check that the lists aa_scatter_samples / aa_gather_rows / aa_comm_barrier issue
 * for worlds of 2, 3 and 8 ranks pair up and carry the ranges of aa_shard_frames / aa_shard_samples. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm { int world, rank; };
typedef struct ncclComm *ncclComm_t;

enum { kMaxLog = 1 << 16, kMaxRanks = 64, kMaxPending = 4096 };
typedef struct { int op, rank, peer, in_group, device; uintptr_t ptr; size_t count; uintptr_t stream; } LogEntry;   /* op 0 = send, 1 = recv */
static LogEntry g_log[kMaxLog];
static int g_log_n, g_errors, g_group_depth, g_next_id = 1, g_group_open_calls, g_group_close_calls;
typedef struct { void *ptr; size_t count; } Pending;
static Pending g_send[kMaxRanks][kMaxRanks][64], g_recv[kMaxRanks][kMaxRanks][64];
static int g_send_n[kMaxRanks][kMaxRanks], g_recv_n[kMaxRanks][kMaxRanks];

static int copy_bytes(void *dst, const void *src, size_t bytes)
{
    typedef int (*memcpy_fn)(void *, const void *, size_t, int);
    static memcpy_fn fn;
    if (!fn)
        fn = (memcpy_fn)dlsym(RTLD_DEFAULT, "hipMemcpy");
    if (!fn) {
        memcpy(dst, src, bytes);
        return 0;
    }
    return fn(dst, src, bytes, 4 /* hipMemcpyDefault */);
}

static void match(int src, int dst)
{
    while (g_send_n[src][dst] > 0 && g_recv_n[src][dst] > 0) {
        Pending s = g_send[src][dst][0], r = g_recv[src][dst][0];
        memmove(&g_send[src][dst][0], &g_send[src][dst][1], sizeof(Pending) * (size_t)--g_send_n[src][dst]);
        memmove(&g_recv[src][dst][0], &g_recv[src][dst][1], sizeof(Pending) * (size_t)--g_recv_n[src][dst]);
        if (s.count != r.count) {
            ++g_errors;                      /* NCCL would hang or corrupt: a test failure here */
            continue;
        }
        if (s.count && copy_bytes(r.ptr, s.ptr, s.count * 4) != 0)
            ++g_errors;
    }
}

static void log_op(int op, int rank, int peer, const void *ptr, size_t count, void *stream)
{
    typedef int (*getdev_fn)(int *);
    static getdev_fn getdev;
    int dev = -1;
    if (!getdev)
        getdev = (getdev_fn)dlsym(RTLD_DEFAULT, "hipGetDevice");
    if (getdev)
        (void)getdev(&dev);                  /* the device that is current while the call is made */
    if (g_log_n < kMaxLog) {
        LogEntry e = {op, rank, peer, g_group_depth > 0, dev, (uintptr_t)ptr, count, (uintptr_t)stream};
        g_log[g_log_n++] = e;
    }
}

int ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    memcpy(id->internal, "aa-loopback", 11);
    id->internal[12] = (char)g_next_id++;
    return 0;
}
int ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank)
{
    (void)id;
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
        return 4;   /* ncclInvalidArgument */
    *comm = (ncclComm_t)malloc(sizeof **comm);
    (*comm)->world = world;
    (*comm)->rank = rank;
    return 0;
}
int ncclCommDestroy(ncclComm_t comm)
{
    free(comm);
    return 0;
}
int ncclGroupStart(void)
{
    ++g_group_depth;
    ++g_group_open_calls;
    return 0;
}
int ncclGroupEnd(void)
{
    ++g_group_close_calls;
    if (g_group_depth <= 0) {
        ++g_errors;
        return 5;   /* ncclInvalidUsage */
    }
    --g_group_depth;
    return 0;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t comm, void *stream)
{
    if (dtype != 7 || peer < 0 || peer >= comm->world || g_send_n[comm->rank][peer] >= 64) {
        ++g_errors;
        return 4;
    }
    log_op(0, comm->rank, peer, buf, count, stream);
    Pending p = {(void *)buf, count};
    g_send[comm->rank][peer][g_send_n[comm->rank][peer]++] = p;
    match(comm->rank, peer);
    return 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t comm, void *stream)
{
    if (dtype != 7 || peer < 0 || peer >= comm->world || g_recv_n[peer][comm->rank] >= 64) {
        ++g_errors;
        return 4;
    }
    log_op(1, comm->rank, peer, buf, count, stream);
    Pending p = {buf, count};
    g_recv[peer][comm->rank][g_recv_n[peer][comm->rank]++] = p;
    match(peer, comm->rank);
    return 0;
}
const char *ncclGetErrorString(int r)
{
    return r == 0 ? "no error" : r == 4 ? "invalid argument (loopback stub)" : "invalid usage (loopback stub)";
}

/* ---- inspection (tests only) ---- */
int aa_stub_log_count(void) { return g_log_n; }
int aa_stub_log_get(int i, int *op, int *rank, int *peer, int *in_group, int *device, size_t *ptr, size_t *count)
{
    if (i < 0 || i >= g_log_n)
        return -1;
    *op = g_log[i].op; *rank = g_log[i].rank; *peer = g_log[i].peer; *in_group = g_log[i].in_group;
    *device = g_log[i].device;
    *ptr = g_log[i].ptr; *count = g_log[i].count;
    return 0;
}
size_t aa_stub_log_stream(int i) { return i < 0 || i >= g_log_n ? 0 : (size_t)g_log[i].stream; }
int aa_stub_pending(void)
{
    int n = 0;
    for (int a = 0; a < kMaxRanks; ++a)
        for (int b = 0; b < kMaxRanks; ++b)
            n += g_send_n[a][b] + g_recv_n[a][b];
    return n;
}
int aa_stub_errors(void) { return g_errors; }
int aa_stub_group_depth(void) { return g_group_depth; }
int aa_stub_group_calls(int *opened, int *closed)
{
    *opened = g_group_open_calls;
    *closed = g_group_close_calls;
    return 0;
}
void aa_stub_reset(void)
{
    g_log_n = g_errors = g_group_depth = g_group_open_calls = g_group_close_calls = 0;
    memset(g_send_n, 0, sizeof g_send_n);
    memset(g_recv_n, 0, sizeof g_recv_n);
}
