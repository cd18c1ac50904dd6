// Collective helpers of the sharded job (include/aa_stft.h, "collectives"): RCCL over xGMI, one
// process per GPU.  The frame-range partition itself needs no collective (aa_shard_*); these two
// exchanges exist for hosts whose samples start on one GPU or whose consumer wants every row on
// one GPU: scatter = grouped ncclSend from the root / one ncclRecv per rank,
// gather = one ncclSend per rank / grouped ncclRecv straight into the root's [F, cols] slab —
// unpadded blocks, nothing staged, nothing broadcast.
//
// librccl is bound at run time (dlopen), not at link time: a process that also embeds PyTorch
// already carries a librccl and a libamdhip64 of its own, and a second copy of either in one
// process disagrees with the first about the devices.  The copy already mapped wins; otherwise
// AA_RCCL_PATH, then the loader's search path, then /opt/rocm/lib.
#include "../../include/aa_stft.h"
#include "aa_internal.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#define fail(...) aa::abi_fail(__VA_ARGS__)

namespace {

// The slice of rccl.h this file uses (ABI of NCCL 2.x / RCCL 2.x: stable since 2.7).
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == AA_COMM_ID_BYTES, "aa_comm_id mirrors ncclUniqueId");
typedef int ncclResult_t;          // ncclSuccess == 0
constexpr int kNcclFloat32 = 7;    // ncclDataType_t::ncclFloat32

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    const char *origin = "";
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_error[256] = "";

void load_rccl()
{
    struct Cand {
        const char *path;
        int flags;
        const char *origin;
    };
    const char *env = getenv("AA_RCCL_PATH");
    const Cand cands[] = {
        {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD, "librccl.so.1 already mapped in this process"},
        {"librccl.so", RTLD_NOW | RTLD_NOLOAD, "librccl.so already mapped in this process"},
        {env, RTLD_NOW | RTLD_LOCAL, "AA_RCCL_PATH"},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL, "librccl.so.1 (loader search path)"},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL, "/opt/rocm/lib/librccl.so.1"},
    };
    void *h = nullptr;
    char why[160] = "";            // message of the first real load that failed (RTLD_NOLOAD probes do not count)
    for (const Cand &c : cands) {
        if (!c.path || !*c.path)
            continue;
        h = dlopen(c.path, c.flags);
        if (h) {
            g_rccl.origin = c.origin;
            break;
        }
        const char *msg = dlerror();                         // once: the call clears the error
        if (!(c.flags & RTLD_NOLOAD) && !why[0])
            snprintf(why, sizeof why, "%s: %s", c.origin, msg ? msg : "dlopen failed");
    }
    if (!h) {
        snprintf(g_rccl_error, sizeof g_rccl_error, "librccl not found (%s)", why[0] ? why : "no candidate path");
        return;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) {
            ok = false;
            snprintf(g_rccl_error, sizeof g_rccl_error, "librccl lacks %s", name);
        }
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (ok)
        g_rccl.handle = h;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

int nccl_fail(const char *what, ncclResult_t r)
{
    return fail(AA_ERR_NCCL, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
}

#define NCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != 0)                                                                           \
            return nccl_fail(#expr, r_);                                                       \
    } while (0)

// ncclGroupStart .. ncclGroupEnd with the end guaranteed: an error between the two must not leave
// the group open (every later RCCL call of the thread would queue into it).
struct GroupScope {
    bool open = false;
    ncclResult_t begin()
    {
        ncclResult_t r = g_rccl.GroupStart();
        open = r == 0;
        return r;
    }
    ncclResult_t end()
    {
        open = false;
        return g_rccl.GroupEnd();
    }
    ~GroupScope()
    {
        if (open)
            (void)g_rccl.GroupEnd();
    }
};

// The communicator's device is selected for the duration of every call that takes a comm (aa::DeviceGuard): a host
// thread that never called hipSetDevice sits on device 0, and hipMemcpyAsync / hipStreamSynchronize / ncclSend /
// ncclRecv act on the CURRENT device.
#define COMM_DEVICE(c, what)                                                                   \
    aa::DeviceGuard dev_guard_((c)->device);                                                   \
    if (!dev_guard_.ok)                                                                        \
    return fail(AA_ERR_HIP, what ": cannot select device %d", (c)->device)

void frames_of(int64_t total, int world, int r, int64_t *f0, int64_t *f1)
{
    *f0 = (int64_t)(((__int128)total * r) / world);          // == aa_shard_frames
    *f1 = (int64_t)(((__int128)total * (r + 1)) / world);
}
void samples_of(int64_t f0, int64_t f1, int n_fft, int hop, int64_t *s0, int64_t *s1)
{
    *s0 = f0 * (int64_t)hop;                                 // == aa_shard_samples
    *s1 = (f1 == f0) ? *s0 : (f1 - 1) * (int64_t)hop + n_fft;
}
// Piece k of P of the frame range [f0, f1) (== aa_shard_piece): the same floor rule as the ranks' partition.
void piece_of(int64_t f0, int64_t f1, int P, int k, int64_t *p0, int64_t *p1)
{
    const int64_t n = f1 - f0;
    *p0 = f0 + (int64_t)(((__int128)n * k) / P);
    *p1 = f0 + (int64_t)(((__int128)n * (k + 1)) / P);
}
// The samples piece k ADDS to what pieces 0 .. k-1 already delivered of the range of [f0, f1): consecutive pieces
// share n_fft - hop samples (as consecutive ranks do) and every sample travels once.  [*a, *b) in global sample
// indices; the pieces tile [s0, s1) of samples_of(f0, f1) exactly.
void piece_samples_of(int64_t f0, int64_t f1, int P, int k, int n_fft, int hop, int64_t *a, int64_t *b)
{
    auto end_of = [&](int kk) {            // one past the last sample pieces 0 .. kk need
        int64_t q0, q1;
        piece_of(f0, f1, P, kk, &q0, &q1);
        return q1 > f0 ? (q1 - 1) * (int64_t)hop + n_fft : f0 * (int64_t)hop;
    };
    *a = k == 0 ? f0 * (int64_t)hop : end_of(k - 1);
    *b = end_of(k);
}

}  // namespace

struct aa_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
    float *token = nullptr;            // 2 * world floats on the device (aa_comm_barrier)
    // aa_sharded_pipeline: the stream this communicator's exchanges run on next to the caller's compute stream,
    // and the events that order the three against each other (created on first use, reused by later calls)
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> events;
};

int aa_comm_device_index(const aa_comm *c) { return c ? c->device : -1; }

extern "C" {

const char *aa_comm_backend(void)
{
    return rccl() ? g_rccl.origin : g_rccl_error;
}

int aa_comm_unique_id(aa_comm_id *out_id)
{
    if (!out_id)
        return fail(AA_ERR_ARG, "aa_comm_unique_id: NULL argument");
    const Rccl *n = rccl();
    if (!n)
        return fail(AA_ERR_NCCL, "%s", g_rccl_error);
    ncclUniqueId id;
    NCCL_TRY(n->GetUniqueId(&id));
    memcpy(out_id->bytes, id.internal, AA_COMM_ID_BYTES);
    return AA_OK;
}

int aa_comm_create(const aa_comm_id *id, int32_t world, int32_t rank, aa_comm **out_comm)
{
    if (!id || !out_comm || world <= 0 || rank < 0 || rank >= world)
        return fail(AA_ERR_ARG, "aa_comm_create: bad argument (world %d, rank %d)", world, rank);
    *out_comm = nullptr;
    const Rccl *n = rccl();
    if (!n)
        return fail(AA_ERR_NCCL, "%s", g_rccl_error);
    aa_comm *c = new (std::nothrow) aa_comm();
    if (!c)
        return fail(AA_ERR_NOMEM, "aa_comm_create: out of host memory");
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess)
        e = hipMalloc(&c->token, sizeof(float) * 2 * (size_t)world);
    if (e == hipSuccess)
        e = hipMemset(c->token, 0, sizeof(float) * 2 * (size_t)world);
    if (e != hipSuccess) {
        delete c;
        return fail(AA_ERR_HIP, "aa_comm_create: %s (no usable HIP device)", hipGetErrorString(e));
    }
    ncclUniqueId uid;
    memcpy(uid.internal, id->bytes, AA_COMM_ID_BYTES);
    ncclResult_t r = n->CommInitRank(&c->comm, world, uid, rank);
    if (r != 0) {
        (void)hipFree(c->token);
        delete c;
        return nccl_fail("ncclCommInitRank", r);
    }
    c->world = world;
    c->rank = rank;
    *out_comm = c;
    return AA_OK;
}

int aa_comm_destroy(aa_comm *c)
{
    if (!c)
        return AA_OK;
    int rc = AA_OK;
    aa::DeviceGuard g(c->device);                            // best effort: the handle is released either way
    if (c->comm && g_rccl.CommDestroy) {
        ncclResult_t r = g_rccl.CommDestroy(c->comm);
        if (r != 0)
            rc = nccl_fail("ncclCommDestroy", r);
    }
    (void)hipFree(c->token);
    for (hipEvent_t e : c->events)
        (void)hipEventDestroy(e);
    if (c->side)
        (void)hipStreamDestroy(c->side);
    delete c;
    return rc;
}

int32_t aa_comm_world(const aa_comm *c) { return c ? c->world : 0; }
int32_t aa_comm_rank(const aa_comm *c) { return c ? c->rank : -1; }

int aa_comm_barrier(aa_comm *c, void *stream)
{
    if (!c)
        return fail(AA_ERR_ARG, "aa_comm_barrier: NULL comm");
    COMM_DEVICE(c, "aa_comm_barrier");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GroupScope grp;
    NCCL_TRY(grp.begin());
    for (int r = 0; r < c->world; ++r) {
        NCCL_TRY(g_rccl.Send(c->token + r, 1, kNcclFloat32, r, c->comm, st));
        NCCL_TRY(g_rccl.Recv(c->token + c->world + r, 1, kNcclFloat32, r, c->comm, st));
    }
    NCCL_TRY(grp.end());
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return fail(AA_ERR_HIP, "aa_comm_barrier: %s", hipGetErrorString(e));
    return AA_OK;
}

int aa_shard_piece(int64_t f0, int64_t f1, int32_t n_pieces, int32_t piece, int64_t *p0, int64_t *p1)
{
    if (!p0 || !p1 || f0 < 0 || f1 < f0 || n_pieces <= 0 || piece < 0 || piece >= n_pieces)
        return fail(AA_ERR_ARG, "aa_shard_piece: bad argument (frames %lld..%lld, piece %d of %d)", (long long)f0,
                    (long long)f1, piece, n_pieces);
    piece_of(f0, f1, n_pieces, piece, p0, p1);
    return AA_OK;
}

int aa_scatter_samples_piece(aa_comm *c, const float *full_dev, int64_t n_samples, int32_t n_fft, int32_t hop,
                             int32_t root, float *local_dev, int64_t local_capacity, int32_t n_pieces, int32_t piece,
                             void *stream)
{
    if (!c || n_samples < 0 || n_fft <= 0 || hop <= 0 || root < 0 || root >= c->world || local_capacity < 0 ||
        n_pieces <= 0 || piece < 0 || piece >= n_pieces)
        return fail(AA_ERR_ARG, "aa_scatter_samples_piece: bad argument");
    if (c->rank == root && !full_dev && n_samples > 0)
        return fail(AA_ERR_ARG, "aa_scatter_samples_piece: the root needs the whole buffer");
    COMM_DEVICE(c, "aa_scatter_samples_piece");
    const int64_t total = n_samples < n_fft ? 0 : (n_samples - n_fft) / hop + 1;
    int64_t f0, f1, s0, s1, a, b;
    frames_of(total, c->world, c->rank, &f0, &f1);
    samples_of(f0, f1, n_fft, hop, &s0, &s1);
    if (s1 - s0 > local_capacity || (s1 > s0 && !local_dev))
        return fail(AA_ERR_LEN, "aa_scatter_samples_piece: rank %d needs %lld samples, local buffer holds %lld",
                    c->rank, (long long)(s1 - s0), (long long)local_capacity);
    piece_samples_of(f0, f1, n_pieces, piece, n_fft, hop, &a, &b);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->rank == root) {
        if (b > a && local_dev + (a - s0) != full_dev + a) {
            hipError_t e = hipMemcpyAsync(local_dev + (a - s0), full_dev + a, sizeof(float) * (size_t)(b - a),
                                          hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess)
                return fail(AA_ERR_HIP, "aa_scatter_samples_piece: %s", hipGetErrorString(e));
        }
        GroupScope grp;
        NCCL_TRY(grp.begin());
        for (int r = 0; r < c->world; ++r) {
            if (r == root)
                continue;
            int64_t g0, g1, t0, t1;
            frames_of(total, c->world, r, &g0, &g1);
            piece_samples_of(g0, g1, n_pieces, piece, n_fft, hop, &t0, &t1);
            if (t1 > t0)
                NCCL_TRY(g_rccl.Send(full_dev + t0, (size_t)(t1 - t0), kNcclFloat32, r, c->comm, st));
        }
        NCCL_TRY(grp.end());
    } else if (b > a) {
        NCCL_TRY(g_rccl.Recv(local_dev + (a - s0), (size_t)(b - a), kNcclFloat32, root, c->comm, st));
    }
    return AA_OK;
}

int aa_scatter_samples(aa_comm *c, const float *full_dev, int64_t n_samples, int32_t n_fft, int32_t hop,
                       int32_t root, float *local_dev, int64_t local_capacity, void *stream)
{
    return aa_scatter_samples_piece(c, full_dev, n_samples, n_fft, hop, root, local_dev, local_capacity, 1, 0, stream);
}

int aa_gather_rows_piece(aa_comm *c, const float *rows_dev, int64_t total_frames, int32_t cols, int32_t root,
                         float *out_dev, int32_t n_pieces, int32_t piece, void *stream)
{
    if (!c || total_frames < 0 || cols <= 0 || root < -1 || root >= c->world || n_pieces <= 0 || piece < 0 ||
        piece >= n_pieces)
        return fail(AA_ERR_ARG, "aa_gather_rows_piece: bad argument");
    COMM_DEVICE(c, "aa_gather_rows_piece");
    int64_t f0, f1, p0, p1;
    frames_of(total_frames, c->world, c->rank, &f0, &f1);
    piece_of(f0, f1, n_pieces, piece, &p0, &p1);
    const bool receiver = root < 0 || root == c->rank;
    if ((f1 > f0 && !rows_dev) || (receiver && total_frames > 0 && !out_dev))
        return fail(AA_ERR_ARG, "aa_gather_rows_piece: NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *mine = rows_dev + (p0 - f0) * cols;           // rows_dev holds the rank's whole range
    if (receiver && p1 > p0 && mine != out_dev + p0 * cols) {
        hipError_t e = hipMemcpyAsync(out_dev + p0 * cols, mine, sizeof(float) * (size_t)((p1 - p0) * cols),
                                      hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess)
            return fail(AA_ERR_HIP, "aa_gather_rows_piece: %s", hipGetErrorString(e));
    }
    if (c->world == 1)
        return AA_OK;
    GroupScope grp;
    NCCL_TRY(grp.begin());
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank)
            continue;
        if (p1 > p0 && (root < 0 || root == r))              // my piece goes to r
            NCCL_TRY(g_rccl.Send(mine, (size_t)((p1 - p0) * cols), kNcclFloat32, r, c->comm, st));
        if (receiver) {                                      // r's piece comes to me
            int64_t g0, g1, q0, q1;
            frames_of(total_frames, c->world, r, &g0, &g1);
            piece_of(g0, g1, n_pieces, piece, &q0, &q1);
            if (q1 > q0)
                NCCL_TRY(g_rccl.Recv(out_dev + q0 * cols, (size_t)((q1 - q0) * cols), kNcclFloat32, r, c->comm, st));
        }
    }
    NCCL_TRY(grp.end());
    return AA_OK;
}

int aa_gather_rows(aa_comm *c, const float *rows_dev, int64_t total_frames, int32_t cols, int32_t root,
                   float *out_dev, void *stream)
{
    return aa_gather_rows_piece(c, rows_dev, total_frames, cols, root, out_dev, 1, 0, stream);
}

// ---- scatter k + 1 || compute k || gather k - 1 ---------------------------------------------------------------
namespace {
#define HIP_TRY(expr, what)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(AA_ERR_HIP, what ": %s", hipGetErrorString(e_));                       \
    } while (0)

int side_stream(aa_comm *c)
{
    if (!c->side)
        HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking), "aa_sharded_pipeline");
    return AA_OK;
}
int event_pool(aa_comm *c, size_t n)
{
    while (c->events.size() < n) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), "aa_sharded_pipeline");
        c->events.push_back(e);
    }
    return AA_OK;
}
}  // namespace

int aa_sharded_pipeline(aa_comm *cin, aa_comm *cout, const float *full_dev, int64_t n_samples, int32_t n_fft,
                        int32_t hop, int32_t cols, int32_t root, float *local_dev, int64_t local_capacity,
                        float *rows_dev, float *out_dev, int32_t n_pieces, aa_piece_fn compute, void *user,
                        void *stream)
{
    if (!cin || !cout || !compute || n_pieces <= 0 || n_pieces > 1024 || cols <= 0 || root < 0 || root >= cin->world ||
        n_samples < 0 || n_fft <= 0 || hop <= 0)
        return fail(AA_ERR_ARG, "aa_sharded_pipeline: bad argument");
    if (cin->world != cout->world || cin->rank != cout->rank || cin->device != cout->device)
        return fail(AA_ERR_ARG, "aa_sharded_pipeline: the two communicators must describe the same rank (world %d/%d, "
                    "rank %d/%d, device %d/%d)", cin->world, cout->world, cin->rank, cout->rank, cin->device, cout->device);
    COMM_DEVICE(cin, "aa_sharded_pipeline");
    const int64_t total = n_samples < n_fft ? 0 : (n_samples - n_fft) / hop + 1;
    int64_t f0, f1;
    frames_of(total, cin->world, cin->rank, &f0, &f1);
    int rc = side_stream(cin);
    if (rc == AA_OK && cout != cin)
        rc = side_stream(cout);
    const int P = n_pieces;
    if (rc == AA_OK)
        rc = event_pool(cin, (size_t)(2 * P + 2));
    if (rc != AA_OK)
        return rc;
    hipStream_t st = static_cast<hipStream_t>(stream), sin = cin->side, sout = cout->side;
    hipEvent_t *ev = cin->events.data();        // [0] fork, [1 + k] samples of piece k landed, [1 + P + k] rows of piece k computed,
                                                // [1 + 2 P] last gather issued
    // From here on no early return: every failure goes through the join at the bottom, and a failure of the COMPUTE side
    // (the callback, an event call) does not stop this rank's remaining exchanges - its peers' matching sends / receives
    // are already queued or about to be, and a rank that stopped posting would leave them waiting in RCCL for ever.  The
    // rows of the pieces behind such a failure are undefined; the first error is what the call returns.  A failure of an
    // exchange itself (AA_ERR_NCCL, or HIP inside one) ends the posting: the communicators are then unusable and must
    // be destroyed (include/aa_stft.h says so).
    int rc_first = AA_OK;
    std::string first_msg;
    auto note = [&](int r) {
        if (r != AA_OK && rc_first == AA_OK) {
            rc_first = r;
            first_msg = aa_last_error();
        }
        return r;
    };
    auto hip_ok = [&](hipError_t e) {
        if (e != hipSuccess)
            note(fail(AA_ERR_HIP, "aa_sharded_pipeline: %s", hipGetErrorString(e)));
        return e == hipSuccess;
    };
    // fork: both exchange streams start behind whatever the caller queued on `stream` (the producer of full_dev)
    bool forked = hip_ok(hipEventRecord(ev[0], st)) && hip_ok(hipStreamWaitEvent(sin, ev[0], 0)) &&
                  (sout == sin || hip_ok(hipStreamWaitEvent(sout, ev[0], 0)));
    auto scatter = [&](int k) -> int {
        int r = aa_scatter_samples_piece(cin, full_dev, n_samples, n_fft, hop, root, local_dev, local_capacity, P, k, sin);
        if (r != AA_OK)
            return r;
        hip_ok(hipEventRecord(ev[1 + k], sin));
        return AA_OK;
    };
    // The issue order is the same on every rank - S0, S1, [C0], G0, S2, [C1], G1, .. - which is what RCCL asks of
    // operations on one communicator (cin == cout) and of the two directions of a pair of ranks.
    bool exchange_ok = forked;
    if (exchange_ok)
        exchange_ok = note(scatter(0)) == AA_OK;
    for (int k = 0; k < P && exchange_ok; ++k) {
        if (k + 1 < P && note(scatter(k + 1)) != AA_OK) {
            exchange_ok = false;
            break;
        }
        int64_t p0, p1;
        piece_of(f0, f1, P, k, &p0, &p1);
        hip_ok(hipStreamWaitEvent(st, ev[1 + k], 0));
        if (p1 > p0 && rc_first == AA_OK)
            note(compute(user, p0 - f0, p1 - f0, stream));       // frames relative to the rank's range, as local_dev / rows_dev are
        hip_ok(hipEventRecord(ev[1 + P + k], st));
        hip_ok(hipStreamWaitEvent(sout, ev[1 + P + k], 0));
        if (note(aa_gather_rows_piece(cout, rows_dev, total, cols, root, out_dev, P, k, sout)) != AA_OK)
            exchange_ok = false;                                  // (an exchange failed: stop posting)
    }
    // join: `stream` continues behind the last gather (and, through it, behind everything the pipeline queued) -
    // also on an error path, so that nothing the caller queues next overtakes what was already issued
    (void)hipEventRecord(ev[1 + 2 * P], sout);
    (void)hipStreamWaitEvent(st, ev[1 + 2 * P], 0);
    if (sin != sout) {
        (void)hipEventRecord(ev[0], sin);
        (void)hipStreamWaitEvent(st, ev[0], 0);
    }
    return rc_first == AA_OK ? AA_OK : fail(rc_first, "%s", first_msg.c_str());
}

}  // extern "C"
