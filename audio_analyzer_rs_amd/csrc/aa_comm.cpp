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

}  // namespace

struct aa_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
    float *token = nullptr;            // 2 * world floats on the device (aa_comm_barrier)
};

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

int aa_scatter_samples(aa_comm *c, const float *full_dev, int64_t n_samples, int32_t n_fft, int32_t hop,
                       int32_t root, float *local_dev, int64_t local_capacity, void *stream)
{
    if (!c || n_samples < 0 || n_fft <= 0 || hop <= 0 || root < 0 || root >= c->world || local_capacity < 0)
        return fail(AA_ERR_ARG, "aa_scatter_samples: bad argument");
    if (c->rank == root && !full_dev && n_samples > 0)
        return fail(AA_ERR_ARG, "aa_scatter_samples: the root needs the whole buffer");
    COMM_DEVICE(c, "aa_scatter_samples");
    const int64_t total = n_samples < n_fft ? 0 : (n_samples - n_fft) / hop + 1;
    int64_t f0, f1, s0, s1;
    frames_of(total, c->world, c->rank, &f0, &f1);
    samples_of(f0, f1, n_fft, hop, &s0, &s1);
    if (s1 - s0 > local_capacity || (s1 > s0 && !local_dev))
        return fail(AA_ERR_LEN, "aa_scatter_samples: rank %d needs %lld samples, local buffer holds %lld",
                    c->rank, (long long)(s1 - s0), (long long)local_capacity);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->rank == root) {
        if (s1 > s0 && local_dev != full_dev + s0) {
            hipError_t e = hipMemcpyAsync(local_dev, full_dev + s0, sizeof(float) * (size_t)(s1 - s0),
                                          hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess)
                return fail(AA_ERR_HIP, "aa_scatter_samples: %s", hipGetErrorString(e));
        }
        GroupScope grp;
        NCCL_TRY(grp.begin());
        for (int r = 0; r < c->world; ++r) {
            if (r == root)
                continue;
            int64_t g0, g1, t0, t1;
            frames_of(total, c->world, r, &g0, &g1);
            samples_of(g0, g1, n_fft, hop, &t0, &t1);
            if (t1 > t0)
                NCCL_TRY(g_rccl.Send(full_dev + t0, (size_t)(t1 - t0), kNcclFloat32, r, c->comm, st));
        }
        NCCL_TRY(grp.end());
    } else if (s1 > s0) {
        NCCL_TRY(g_rccl.Recv(local_dev, (size_t)(s1 - s0), kNcclFloat32, root, c->comm, st));
    }
    return AA_OK;
}

int aa_gather_rows(aa_comm *c, const float *rows_dev, int64_t total_frames, int32_t cols, int32_t root,
                   float *out_dev, void *stream)
{
    if (!c || total_frames < 0 || cols <= 0 || root < -1 || root >= c->world)
        return fail(AA_ERR_ARG, "aa_gather_rows: bad argument");
    COMM_DEVICE(c, "aa_gather_rows");
    int64_t f0, f1;
    frames_of(total_frames, c->world, c->rank, &f0, &f1);
    const bool receiver = root < 0 || root == c->rank;
    if ((f1 > f0 && !rows_dev) || (receiver && total_frames > 0 && !out_dev))
        return fail(AA_ERR_ARG, "aa_gather_rows: NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (receiver && f1 > f0 && rows_dev != out_dev + f0 * cols) {
        hipError_t e = hipMemcpyAsync(out_dev + f0 * cols, rows_dev, sizeof(float) * (size_t)((f1 - f0) * cols),
                                      hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess)
            return fail(AA_ERR_HIP, "aa_gather_rows: %s", hipGetErrorString(e));
    }
    if (c->world == 1)
        return AA_OK;
    GroupScope grp;
    NCCL_TRY(grp.begin());
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank)
            continue;
        if (f1 > f0 && (root < 0 || root == r))              // my block goes to r
            NCCL_TRY(g_rccl.Send(rows_dev, (size_t)((f1 - f0) * cols), kNcclFloat32, r, c->comm, st));
        if (receiver) {                                      // r's block comes to me
            int64_t g0, g1;
            frames_of(total_frames, c->world, r, &g0, &g1);
            if (g1 > g0)
                NCCL_TRY(g_rccl.Recv(out_dev + g0 * cols, (size_t)((g1 - g0) * cols), kNcclFloat32, r, c->comm, st));
        }
    }
    NCCL_TRY(grp.end());
    return AA_OK;
}

}  // extern "C"
