/* Test infrastructure, not product code: the handful of HIP runtime entry points that
 * audio_analyzer_rs_amd/csrc/aa_comm.cpp calls, answered from host memory, so that the multi-rank branches of
 * aa_scatter_samples / aa_gather_rows / aa_comm_barrier run on a box without a GPU (LD_PRELOADed into the
 * child process of tests/test_comm_stub.py, which touches nothing else of the library).  "Device" pointers
 * are host pointers; there is one device (0) and streams complete at once.  Streams and events are small integers
 * in pointer clothing; every hipEventRecord / hipStreamWaitEvent is logged (kind, stream, event) so that the test
 * can check how aa_sharded_pipeline wires its three streams together. */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static int g_device, g_set_calls;
int hipGetDevice(int *d) { *d = g_device; return 0; }
int hipSetDevice(int d)
{
    ++g_set_calls;
    if (d < 0 || d > 7)
        return 101;   /* hipErrorInvalidDevice */
    g_device = d;
    return 0;
}
int hipGetDeviceCount(int *n) { *n = 8; return 0; }
int hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
int hipFree(void *p) { free(p); return 0; }
int hipMemset(void *p, int v, size_t n) { memset(p, v, n); return 0; }
int hipMemcpy(void *d, const void *s, size_t n, int kind) { (void)kind; memmove(d, s, n); return 0; }
int hipMemcpyAsync(void *d, const void *s, size_t n, int kind, void *st) { (void)kind; (void)st; memmove(d, s, n); return 0; }
int hipStreamSynchronize(void *st) { (void)st; return 0; }
int hipDeviceSynchronize(void) { return 0; }
const char *hipGetErrorString(int e) { return e == 0 ? "success" : "error (host stand-in)"; }

/* streams / events of aa_sharded_pipeline */
enum { kMaxSync = 1 << 16 };
static struct { int kind; size_t stream, event; } g_sync[kMaxSync];     /* kind 0 = record, 1 = wait */
static int g_sync_n;
static size_t g_next_stream = 100, g_next_event = 1000;
int hipStreamCreateWithFlags(void **st, unsigned flags) { (void)flags; *st = (void *)(g_next_stream++); return 0; }
int hipStreamDestroy(void *st) { (void)st; return 0; }
int hipEventCreateWithFlags(void **ev, unsigned flags) { (void)flags; *ev = (void *)(g_next_event++); return 0; }
int hipEventDestroy(void *ev) { (void)ev; return 0; }
static int sync_log(int kind, void *st, void *ev)
{
    if (g_sync_n < kMaxSync) {
        g_sync[g_sync_n].kind = kind;
        g_sync[g_sync_n].stream = (size_t)st;
        g_sync[g_sync_n].event = (size_t)ev;
        ++g_sync_n;
    }
    return 0;
}
int hipEventRecord(void *ev, void *st) { return sync_log(0, st, ev); }
int hipStreamWaitEvent(void *st, void *ev, unsigned flags) { (void)flags; return sync_log(1, st, ev); }
int aa_hip_host_sync_count(void) { return g_sync_n; }
void aa_hip_host_sync_reset(void) { g_sync_n = 0; }
int aa_hip_host_sync_get(int i, int *kind, size_t *stream, size_t *event)
{
    if (i < 0 || i >= g_sync_n)
        return 1;
    *kind = g_sync[i].kind;
    *stream = g_sync[i].stream;
    *event = g_sync[i].event;
    return 0;
}
int aa_hip_host_current_device(void) { return g_device; }
int aa_hip_host_set_calls(void) { return g_set_calls; }
