/* Test infrastructure, not product code: the handful of HIP runtime entry points that
 * audio_analyzer_rs_amd/csrc/aa_comm.cpp calls, answered from host memory, so that the multi-rank branches of
 * aa_scatter_samples / aa_gather_rows / aa_comm_barrier run on a box without a GPU (LD_PRELOADed into the
 * child process of tests/test_comm_stub.py, which touches nothing else of the library).  "Device" pointers
 * are host pointers; there is one device (0) and streams complete at once. */
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
const char *hipGetErrorString(int e) { return e == 0 ? "success" : "error (host stand-in)"; }
int aa_hip_host_current_device(void) { return g_device; }
int aa_hip_host_set_calls(void) { return g_set_calls; }
