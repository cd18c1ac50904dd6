// Micro-benchmark (design aid for aa_diag_stream_copy_device): a 173 MB -> 173 MB copy and the 115 MB -> 4 x 115 MB mix, in
// variants of grid size, unroll, contiguous runs against grid-stride interleaving, nontemporal against plain accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, int WRITES, bool RUNS, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k(const v4f *__restrict__ src, v4f *__restrict__ dst, long n4)
{
    long i, i1, step;
    if (RUNS) {
        const long per = (n4 + gridDim.x - 1) / gridDim.x;
        i = (long)blockIdx.x * per + threadIdx.x;
        i1 = (long)blockIdx.x * per + per < n4 ? (long)blockIdx.x * per + per : n4;
        step = 256;
    } else {
        i = (long)blockIdx.x * 256 + threadIdx.x;
        i1 = n4;
        step = (long)gridDim.x * 256;
    }
    for (; i + (U - 1) * step < i1; i += U * step) {
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            r[u] = NTL ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w = 0; w < WRITES; ++w) {
                if (NTS)
                    __builtin_nontemporal_store(r[u], dst + w * n4 + i + u * step);
                else
                    dst[w * n4 + i + u * step] = r[u];
            }
    }
    for (; i < i1; i += step) {
        v4f r = src[i];
#pragma unroll
        for (int w = 0; w < WRITES; ++w)
            dst[w * n4 + i] = r;
    }
}
template <int U, int WRITES, bool RUNS, bool NTL, bool NTS>
void run(const v4f *a, v4f *b, long n4, int grid, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<U, WRITES, RUNS, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, n4);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 20);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-58s grid %5d  %7.2f us  %.2f TB/s\n", name, grid, ts[3] * 1e3, (1.0 + WRITES) * n4 * 16 / (ts[3] * 1e-3) / 1e12);
}
int main()
{
    const long n_copy = 173022720 / 16, n_mix = 115200000 / 16;
    v4f *a, *b; hipMalloc(&a, n_copy * 16); hipMalloc(&b, n_mix * 16 * 4 + 64); hipMemset(a, 0x3c, n_copy * 16);
    for (int warm = 0; warm < 200; ++warm) hipLaunchKernelGGL((k<4, 1, true, true, true>), dim3(2048), dim3(256), 0, 0, a, b, n_copy);
    for (int g : {1024, 2048, 4096, 8192}) {
        run<4, 1, true, true, true>(a, b, n_copy, g, "copy  runs   U4 nt/nt");
        run<8, 1, true, true, true>(a, b, n_copy, g, "copy  runs   U8 nt/nt");
        run<4, 1, false, true, true>(a, b, n_copy, g, "copy  stride U4 nt/nt");
        run<8, 1, false, true, true>(a, b, n_copy, g, "copy  stride U8 nt/nt");
        run<4, 1, true, false, true>(a, b, n_copy, g, "copy  runs   U4 plain loads / nt stores");
        run<4, 1, true, false, false>(a, b, n_copy, g, "copy  runs   U4 plain / plain");
        run<4, 4, true, true, true>(a, b, n_mix, g, "1r4w  runs   U4 nt/nt");
        run<4, 4, false, true, true>(a, b, n_mix, g, "1r4w  stride U4 nt/nt");
        run<2, 4, true, true, true>(a, b, n_mix, g, "1r4w  runs   U2 nt/nt");
    }
    return 0;
}
