// Micro-benchmark (design aid): 1 KB load + 2052 B row store per frame per wave, with
// software prefetch depth D (loads issued D frames ahead) and store width variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int D, int WIDE>
__global__ __launch_bounds__(256) void k(const float *x, float *out, long n_frames, float *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * n_frames / gridDim.x, r1 = (long)(blockIdx.x + 1) * n_frames / gridDim.x;
    float4 q[D + 1];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        long tt = r0 + wave + 4 * d;
        q[d] = tt < r1 ? reinterpret_cast<const float4 *>(x + tt * 256)[lane] : make_float4(0, 0, 0, 0);
    }
    for (long t = r0 + wave; t < r1; t += 4) {
        long tp = t + 4 * D;
        q[D] = tp < r1 ? reinterpret_cast<const float4 *>(x + tp * 256)[lane] : make_float4(0, 0, 0, 0);
        float4 c = q[0];
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] = q[d + 1];
        float *row = out + t * 513;
        if (WIDE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                row[lane + 64 * j] = c.x + j;
                row[512 - lane - 64 * j] = c.y - j;
            }
            if (lane == 0) row[256] = c.z;
        } else if (WIDE == 1) {
            *reinterpret_cast<float4 *>(row + 4 * lane) = c;
            *reinterpret_cast<float4 *>(row + 257 + 4 * lane) = make_float4(c.y, c.x, c.w, c.z);
            if (lane == 0) row[256] = c.z;
        } else if (WIDE == 2 || WIDE == 3) {
            // 16-byte aligned window of the row
            const int sh = (int)(t & 3);
            float *base = row - sh;
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 cc = {c.x, c.y, c.z, c.w};
            if (WIDE == 3) {
                __builtin_nontemporal_store(cc, reinterpret_cast<f4 *>(base + 4 + 4 * lane));
                if (lane < 63) __builtin_nontemporal_store(cc, reinterpret_cast<f4 *>(base + 260 + 4 * lane));
            } else {
                *reinterpret_cast<f4 *>(base + 4 + 4 * lane) = cc;
                if (lane < 63) *reinterpret_cast<f4 *>(base + 260 + 4 * lane) = cc;
            }
            int i = lane < 4 ? lane : 508 + lane;
            bool ok = lane < 4 ? (lane >= sh) : (lane < 8 && lane - 4 <= sh);
            if (ok) base[i] = c.w;
        } else {   // WIDE == 4: dword stores, nontemporal
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_nontemporal_store(c.x + j, row + lane + 64 * j);
                __builtin_nontemporal_store(c.y - j, row + 512 - lane - 64 * j);
            }
            if (lane == 0) row[256] = c.z;
        }
    }
}
template <int D, int WIDE>
void run(const float *x, float *out, float *sink, long n_frames, int per_cu, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<D, WIDE>), dim3(256 * per_cu), dim3(256), 0, 0, x, out, n_frames, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 10);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-44s %d WG/CU  %.2f us  (%.2f TB/s)\n", name, per_cu, ts[2] * 1e3, n_frames * 3076.0 / (ts[2] * 1e-3) / 1e12);
}
__global__ void copyk(const float4 *a, float4 *b, long n) { for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i]; }
int main()
{
    long n_samples = 28800000, n_frames = (n_samples - 1024) / 256 + 1;
    float *x, *out, *sink; hipMalloc(&x, n_samples * 4); hipMalloc(&out, (n_frames * 513 + 16) * 4); hipMalloc(&sink, 4);
    hipMemset(x, 0, n_samples * 4);
    for (int pc : {4}) {
        run<1, 0>(x, out, sink, n_frames, pc, "prefetch 1, dword stores");
        run<1, 1>(x, out, sink, n_frames, pc, "prefetch 1, unaligned dwordx4 stores");
        run<1, 2>(x, out, sink, n_frames, pc, "prefetch 1, aligned dwordx4 stores");
        run<1, 3>(x, out, sink, n_frames, pc, "prefetch 1, aligned dwordx4 nt stores");
        run<1, 4>(x, out, sink, n_frames, pc, "prefetch 1, dword nt stores");
        run<1, 0>(x, out, sink, n_frames, pc, "prefetch 1, dword stores (again)");
    }
    // plain float4 copy of 173 MB -> 173 MB for reference
    {
        long n = 173 * 1000 * 1000 / 16; float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 0, n * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copyk, dim3(2048), dim3(256), 0, 0, a, b, n);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("float4 copy 173 MB->173 MB: %.2f us (%.2f TB/s)\n", ms / 10 * 1e3, 2.0 * n * 16 / (ms / 10 * 1e-3) / 1e12);
        }
    }
    return 0;
}
