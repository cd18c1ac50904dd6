// Micro-benchmark (design aid): how fast can waves write 513-float rows (2052 B, rows only
// 4-byte aligned) with (A) 9 dword stores, (B) 2 unaligned dwordx4 + 1 dword, (C) aligned
// dwordx4 body + 1 masked dword head/tail store.  Each wave writes rows r, r+W, ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE>
__global__ void k(float *out, long rows)
{
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * (blockDim.x >> 6);
    float4 v = make_float4(lane, lane + 1, lane + 2, lane + 3);
    for (long r = wid; r < rows; r += nw) {
        float *row = out + r * 513;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                row[lane + 64 * j] = v.x;
                row[512 - lane - 64 * j] = v.y;
            }
            if (lane == 0) row[256] = v.z;
        } else if (MODE == 1) {
            *reinterpret_cast<float4 *>(row + 4 * lane) = v;          // 4-byte aligned only
            *reinterpret_cast<float4 *>(row + 257 + 4 * lane) = v;
            if (lane == 0) row[256] = v.z;
        } else {
            // aligned window of the row: global float index g0 = r*513 - (r&3)
            const int sh = (int)(r & 3);
            float *base = row - sh;                                   // 16-byte aligned
            *reinterpret_cast<float4 *>(base + 4 + 4 * lane) = v;     // i = 4..259
            if (lane < 63) *reinterpret_cast<float4 *>(base + 260 + 4 * lane) = v;   // i = 260..511
            // head i = sh..3 (lanes 0..3), tail i = 512..512+sh (lanes 4..7)
            int i = lane < 4 ? lane : 508 + lane;
            bool ok = lane < 4 ? (lane >= sh) : (lane < 8 && lane - 4 <= sh);
            if (ok) base[i] = v.w;
        }
    }
}
template <int MODE>
void run(float *out, long rows, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, out, rows);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 10);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-40s %.2f us  %.2f TB/s\n", name, ts[2] * 1e3, rows * 2052.0 / (ts[2] * 1e-3) / 1e12);
}
int main()
{
    long rows = 112497;
    float *out; hipMalloc(&out, (rows * 513 + 16) * 4);
    run<0>(out, rows, "9 dword stores / row");
    run<1>(out, rows, "2 unaligned dwordx4 + 1 dword / row");
    run<2>(out, rows, "2 aligned dwordx4 + 1 masked dword / row");
    run<0>(out, rows, "9 dword stores / row (again)");
    return 0;
}
