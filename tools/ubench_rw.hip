// Micro-benchmark (design aid): the STFT kernel's memory traffic without its compute.
// Per frame a wave loads samples and stores one 513-float row.
//   MODE 0: load the frame's 4 KB (8 x dwordx2, as kernel v4), store 9 dword stores
//   MODE 1: load only the 1 KB of new samples (1 x dwordx4), store 9 dword stores
//   MODE 2: like 0 but stores only        MODE 3: like 0 but loads only
// Frames are assigned like v4: workgroup = contiguous run, wave w takes run + 4i + w.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE, int MAP>
__global__ __launch_bounds__(256) void k(const float *x, float *out, long n_frames, float *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * n_frames / gridDim.x, r1 = (long)(blockIdx.x + 1) * n_frames / gridDim.x;
    float acc = 0.f;
    // MAP 0: contiguous run per workgroup; MAP 1: grid-stride groups of 4 frames;
    // MAP 2: grid-stride, XCD-banded (workgroups b, b+8, .. = one XCD take adjacent groups)
    const long G = gridDim.x;
    long bsw = blockIdx.x;
    if (MAP == 2) bsw = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const long tstart = MAP == 0 ? r0 + wave : bsw * 4 + wave;
    const long tend = MAP == 0 ? r1 : n_frames;
    const long tstep = MAP == 0 ? 4 : 4 * G;
    for (long t = tstart; t < tend; t += tstep) {
        float2 v[8];
        if (MODE == 1) {
            float4 q = reinterpret_cast<const float4 *>(x + t * 256)[lane];
            acc += q.x + q.y + q.z + q.w;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = make_float2(q.x + j, q.y - j);
        } else {
            const float2 *src = reinterpret_cast<const float2 *>(x + t * 256) + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[64 * j];
        }
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x * v[j].y;
        } else {
            float *row = out + t * 513;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                row[lane + 64 * j] = v[j].x + v[j + 4].y;
                row[512 - lane - 64 * j] = v[j].y - v[j + 4].x;
            }
            if (lane == 0) row[256] = v[0].x;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int MODE, int MAP>
void run(const float *x, float *out, float *sink, long n_frames, int per_cu, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, MAP>), dim3(256 * per_cu), dim3(256), 0, 0, x, out, n_frames, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 10);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-52s %d WG/CU  %.2f us\n", name, per_cu, ts[2] * 1e3);
}
int main()
{
    long n_samples = 28800000, n_frames = (n_samples - 1024) / 256 + 1;
    float *x, *out, *sink; hipMalloc(&x, n_samples * 4); hipMalloc(&out, (n_frames * 513 + 16) * 4); hipMalloc(&sink, 4);
    hipMemset(x, 0, n_samples * 4);
    for (int pc : {4, 8}) {
        run<0, 0>(x, out, sink, n_frames, pc, "contiguous runs: load 4 KB/frame + store row");
        run<1, 0>(x, out, sink, n_frames, pc, "contiguous runs: load 1 KB/frame + store row");
        run<0, 1>(x, out, sink, n_frames, pc, "grid-stride:     load 4 KB/frame + store row");
        run<1, 1>(x, out, sink, n_frames, pc, "grid-stride:     load 1 KB/frame + store row");
        run<0, 2>(x, out, sink, n_frames, pc, "grid-stride XCD: load 4 KB/frame + store row");
        run<1, 2>(x, out, sink, n_frames, pc, "grid-stride XCD: load 1 KB/frame + store row");
        run<3, 1>(x, out, sink, n_frames, pc, "grid-stride:     load 4 KB/frame only");
        run<3, 2>(x, out, sink, n_frames, pc, "grid-stride XCD: load 4 KB/frame only");
    }
    return 0;
}
