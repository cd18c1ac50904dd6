// Design aid: the shader clock a lightly loaded chip runs at.  One workgroup of 320 threads spins on dependent f32 FMAs for
// ~50 ms; s_memtime (shader cycles, MI355X_MICROARCH.md) against wall_clock64 (100 MHz) gives the frequency, and the
// cycle count per dependent FMA the latency of a lone wave's VALU chain.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_lone_wave_clock tools/ubench_lone_wave_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float *out, long long *t, int iters)
{
    float a = 1.0f + threadIdx.x * 1e-7f, b = 0.9999999f, c = 1e-9f;
    const long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j)
            a = __builtin_fmaf(a, b, c);
    }
    const long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) {
        t[0] = w1 - w0;
        t[1] = c1 - c0;
    }
}
int main()
{
    float *o;
    long long *t, h[2];
    hipMalloc(&o, 4096);
    hipMalloc(&t, 16);
    for (int blocks : {1, 1, 256, 2048}) {
        const int iters = 200000;
        spin<<<blocks, 320>>>(o, t, iters);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double sec = h[0] / 1e8;
        printf("%4d workgroup(s) of 5 waves: %.1f ms, s_memtime %.0f MHz, %.2f shader cycles per dependent v_fma_f32\n", blocks,
               sec * 1e3, h[1] / sec / 1e6, (double)h[1] / (64.0 * iters));
    }
    return 0;
}
