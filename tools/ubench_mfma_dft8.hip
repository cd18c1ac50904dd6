// Micro-benchmark (design aid, never shipped): ONE radix-8 pass of one frame - 64 columns of an
// 8-point complex DFT, D[16 x 64] = F[16 x 16] X[16 x 64] in real arithmetic - as the 28 packed-f32
// butterfly instructions of pk_dft8 (what the kernels run) against 16 v_mfma_f32_16x16x4_f32 (the
// "pass A as a small f32 MFMA" candidate of the round-1 review), at 4 waves per SIMD, held for a few
// seconds each so that tools/power_mfma_dft8.sh can line rocm-smi power samples up with it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I audio_analyzer_rs_amd/csrc -DAA_ABLATE -o tools/bin/ubench_mfma_dft8 tools/ubench_mfma_dft8.hip
#include "aa_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <ctime>

using aa::v2f;
using aa::v4f;

__global__ __launch_bounds__(256) void k_valu(float *out, int iters)
{
    v2f v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        v[j] = v2f{(float)(threadIdx.x + j) * 1e-3f, (float)(threadIdx.x - j) * 1e-3f};
    const v2f s = {0.35355339f, 0.35355339f};       // keep the values bounded: |DFT8| <= 8 |x|
    for (int i = 0; i < iters; ++i) {
        aa::pk_dft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("" : "+v"(v[j]));
        aa::pk_dft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = v[j] * s * s;                     // (not counted: 8 of 64 instructions per two passes)
    }
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        acc += v[j].x + v[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_mfma(float *out, int iters)
{
    // A operand: F[i][k], lane = (i = lane % 16, k-group = lane / 16): 4 values (K = 16 in 4 steps)
    float a[4], x[16];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        a[s] = 0.25f * __cosf(0.785398f * (float)((lane & 15) * (4 * s + (lane >> 4))));
#pragma unroll
    for (int j = 0; j < 16; ++j)
        x[j] = (float)(threadIdx.x + j) * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        v4f d[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {               // four column blocks of 16 points
            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[4 * b + s], acc, 0, 0, 0);
            d[b] = acc;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            x[4 * b + 0] = d[b].x; x[4 * b + 1] = d[b].y; x[4 * b + 2] = d[b].z; x[4 * b + 3] = d[b].w;
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        acc += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// streams of one packed instruction (8 independent chains, 64 per iteration): energy per instruction
template <int OP>
__global__ __launch_bounds__(256) void k_stream(float *out, int iters)
{
    v2f p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        p[j] = v2f{1.0f + (float)(threadIdx.x + j) * 1e-6f, 1.0f - (float)(threadIdx.x + j) * 1e-6f};
    v2f b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    for (int i = 0; i < iters; ++i) {
#define ONE(INS) asm volatile(INS : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(b), "v"(c))
#define R8(X) X X X X X X X X
        if (OP == 0)
            ONE(R8("v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %9\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n"));
        else if (OP == 1)
            ONE(R8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"));
        else if (OP == 2)
            ONE(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"));
        else if (OP == 3)   // fma whose addend is another live register (as in a scaled butterfly)
            ONE(R8("v_pk_fma_f32 %0, %1, %8, %0\n v_pk_fma_f32 %1, %2, %8, %1\n v_pk_fma_f32 %2, %3, %8, %2\n v_pk_fma_f32 %3, %4, %8, %3\n v_pk_fma_f32 %4, %5, %8, %4\n v_pk_fma_f32 %5, %6, %8, %5\n v_pk_fma_f32 %6, %7, %8, %6\n v_pk_fma_f32 %7, %0, %8, %7\n"));
        else
            ONE(R8("v_pk_add_f32 %0, %1, %0\n v_pk_add_f32 %1, %2, %1\n v_pk_add_f32 %2, %3, %2\n v_pk_add_f32 %3, %4, %3\n v_pk_add_f32 %4, %5, %4\n v_pk_add_f32 %5, %6, %5\n v_pk_add_f32 %6, %7, %6\n v_pk_add_f32 %7, %0, %7\n"));
#pragma unroll
        for (int j = 0; j < 8; ++j)                 // keep the values near 1 (not counted: 8 of 72)
            p[j] = p[j] * v2f{0.5f, 0.5f} + v2f{0.5f, 0.5f};
    }
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        acc += p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K>
static void sustain(K kern, const char *name, float *out, int iters, int passes_per_iter, double seconds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const double t_begin = ts.tv_sec + 1e-9 * ts.tv_nsec;
    double now = t_begin, last_ms = 0;
    while (now - t_begin < seconds) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        last_ms = ms / 20;
        clock_gettime(CLOCK_REALTIME, &ts);
        now = ts.tv_sec + 1e-9 * ts.tv_nsec;
    }
    // passes per launch: 4096 waves x iters x passes_per_iter
    const double passes = 4096.0 * iters * passes_per_iter;
    printf("SUSTAIN %-44s begin %.2f end %.2f  last %.2f us/launch  %.3f ns/pass/wave-slot  passes %.4g\n", name, t_begin, now,
           last_ms * 1e3, last_ms * 1e6 / (iters * passes_per_iter), passes);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double sec = argc > 1 ? atof(argv[1]) : 4.0;
    float *out;
    (void)hipMalloc(&out, 1024 * 256 * 4);
    sustain(k_valu, "pk_dft8 (28 packed f32 instructions)", out, 2000, 2, sec);
    sustain(k_mfma, "16 x v_mfma_f32_16x16x4_f32", out, 2000, 1, sec);
    sustain(k_valu, "pk_dft8 (again)", out, 2000, 2, sec);
    // "pass" = 64 instructions here
    sustain(k_stream<0>, "64 x v_pk_add_f32 (reg + const)", out, 2000, 1, sec);
    sustain(k_stream<4>, "64 x v_pk_add_f32 (reg + reg)", out, 2000, 1, sec);
    sustain(k_stream<1>, "64 x v_pk_mul_f32", out, 2000, 1, sec);
    sustain(k_stream<2>, "64 x v_pk_fma_f32 (reg, const, const)", out, 2000, 1, sec);
    sustain(k_stream<3>, "64 x v_pk_fma_f32 (reg, const, reg)", out, 2000, 1, sec);
    return 0;
}
