// Micro-benchmark (design aid, never shipped): v_mfma_f32_4x4x1_16b_f32 on gfx950 - the sixteen-block
// 4 x 4 x 1 form the band-blocked mel contraction runs on.
//   1. operand / result lane map, checked with exact integer data and an asymmetric B;
//   2. issue interval with 1, 2, 4 independent accumulators (cycles per instruction, s_memtime);
//   3. the same stream next to a packed-f32 VALU stream in a second wave of the SIMD (do the pipes overlap?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/ubench_mfma4x4 tools/ubench_mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k_layout(const float *a, const float *b, float *d)
{
    const int lane = threadIdx.x;
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + lane], b[64 + lane], acc, 0, 0, 0);
    d[4 * lane + 0] = acc.x;
    d[4 * lane + 1] = acc.y;
    d[4 * lane + 2] = acc.z;
    d[4 * lane + 3] = acc.w;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(float *out, long long *cyc, int iters)
{
    v4f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        acc[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    float a = 1.0f + 1e-6f * threadIdx.x, b = 1.0f - 1e-6f * threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        cyc[0] = t1 - t0;
}

// waves 0..3 of a 512-thread group run the MFMA stream, waves 4..7 (the second residents of the
// four SIMDs) a packed-FMA stream; mode 1: MFMA only, 2: VALU only, 3: both
__global__ __launch_bounds__(512) void k_coexec(float *out, long long *cyc, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    const long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    if (wave < 4) {
        if (mode & 1) {
            v4f acc[2] = {v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}};
            float a = 1.0f + 1e-6f * threadIdx.x, b = 1.0f - 1e-6f * threadIdx.x;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc[1], 0, 0, 0);
                }
            }
            s = acc[0].x + acc[1].y;
        }
    } else {
        if (mode & 2) {
            v2f p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                p[j] = v2f{1.0f + (float)(threadIdx.x + j) * 1e-6f, 1.0f - (float)(threadIdx.x + j) * 1e-6f};
            const v2f m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        p[j] = p[j] * m + c;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                s += p[j].x + p[j].y;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        cyc[wave] = t1 - t0;
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main()
{
    // ---- 1. lane map.  Hypothesis: block = lane / 4; A: i = lane % 4; B: j = lane % 4; D: register = i, lane % 4 = j
    std::vector<float> a(128), b(128), d(256);
    for (int k = 0; k < 2; ++k)
        for (int l = 0; l < 64; ++l) {
            a[64 * k + l] = (float)(1 + (l % 4) + 5 * (l / 4) + 100 * k);        // A_blk[i](k)
            b[64 * k + l] = (float)(1 + 2 * (l % 4) * (l % 4) + 3 * (l / 4) + 7 * k);   // B_blk[j](k), asymmetric in j
        }
    float *da, *db, *dd;
    long long *dc;
    CK(hipMalloc(&da, 512));
    CK(hipMalloc(&db, 512));
    CK(hipMalloc(&dd, 1 << 22));
    CK(hipMalloc(&dc, 64));
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(da, db, dd);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4, i = r;
            float want = 0.0f;
            for (int k = 0; k < 2; ++k)
                want += a[64 * k + 4 * blk + i] * b[64 * k + 4 * blk + j];
            if (d[4 * l + r] != want) {
                if (bad < 8)
                    printf("lane %d reg %d: got %g want %g\n", l, r, d[4 * l + r], want);
                ++bad;
            }
        }
    printf("LAYOUT 4x4x1_16b: block = lane/4, A i = lane%%4, B j = lane%%4, D[reg = i][lane%%4 = j]: %s (%d mismatches)\n",
           bad ? "WRONG" : "confirmed", bad);

    // ---- 2. issue interval
    const int iters = 2000;
    long long c;
    k_rate<1><<<1, 64>>>(dd, dc, iters);
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("RATE 1 accumulator (dependent chain), 1 wave: %.2f cycles / mfma\n", (double)c / (8.0 * iters));
    k_rate<2><<<1, 64>>>(dd, dc, iters);
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("RATE 2 accumulators, 1 wave: %.2f cycles / mfma\n", (double)c / (16.0 * iters));
    k_rate<4><<<1, 64>>>(dd, dc, iters);
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("RATE 4 accumulators, 1 wave: %.2f cycles / mfma\n", (double)c / (32.0 * iters));
    k_rate<2><<<1, 256>>>(dd, dc, iters);
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("RATE 2 accumulators, 4 waves (one per SIMD): %.2f cycles / mfma\n", (double)c / (16.0 * iters));
    k_rate<2><<<1, 512>>>(dd, dc, iters);
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("RATE 2 accumulators, 8 waves (two per SIMD): %.2f cycles / mfma per wave\n", (double)c / (16.0 * iters));

    // ---- 3. co-execution with a packed-FMA stream in the SIMD's second wave
    long long cw[8];
    for (int mode = 1; mode <= 3; ++mode) {
        k_coexec<<<1, 512>>>(dd, dc, iters, mode);
        CK(hipMemcpy(cw, dc, 64, hipMemcpyDeviceToHost));
        printf("COEXEC mode %d (1 mfma only, 2 valu only, 3 both): mfma wave %lld cycles (%.2f / mfma), valu wave %lld cycles (%.2f / v_pk_fma)\n",
               mode, cw[0], (double)cw[0] / (32.0 * iters), cw[4], (double)cw[4] / (32.0 * iters));
    }
    CK(hipDeviceSynchronize());
    return 0;
}
