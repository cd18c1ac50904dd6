// Micro-benchmark (design aid, not product): issue cost of the VALU / cross-lane / LDS
// instructions the STFT kernel is built from, on gfx950, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

enum Op { FMA, PKFMA, ADD, PKADD, PKMUL, MOV, DPP_QUAD, DPP_ROW, CNDMASK, PERM32, PERM16, SQRT, BPERM, DSREAD64, DSWRITE64, DSREAD2_64, MIXFMA_DPP, N_OPS };
static const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mov_b32", "v_mov_dpp quad_perm", "v_mov_dpp row_shr:4", "v_cndmask_b32", "v_permlane32_swap", "v_permlane16_swap", "v_sqrt_f32", "ds_bpermute_b32", "ds_read_b64", "ds_write_b64", "ds_read2_b64", "fma+dpp interleaved"};

template <int OP>
__global__ void k(float *out, long long *cycles, int iters)
{
    __shared__ float lds[4096];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 pb = {b, b}, pc = {c, c};
    int addr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 1024;
    lds[threadIdx.x] = a0;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == FMA) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (OP == PKFMA) {
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        } else if (OP == ADD) {
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (OP == PKADD) {
            asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (OP == PKMUL) {
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (OP == MOV) {
            asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == DPP_QUAD) {
            asm volatile(REP8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == DPP_ROW) {
            asm volatile(REP8("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %1, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %2, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %3, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %4, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %5, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %6, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %7, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == CNDMASK) {
            asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else if (OP == PERM32) {
            asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == PERM16) {
            asm volatile(REP8("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == SQRT) {
            asm volatile(REP8("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == BPERM) {
            int ad = ((threadIdx.x * 7 + 3) & 63) * 4;
            asm volatile(REP8("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ad));
        } else if (OP == DSREAD64) {
            asm volatile(REP8("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(addr));
        } else if (OP == DSWRITE64) {
            asm volatile(REP8("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n ds_write_b64 %8, %3 offset:1536\n ds_write_b64 %8, %4 offset:2048\n ds_write_b64 %8, %5 offset:2560\n ds_write_b64 %8, %6 offset:3072\n ds_write_b64 %8, %7 offset:3584\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(addr));
        } else if (OP == DSREAD2_64) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 q0, q1, q2, q3;
            asm volatile(REP8("ds_read2_b64 %0, %4 offset0:0 offset1:64\n ds_read2_b64 %1, %4 offset0:128 offset1:192\n ds_read2_b64 %2, %4 offset0:32 offset1:96\n ds_read2_b64 %3, %4 offset0:160 offset1:224\n s_waitcnt lgkmcnt(0)\n")
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr));
            a0 += q0.x + q1.x + q2.x + q3.x;
        } else if (OP == MIXFMA_DPP) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fma_f32 %2, %2, %8, %9\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fma_f32 %4, %4, %8, %9\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fma_f32 %6, %6, %8, %9\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    long long t1 = clock64();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
        cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(float *out, long long *cyc, int block)
{
    const int iters = 200, grid = 256;
    int per_iter = (OP == DSREAD2_64) ? 32 : 64;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    int nw = grid * block / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = h[nw / 2];
    int waves_per_simd = block / 256;
    // cycles per instruction per wave, and per SIMD (all co-resident waves interleaved)
    printf("%-24s block %4d (%d waves/SIMD): %.2f cyc/instr/wave  -> %.2f cyc/instr per SIMD\n", names[OP], block,
           waves_per_simd, med / (iters * per_iter), med / (iters * per_iter * (double)waves_per_simd));
}

template <int OP>
void run_all(float *out, long long *cyc)
{
    run<OP>(out, cyc, 256);
    run<OP>(out, cyc, 512);
    run<OP>(out, cyc, 1024);
}

int main()
{
    float *out;
    long long *cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 256 * 16 * sizeof(long long));
    run_all<FMA>(out, cyc);
    run_all<PKFMA>(out, cyc);
    run_all<ADD>(out, cyc);
    run_all<PKADD>(out, cyc);
    run_all<PKMUL>(out, cyc);
    run_all<MOV>(out, cyc);
    run_all<DPP_QUAD>(out, cyc);
    run_all<DPP_ROW>(out, cyc);
    run_all<CNDMASK>(out, cyc);
    run_all<PERM32>(out, cyc);
    run_all<PERM16>(out, cyc);
    run_all<SQRT>(out, cyc);
    run_all<BPERM>(out, cyc);
    run_all<DSREAD64>(out, cyc);
    run_all<DSWRITE64>(out, cyc);
    run_all<DSREAD2_64>(out, cyc);
    run_all<MIXFMA_DPP>(out, cyc);
    return 0;
}
