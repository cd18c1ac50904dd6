// Micro-benchmark (design aid, not product): LDS float atomic-add issue cost on gfx950 for the
// fused mel reduction (each power-spectrum bin adds w*P to <= 2 mel accumulators in LDS).
// Patterns: distinct addresses, 4 lanes per address, 16 lanes per address, all lanes one address;
// compared with ds_write_b32 to the same addresses.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_ldsatomic tools/ubench_ldsatomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X

template <int ATOMIC>
__global__ void k(float *out, long long *cycles, int iters, int share)
{
    __shared__ float lds[16 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x)
        lds[i] = 0.f;
    __syncthreads();
    // `share` lanes per address; successive instructions hit different accumulators
    int addr = (wave * 1024 + (lane / share)) * 4;
    float v = 1.0f + lane;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (ATOMIC)
            asm volatile(REP8("ds_add_f32 %0, %1\n ds_add_f32 %0, %1 offset:256\n ds_add_f32 %0, %1 offset:512\n ds_add_f32 %0, %1 offset:768\n")
                         "s_waitcnt lgkmcnt(0)\n" ::"v"(addr), "v"(v) : "memory");
        else
            asm volatile(REP8("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n")
                         "s_waitcnt lgkmcnt(0)\n" ::"v"(addr), "v"(v) : "memory");
    }
    long long t1 = clock64();
    __syncthreads();
    if (lane == 0)
        cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x];
}

int main()
{
    const int iters = 200, per_iter = 32;
    float *out;
    long long *cyc;
    (void)hipMalloc(&out, 1 << 22);
    (void)hipMalloc(&cyc, 1 << 16);
    int shares[] = {1, 2, 4, 16, 64};
    for (int block : {256, 512, 1024}) {            // 1, 2, 4 waves per SIMD (one WG per CU)
        for (int atomic = 0; atomic < 2; ++atomic)
            for (int s : shares) {
                const int wgs = 256;
                for (int rep = 0; rep < 2; ++rep) {
                    if (atomic)
                        hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(block), 0, 0, out, cyc, iters, s);
                    else
                        hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(block), 0, 0, out, cyc, iters, s);
                    (void)hipDeviceSynchronize();
                }
                std::vector<long long> h(wgs * (block / 64));
                (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
                double sum = 0;
                for (auto c : h)
                    sum += c;
                // same unit as tools/ubench_valu.hip (clock64 per instruction)
                double per_wave = sum / h.size() / (iters * per_iter);
                printf("block %4d  %-12s lanes/addr %2d : %.2f clk/instr/wave -> %.2f clk/instr per CU (all %d waves)\n",
                       block, atomic ? "ds_add_f32" : "ds_write_b32", s, per_wave, per_wave / (block / 64), block / 64);
            }
    }
    return 0;
}
