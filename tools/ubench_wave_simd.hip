// Design aid: which SIMD each wave of a 5-wave workgroup lands on (HW_ID bits 5:4), for a few launches.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_wave_simd tools/ubench_wave_simd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out)
{
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 8 + wave] = (int)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID[15:0]
}
int main()
{
    int *d, h[64];
    (void)hipMalloc(&d, sizeof h);
    for (int blocks : {1, 4}) {
        for (int rep = 0; rep < 3; ++rep) {
            probe<<<blocks, 320>>>(d);
            (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            for (int b = 0; b < blocks; ++b) {
                printf("launch of %d, workgroup %d:", blocks, b);
                for (int w = 0; w < 5; ++w)
                    printf("  wave %d: wave_id %d simd %d cu %d", w, h[b * 8 + w] & 15, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15);
                printf("\n");
            }
        }
    }
    return 0;
}
