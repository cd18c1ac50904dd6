// Microbenchmark (diagnostic, not shipped): cost of a device-scope ticket counter used as a dynamic
// work queue by 1024 persistent workgroups - one counter, 8 (blockIdx % 8) or 256 (blockIdx % 256).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_ticket tools/ubench_ticket.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void ticket_kernel(unsigned *ctr, int n_queues, int chunks, unsigned *sink, int work_iters)
{
    const int q = blockIdx.x % n_queues;
    // queue q owns chunks q, q + n_queues, ...
    const int mine = (chunks - q + n_queues - 1) / n_queues;
    unsigned acc = 0;
    __shared__ unsigned s_t;
    for (;;) {
        if (threadIdx.x == 0)
            s_t = __hip_atomic_fetch_add(&ctr[q * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if ((int)t >= mine)
            break;
        float v = (float)t;
        for (int i = 0; i < work_iters; ++i)
            v = v * 1.0001f + 0.5f;
        acc += (unsigned)v;
    }
    if (threadIdx.x == 0)
        sink[blockIdx.x] = acc;
}

int main()
{
    unsigned *ctr, *sink;
    hipMalloc(&ctr, 256 * 32 * 4);
    hipMalloc(&sink, 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int chunks = 14063;
    for (int work : {0, 2000})
        for (int nq : {1, 8, 256}) {
            std::vector<float> ts;
            for (int rep = 0; rep < 12; ++rep) {
                hipMemsetAsync(ctr, 0, 256 * 32 * 4, 0);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(ticket_kernel, dim3(1024), dim3(256), 0, 0, ctr, nq, chunks, sink, work);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            printf("work %4d  queues %3d : median %.1f us  min %.1f us  (%d tickets + %d misses)\n", work, nq,
                   ts[ts.size() / 2], ts[0], chunks, 1024);
        }
    return 0;
}
