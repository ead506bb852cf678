// Developer microbenchmark: effective shader clock and per-op costs for low-occupancy short kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f; // dependent fp32 chain
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = r1 - r0; out[3 * blockIdx.x + 2] = (unsigned long long)a; }
}
__global__ void k_probe64(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double a = threadIdx.x * 1e-3, b = 1.0001;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 2] = (unsigned long long)a; }
}
__global__ void k_lds(unsigned long long *out, int iters)
{
    __shared__ int s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int p = threadIdx.x;
    for (int i = 0; i < iters; ++i) p = s[p]; // dependent LDS chain
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 2] = p; }
}
__global__ void k_barrier(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[3 * blockIdx.x] = t1 - t0;
}
__global__ void k_empty() {}
int main()
{
    unsigned long long *d; hipMalloc(&d, 3 * 4096 * 8);
    std::vector<unsigned long long> h(3 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int blocks : {1, 64, 256, 2048}) {
            hipEventRecord(e0); k_probe<<<blocks, 256>>>(d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, 24, hipMemcpyDeviceToHost);
            printf("fp32 chain blocks=%4d: %.1f us wall, %llu memtime ticks, %llu realtime(100MHz) -> %.0f MHz, %.2f ticks/iter\n", blocks, ms * 1e3,
                   h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0, h[0] / 20000.0);
        }
    }
    k_probe64<<<1, 256>>>(d, 20000); hipMemcpy(h.data(), d, 24, hipMemcpyDeviceToHost); printf("fp64 mul+add chain: %.2f ticks/iter\n", h[0] / 20000.0);
    k_lds<<<1, 256>>>(d, 20000); hipMemcpy(h.data(), d, 24, hipMemcpyDeviceToHost); printf("dependent LDS read: %.2f ticks/iter\n", h[0] / 20000.0);
    k_lds<<<1, 64>>>(d, 20000); hipMemcpy(h.data(), d, 24, hipMemcpyDeviceToHost); printf("dependent LDS read (1 wave): %.2f ticks/iter\n", h[0] / 20000.0);
    for (int t : {64, 256, 512, 1024}) { k_barrier<<<1, t>>>(d, 20000); hipMemcpy(h.data(), d, 24, hipMemcpyDeviceToHost); printf("__syncthreads %4d threads: %.2f ticks/iter\n", t, h[0] / 20000.0); }
    // back-to-back empty kernels
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 1000; ++i) k_empty<<<1, 64>>>(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("1000 empty launches: %.2f us each\n", ms);
    hipEventRecord(e0); for (int i = 0; i < 1000; ++i) k_empty<<<2048, 256>>>(); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("1000 empty 2048x256 launches: %.2f us each\n", ms);
    return 0;
}
