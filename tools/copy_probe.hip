// GPU-box probe: practical HBM rates on gfx950 for the traffic shapes of the ORB kernels -- read-only stream, copy (1 read : 1 write),
// and the pyramid's 1.44 reads : 1 write -- over spans far beyond the 256 MB last-level cache.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_read(const uint4 *a, uint32_t *out, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_copy(const uint4 *a, uint4 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_write(uint4 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_uint4(1, 2, 3, 4);
}
int main()
{
    const size_t bytes = 2ull << 30, n = bytes / 16;
    uint4 *a, *b; uint32_t *out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 64); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto fn, double moved) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; r++) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, moved / (ms * 1e-3) / 1e9);
    };
    for (int wg : {2048, 8192}) {
        printf("grid %d x 256\n", wg);
        timeit("read 2 GB", [&] { hipLaunchKernelGGL(k_read, dim3(wg), dim3(256), 0, 0, a, out, n); }, (double)bytes);
        timeit("write 2 GB", [&] { hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, b, n); }, (double)bytes);
        timeit("copy 2 GB -> 2 GB", [&] { hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    }
    return 0;
}
