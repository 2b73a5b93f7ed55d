// pmc_calibrate.hip -- calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the ORB
// kernels use (MI355X_MICROARCH.md, "HBM": FETCH_SIZE is only calibrated for 16 B/lane streams).
// Streams a buffer far larger than L2 + Infinity Cache once per kernel with 4-byte and 16-byte loads
// per lane; run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE) and compare with the known bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_read4(const uint32_t *p, size_t n, uint32_t *out)
{
    uint32_t s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 0x12345678u) out[0] = s;
}
__global__ void k_read16(const uint4 *p, size_t n, uint32_t *out)
{
    uint32_t s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 0x12345678u) out[0] = s;
}
__global__ void k_write4(uint32_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main()
{
    const size_t bytes = (size_t)2 << 30;   // 2 GiB
    uint32_t *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    k_read4<<<4096, 256>>>(buf, bytes / 4, out);
    k_read16<<<4096, 256>>>((const uint4 *)buf, bytes / 16, out);
    k_write4<<<4096, 256>>>(buf, bytes / 4);
    hipDeviceSynchronize();
    printf("known bytes per kernel: %zu (KB: %zu)\n", bytes, bytes / 1024);
    return 0;
}
