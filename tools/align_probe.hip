// GPU-box probe: cost of vector loads by width and address alignment (gfx950).  Every lane reads `W` dwords at byte offset
// lane * stride + off from a 256 MB buffer (L2-resident slices re-read many times), nothing else in the loop.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/align_probe tools/align_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2_a1 __attribute__((aligned(1)));
typedef u32x3 u32x3_a1 __attribute__((aligned(1)));
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
template <int W>
__global__ void k_load(const uint8_t *buf, uint32_t *out, int off, int stride, int iters, size_t span)
{
    const size_t lane0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)stride + (size_t)off;
    uint32_t acc = 0;
    size_t p = lane0 % span;
    for (int i = 0; i < iters; i++) {
        if (W == 1) acc += *reinterpret_cast<const uint32_t *>(buf + p);
        if (W == 2) { const u32x2 v = *reinterpret_cast<const u32x2_a1 *>(buf + p); acc += v.x ^ v.y; }
        if (W == 3) { const u32x3 v = *reinterpret_cast<const u32x3_a1 *>(buf + p); acc += v.x ^ v.y ^ v.z; }
        if (W == 4) { const u32x4 v = *reinterpret_cast<const u32x4_a1 *>(buf + p); acc += v.x ^ v.y ^ v.z ^ v.w; }
        p += (size_t)gridDim.x * blockDim.x * (size_t)stride;
        if (p >= span) p -= span;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t span = 64ull << 20;                       // 64 MB: HBM / last-level cache traffic, not L2-resident
    uint8_t *buf; uint32_t *out;
    hipMalloc(&buf, span + 4096); hipMalloc(&out, 4096); hipMemset(buf, 1, span + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, threads = 256, iters = 64;
    printf("width stride off  us  GB/s(useful)\n");
    for (int W = 1; W <= 4; W++)
        for (int stride : {4 * W, 4 * W + 4, 20})
            for (int off : {0, 1, 4, 8}) {
                if (stride < 4 * W) continue;
                auto run = [&]() {
                    if (W == 1) hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(threads), 0, 0, buf, out, off, stride, iters, span);
                    if (W == 2) hipLaunchKernelGGL(k_load<2>, dim3(blocks), dim3(threads), 0, 0, buf, out, off, stride, iters, span);
                    if (W == 3) hipLaunchKernelGGL(k_load<3>, dim3(blocks), dim3(threads), 0, 0, buf, out, off, stride, iters, span);
                    if (W == 4) hipLaunchKernelGGL(k_load<4>, dim3(blocks), dim3(threads), 0, 0, buf, out, off, stride, iters, span);
                };
                run(); hipDeviceSynchronize();
                hipEventRecord(e0); for (int r = 0; r < 5; r++) run(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                const double bytes = (double)blocks * threads * iters * 4.0 * W;
                printf("x%d %3d %2d  %8.1f  %8.1f\n", W, stride, off, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
            }
    return 0;
}
