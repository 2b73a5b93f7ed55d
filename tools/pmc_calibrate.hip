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
// Row segments, as k_fast_cells / k_blur_mfma / k_orient_desc read them: SEGL lanes read 16 * SEGL consecutive bytes of row r at r * pitch + off;
// every row is read once (the buffer is far larger than L2 + Infinity Cache), so the requested bytes, the 64-byte sectors and the
// 128-byte lines touched are all known.
template <int SEGL>
__global__ void k_rows(const uint8_t *p, size_t pitch, size_t off, size_t nrows, uint32_t *out)
{
    uint32_t s = 0;
    const size_t per = (size_t)gridDim.x * blockDim.x / SEGL;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / SEGL; r < nrows; r += per) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + r * pitch + off + 16 * (threadIdx.x % SEGL));
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 0x12345678u) out[0] = s;
}
template <int SEGL>
static void rows_case(const uint8_t *buf, size_t bytes, size_t pitch, size_t off, uint32_t *out)
{
    const size_t nrows = bytes / pitch - 1;
    size_t sect = 0, lines = 0;
    { const size_t a = off, b = off + 16 * SEGL - 1; sect = b / 64 - a / 64 + 1; lines = b / 128 - a / 128 + 1; }      // (pitch is a multiple of 128: the same for every row)
    k_rows<SEGL><<<4096, 256 / SEGL * SEGL>>>(buf, pitch, off, nrows, out);
    hipDeviceSynchronize();
    printf("k_rows<%d> pitch %zu off %zu: requested %zu KB, 64-byte sectors %zu KB, 128-byte lines %zu KB\n", SEGL, pitch, off, nrows * 16 * SEGL / 1024,
           nrows * sect * 64 / 1024, nrows * lines * 128 / 1024);
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
    rows_case<4>((const uint8_t *)buf, bytes, 256, 0, out);      // half of one line
    rows_case<4>((const uint8_t *)buf, bytes, 256, 96, out);     // 32 + 32 bytes of two lines
    rows_case<3>((const uint8_t *)buf, bytes, 640, 16, out);     // k_fast_cells: 48-byte rows
    rows_case<3>((const uint8_t *)buf, bytes, 640, 96, out);     // ... across a line boundary
    rows_case<8>((const uint8_t *)buf, bytes, 256, 0, out);      // every other line, whole
    rows_case<1>((const uint8_t *)buf, bytes, 128, 0, out);      // 16 bytes of every line
    printf("known bytes per kernel: %zu (KB: %zu)\n", bytes, bytes / 1024);
    return 0;
}
