// Operand-slot probe for v_mfma_i32_16x16x64_i8 on gfx950 with exact integer data: prints, for lane group g and element j of the A
// operand, which k the hardware pairs it with in the B operand, and checks the C map (col = lane & 15, row = 4 * (lane >> 4) + reg).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_i8_probe tools/mfma_i8_probe.hip && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
// A[m][slot] and B[slot][n] are given per (lane, element): lane l = (m or n) + 16 g, element j (byte j of the 16-byte operand)
__global__ void k_probe(const int8_t *a, const int8_t *b, int *c)
{
    const int l = threadIdx.x;
    v4i A, B, C = {0, 0, 0, 0};
    const int *ap = reinterpret_cast<const int *>(a + 16 * l), *bp = reinterpret_cast<const int *>(b + 16 * l);
    for (int i = 0; i < 4; i++) { A[i] = ap[i]; B[i] = bp[i]; }
    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C, 0, 0, 0);
    for (int i = 0; i < 4; i++) c[4 * l + i] = C[i];
}
int main()
{
    int8_t ha[1024], hb[1024]; int hc[256];
    int8_t *da, *db; int *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 1024);
    // (1) slot pairing: A = 1 in ONE slot (g, j) of row 0, B = slot id + 1 in every slot of column 0 -> C[0][0] tells the paired B slot
    int bad = 0;
    for (int g = 0; g < 4; g++)
        for (int j = 0; j < 16; j++) {
            for (int i = 0; i < 1024; i++) { ha[i] = 0; hb[i] = 0; }
            ha[16 * (0 + 16 * g) + j] = 1;                               // lane (m = 0, g), element j
            for (int gg = 0; gg < 4; gg++) for (int jj = 0; jj < 16; jj++) hb[16 * (0 + 16 * gg) + jj] = (int8_t)(16 * gg + jj + 1);
            hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dc);
            hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
            const int got = hc[0] - 1;                                   // lane 0, reg 0 = C[row 0][col 0] if the C map holds
            if (got != 16 * g + j) { bad++; printf("A slot (g %d, j %d) pairs with B slot (g %d, j %d)\n", g, j, got / 16, got % 16); }
        }
    printf("slot pairing A(g, j) <-> B(g, j): %s\n", bad ? "NOT the identity (see above)" : "identity for all 64 slots");
    // (2) C map: A[m][slot 0] = m + 1, B[slot 0][n] = n + 1 -> C[m][n] = (m + 1)(n + 1)
    for (int i = 0; i < 1024; i++) { ha[i] = 0; hb[i] = 0; }
    for (int m = 0; m < 16; m++) { ha[16 * m] = (int8_t)(m + 1); hb[16 * m] = (int8_t)(m + 1); }
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dc);
    hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    int cbad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { const int row = 4 * (l >> 4) + r, col = l & 15; if (hc[4 * l + r] != (row + 1) * (col + 1)) cbad++; }
    printf("C map col = lane & 15, row = 4 (lane >> 4) + reg (A rows on lanes & 15, B columns on lanes & 15): %s\n", cbad ? "WRONG" : "holds");
    // (3) signedness: A = -128, B = 127 in one slot -> -16256
    for (int i = 0; i < 1024; i++) { ha[i] = 0; hb[i] = 0; }
    ha[0] = (int8_t)-128; hb[0] = 127;
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dc);
    hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    printf("signed x signed: -128 * 127 = %d\n", hc[0]);
    return 0;
}
