// Checks the operand / result lane maps of v_mfma_i32_16x16x32_i8 on gfx950 with exact integer data (asymmetric A and B).
// Build: hipcc --offload-arch=gfx950 -O2 mfma_i8_layout.hip -o mfma_i8_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t * A, const int8_t * B, int * C) {   // A[16][32] row-major, B[32][16] (k-major), C[16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    long a = 0, b = 0;
    for (int j = 0; j < 8; j++) {
        a |= (long) (uint8_t) A[r * 32 + 8 * g + j] << (8 * j);
        b |= (long) (uint8_t) B[(8 * g + j) * 16 + r] << (8 * j);
    }
    i32x4 acc = { 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 4; q++) C[(g * 4 + q) * 16 + r] = acc[q];
}
int main() {
    int8_t hA[16 * 32], hB[32 * 16]; int hC[256], ref[256];
    for (int i = 0; i < 512; i++) { hA[i] = (int8_t) ((i * 7 + 3) % 16); hB[i] = (int8_t) (((i * 13 + 5) % 255) - 127); }
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { int s = 0; for (int kk = 0; kk < 32; kk++) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    int8_t * dA, * dB; int * dC;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += hC[i] != ref[i];
    printf("mfma_i32_16x16x32_i8 layout: %d of 256 wrong (C[0][1]=%d ref %d)\n", bad, hC[1], ref[1]);
    return bad != 0;
}
