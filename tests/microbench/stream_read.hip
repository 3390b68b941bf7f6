// Ceiling check: how fast can a single short kernel stream N MB from HBM on MI355X with the mat-vec's access pattern
// (each wave reads one contiguous 9216 B tile with nine 16 B/lane loads)?  hipcc --offload-arch=gfx950 -O3 stream_read.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT, int TILES>
__global__ void __launch_bounds__(256) rd(const u32x4 * src, unsigned * out, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned acc = 0;
    for (int k = 0; k < TILES; k++) {
        const int t = (blockIdx.x * nw + wave) * TILES + k;
        if (t >= ntiles) break;
        u32x4 r[9];
#pragma unroll
        for (int i = 0; i < 9; i++) r[i] = NT ? __builtin_nontemporal_load(src + (size_t) t * 576 + i * 64 + lane) : src[(size_t) t * 576 + i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 9; i++) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t maxb = 80u << 20;
    char * buf; unsigned * out; char * junk;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&junk, 512u << 20));
    CK(hipMemset(buf, 1, maxb));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : { 9, 28, 50, 73 }) {
        const int ntiles = (int) (mb * 1000000 / 9216);
        for (int variant = 0; variant < 6; variant++) {
            std::vector<float> t;
            for (int it = 0; it < 9; it++) {
                CK(hipMemsetAsync(junk, it, 512u << 20, st));
                CK(hipEventRecord(e0, st));
                const int threads = variant >= 4 ? 512 : 256;
                const int nw = threads / 64;
                switch (variant) {
                    case 0: rd<1, 1><<<(ntiles + nw - 1) / nw, threads, 0, st>>>((const u32x4 *) buf, out, ntiles); break;
                    case 1: rd<0, 1><<<(ntiles + nw - 1) / nw, threads, 0, st>>>((const u32x4 *) buf, out, ntiles); break;
                    case 2: rd<1, 2><<<(ntiles + 2 * nw - 1) / (2 * nw), threads, 0, st>>>((const u32x4 *) buf, out, ntiles); break;
                    case 3: rd<1, 4><<<(ntiles + 4 * nw - 1) / (4 * nw), threads, 0, st>>>((const u32x4 *) buf, out, ntiles); break;
                    case 4: rd<1, 1><<<(ntiles + nw - 1) / nw, threads, 0, st>>>((const u32x4 *) buf, out, ntiles); break;
                    case 5: rd<1, 2><<<(ntiles + 2 * nw - 1) / (2 * nw), threads, 0, st>>>((const u32x4 *) buf, out, ntiles); break;
                }
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
            }
            std::sort(t.begin(), t.end());
            static const char * names[] = { "nt 256thr 1 tile/wave", "plain 256thr 1 tile/wave", "nt 256thr 2 tiles/wave", "nt 256thr 4 tiles/wave", "nt 512thr 1 tile/wave", "nt 512thr 2 tiles/wave" };
            printf("%3zu MB  %-26s median %6.2f us  -> %5.0f GB/s\n", mb, names[variant], t[4], (double) ntiles * 9216 / t[4] / 1e3);
        }
    }
    return 0;
}
