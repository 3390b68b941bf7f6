// lds_dma_rate.hip — how fast can ONE wave (or a few) per CU issue LDS-DMA fills? (hipcc --offload-arch=gfx950 -O3 -o lds_dma_rate lds_dma_rate.hip)
// Each loader wave copies `nslots` consecutive 1 KB pieces of its own region of a large buffer into an LDS ring, keeping at most `depth` fills
// outstanding. Reports ns per KB per wave and the aggregate GB/s, for several instruction forms.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define GLOBAL_AS __attribute__((address_space(1)))
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int MODE>
__global__ void __launch_bounds__(576) k(const char * src, size_t region, int nslots, unsigned long long * out, int nload) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= nload) return;
    const GLOBAL_AS char * base = (const GLOBAL_AS char *) src + ((size_t) blockIdx.x * nload + wave) * region;
    const unsigned lds0 = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char *) ring + wave * 16384;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    u32x4 acc = { 0, 0, 0, 0 };
    for (int s = 0; s < nslots; s++) {
        const GLOBAL_AS char * p = base + (size_t) s * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (s & 15) * 1024);
        if (MODE == 0) { unsigned keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory"); }
        if (MODE == 1) { unsigned keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory"); }
        if (MODE == 2) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *) p, (__attribute__((address_space(3))) void *) (ring + wave * 16384 + (s & 15) * 1024), 16, 0, 2);
        if (MODE == 3) { const u32x4 v = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) p); acc += v; }
        if (MODE == 4) { unsigned keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory"); }
        if (MODE != 3 && MODE != 2) wait_vmcnt<48>();
    }
    wait_vmcnt<0>();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[blockIdx.x * nload + wave] = t1 - t0 + (MODE == 3 ? (acc.x & 1) : 0);
}

int main(int argc, char ** argv) {
    const int nslots = argc > 1 ? atoi(argv[1]) : 512;
    const size_t region = (size_t) nslots * 1024;
    char * src; unsigned long long * out;
    const size_t total = region * 256 * 8;
    CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total));
    CHECK(hipMalloc(&out, 256 * 8 * 8));
    const char * names[5] = { "global_load_lds_dwordx4 vaddr nt (asm)", "global_load_lds_dwordx4 vaddr (asm)", "builtin global_load_lds 16 nt", "global_load_dwordx4 nt to registers", "global_load_lds_dword (256 B per instruction)" };
    for (int mode = 0; mode < 5; mode++)
        for (int grid : { 64, 256 })
            for (int nload : { 1, 4 }) {
                if (mode == 2 && nload > 4) continue;
                std::vector<unsigned long long> h(grid * nload);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    const size_t smem = 8 * 16384;
                    switch (mode) {
                        case 0: hipFuncSetAttribute((const void *) k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem); k<0><<<grid, 576, smem>>>(src, region, nslots, out, nload); break;
                        case 1: hipFuncSetAttribute((const void *) k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem); k<1><<<grid, 576, smem>>>(src, region, nslots, out, nload); break;
                        case 2: hipFuncSetAttribute((const void *) k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem); k<2><<<grid, 576, smem>>>(src, region, nslots, out, nload); break;
                        case 3: hipFuncSetAttribute((const void *) k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem); k<3><<<grid, 576, smem>>>(src, region, nslots, out, nload); break;
                        case 4: hipFuncSetAttribute((const void *) k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem); k<4><<<grid, 576, smem>>>(src, region, nslots, out, nload); break;
                    }
                    hipEventRecord(e1); CHECK(hipDeviceSynchronize());
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
                double cyc = 0; for (auto v : h) cyc += (double) v; cyc /= h.size();
                const double bytes = (double) grid * nload * nslots * (mode == 4 ? 256.0 : 1024.0);
                printf("%-48s grid %3d loaders/WG %d: %7.1f cycles per instruction per wave, kernel %.1f us, %.0f GB/s aggregate (%.1f GB/s per CU)\n", names[mode], grid, nload, cyc / nslots, ms * 1e3,
                       bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9 / grid);
            }
    return 0;
}
