// Which resource makes the mat-vec stream slower than a bare read? (a) occupancy limited by 42 KB LDS per WG, (b) every WG re-reading
// the same 32 KB activation vectors, (c) LDS staging + reads.  hipcc --offload-arch=gfx950 -O3 stream_read2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int XREAD, int STAGE>
__global__ void __launch_bounds__(256) rd(const u32x4 * src, const float4 * x, unsigned * out, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    float4 xv[8];
    if (XREAD) {
#pragma unroll
        for (int j = 0; j < 8; j++) xv[j] = x[j * 256 + threadIdx.x];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int t = blockIdx.x * 4 + wave;
    u32x4 r[9];
    const int tt = t < ntiles ? t : ntiles - 1;
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = __builtin_nontemporal_load(src + (size_t) tt * 576 + i * 64 + lane);
    __builtin_amdgcn_sched_barrier(0);
    if (XREAD) {
        float s = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) s += xv[j].x + xv[j].y + xv[j].z + xv[j].w;
        acc += __float_as_uint(s);
        __syncthreads();
    }
    if (STAGE) {
        char * stage = smem + wave * 9216;
#pragma unroll
        for (int i = 0; i < 9; i++) ((u32x4 *) stage)[i * 64 + lane] = r[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 9; i++) { const u32x4 v = *(const u32x4 *) (stage + lane * 144 + i * 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t maxb = 80u << 20;
    char * buf; unsigned * out; char * junk; float4 * x;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&junk, 512u << 20)); CK(hipMalloc(&x, 32768));
    CK(hipMemset(buf, 1, maxb)); CK(hipMemset(x, 0, 32768));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t mb = 28;
    const int ntiles = (int) (mb * 1000000 / 9216);
    const int grid = (ntiles + 3) / 4;
    struct V { const char * name; int x, stage; size_t lds; } vs[] = {
        { "bare read, full occupancy", 0, 0, 0 }, { "bare read, 42 KB LDS/WG (3 WG/CU)", 0, 0, 42752 }, { "bare read, 24 KB LDS/WG (6 WG/CU)", 0, 0, 24000 },
        { "+32 KB shared x read, 3 WG/CU", 1, 0, 42752 }, { "+LDS staging, 3 WG/CU", 0, 1, 42752 }, { "+x read +staging, 3 WG/CU", 1, 1, 42752 },
        { "+x read +staging, 80 KB (1 WG/CU)", 1, 1, 81000 } };
    for (auto & v : vs) {
        std::vector<float> t;
        for (int it = 0; it < 9; it++) {
            CK(hipMemsetAsync(junk, it, 512u << 20, st));
            CK(hipEventRecord(e0, st));
            if (v.x == 0 && v.stage == 0) rd<0, 0><<<grid, 256, v.lds, st>>>((const u32x4 *) buf, x, out, ntiles);
            if (v.x == 1 && v.stage == 0) rd<1, 0><<<grid, 256, v.lds, st>>>((const u32x4 *) buf, x, out, ntiles);
            if (v.x == 0 && v.stage == 1) rd<0, 1><<<grid, 256, v.lds, st>>>((const u32x4 *) buf, x, out, ntiles);
            if (v.x == 1 && v.stage == 1) rd<1, 1><<<grid, 256, v.lds, st>>>((const u32x4 *) buf, x, out, ntiles);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%3zu MB  %-40s median %6.2f us  -> %5.0f GB/s\n", mb, v.name, t[4], (double) ntiles * 9216 / t[4] / 1e3);
    }
    return 0;
}
