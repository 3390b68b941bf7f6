// Two launches in flight without cross-stream graph edges: a dependent chain of N kernels, (A) on one stream, each launch behind its predecessor the usual
// way; (B) alternating over two streams (forked once, joined once) with the dependency carried INSIDE the kernels - every launch requests its bytes first,
// then waits for its predecessor's counter of finished workgroups, then does its dependent part. Both captured into a hipGraph and replayed.
// Build: hipcc --offload-arch=gfx950 -O3 -o tests/microbench/two_chains tests/microbench/two_chains.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// grid G x 256 threads; streams `kb` KB per workgroup from w, then (flagged) waits until *wait_for >= (own launches so far + 1) * G, then reads x (written by
// the predecessor), "computes" for `work` sleeps, writes y, counts itself finished
__global__ void __launch_bounds__(256) step_kernel(const u32x4 * w, int kb, const float * x, float * y, int work, u64 * done, const u64 * wait_for, unsigned * err) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, G = gridDim.x;
    u32x4 acc = { 0u, 0u, 0u, 0u };
    const u32x4 * src = w + (size_t) blockIdx.x * kb * 64;
    // kb KB per workgroup in batches of 72 KB: 18 x 16 B per thread, all requested before the first is used (the mat-vec's tiles in flight)
    for (int base = 0; base < kb * 64; base += 18 * 256) {
        u32x4 r[18];
#pragma unroll
        for (int i = 0; i < 18; i++) { const int j = base + i * 256 + tid; r[i] = __builtin_nontemporal_load(src + (j < kb * 64 ? j : kb * 64 - 1)); }
#pragma unroll
        for (int i = 0; i < 18; i++) { acc.x ^= r[i].x; acc.y ^= r[i].y; acc.z ^= r[i].z; acc.w ^= r[i].w; }
    }
    if (wait_for) {
        if (tid == 0) {
            const u64 mine = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 target = (mine / (u64) G + 1ull) * (u64) G;
            unsigned spins = 0;
            while (__hip_atomic_load(wait_for, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > (1u << 22)) { *err = 1u; break; } __builtin_amdgcn_s_sleep(2); }
        }
        __syncthreads();
    }
    // (hand-off bytes: agent-scope loads and stores - served at the device's coherence point, no cache write-back / invalidate)
    float v = __hip_atomic_load(x + (blockIdx.x * 256 + tid) % (G * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 0; i < work; i++) __builtin_amdgcn_s_sleep(8);
    ((volatile char *) smem)[tid] = (char) acc.x;
    if (tid < 16) __hip_atomic_store(y + blockIdx.x * 16 + tid, v + 1.0f + (float) ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) (void) __hip_atomic_fetch_add(done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char ** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 128, kb = argc > 2 ? atoi(argv[2]) : 96, work = argc > 3 ? atoi(argv[3]) : 40, lds = argc > 4 ? atoi(argv[4]) : 60 * 1024;
    const int G = 256, reps = 50;
    u32x4 * w; float * xy; u64 * ctr; unsigned * err;
    const size_t wbytes = (size_t) N * G * kb * 1024;
    CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
    CK(hipMalloc(&xy, (size_t) (N + 1) * G * 16 * 4)); CK(hipMemset(xy, 0, (size_t) (N + 1) * G * 16 * 4));
    CK(hipMalloc(&ctr, (size_t) (N + 1) * 256)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    CK(hipFuncSetAttribute((const void *) step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t ef, ej, t0, t1; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int mode = 0; mode < 3; mode++) {   // 0: one stream, plain; 1: two streams, flags, graph; 2: two streams, flags, eager
        CK(hipMemset(ctr, 0, (size_t) (N + 1) * 256));
        CK(hipDeviceSynchronize());
        auto enqueue = [&]() {
            if (mode) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
            for (int k = 0; k < N; k++) {
                hipStream_t s = mode ? (k & 1 ? s1 : s0) : s0;
                u64 * done = ctr + (size_t) (k + 1) * 32;
                const u64 * wf = mode && k > 0 ? ctr + (size_t) k * 32 : nullptr;
                step_kernel<<<G, 256, lds, s>>>(w + (size_t) k * G * kb * 64, kb, xy + (size_t) k * G * 16, xy + (size_t) (k + 1) * G * 16, work, done, wf, err);
            }
            if (mode) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
        };
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (mode < 2) {
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            enqueue();
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        for (int r = 0; r < 5; r++) { if (ge) CK(hipGraphLaunch(ge, s0)); else enqueue(); }
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(t0, s0));
        for (int r = 0; r < reps; r++) { if (ge) CK(hipGraphLaunch(ge, s0)); else enqueue(); }
        CK(hipEventRecord(t1, s0));
        CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        float last; CK(hipMemcpy(&last, xy + (size_t) N * G * 16, 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %8.2f us per chain of %d, %6.2f us per launch; err %u; y_last %.0f (expect %d)\n", mode,
               mode == 0 ? "one stream, graph" : mode == 1 ? "two streams + in-kernel flags, graph" : "two streams + in-kernel flags, eager", 1e3 * ms / reps, N, 1e3 * ms / reps / N, e, last, N);
        if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
    }
    return 0;
}
