// overlap_launch.hip — can a dependent kernel START before its predecessor has finished (and wait for it on a device flag)?
//   mode 0: same stream, plain launches (the baseline: B starts after A ends + boundary)
//   mode 1: same stream, hipExtAnyOrderLaunch on B
//   mode 2: two streams, eager: A on s0, B on s1, B spins on A's completion counter
//   mode 3: the mode-2 pattern captured into a hipGraph (fork / join by events) and replayed
// A: 256 workgroups busy for ~T us, each bumps a counter at its end. B: 256 workgroups record their start, wait for the counter, record their end.
// hipcc --offload-arch=gfx950 -O3 -o overlap_launch overlap_launch.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 now() { u64 t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void kA(u64 * log, unsigned * counter, int ticks) {
    const u64 t0 = now();
    while (now() - t0 < (u64) ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) log[0] = t0;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        log[1] = now();   // (last writer wins: roughly A's end)
    }
}
__global__ void kB(u64 * log, unsigned * counter, unsigned target) {
    const u64 t0 = now();
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
        if (blockIdx.x == 0) { log[2] = t0; log[3] = now(); log[4] = spins; }
    }
}
int main() {
    u64 * log; unsigned * counter;
    CHECK(hipMalloc(&log, 64 * 8)); CHECK(hipMalloc(&counter, 256));
    hipStream_t s0, s1; CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t ev, ev2; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    const int ticks = 1000;   // 10 us at 100 MHz
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemsetAsync(counter, 0, 256, s0)); CHECK(hipMemsetAsync(log, 0, 64 * 8, s0)); CHECK(hipStreamSynchronize(s0));
            if (mode == 0) { kA<<<256, 256, 0, s0>>>(log, counter, ticks); kB<<<256, 256, 0, s0>>>(log, counter, 256); }
            if (mode == 1) { kA<<<256, 256, 0, s0>>>(log, counter, ticks); hipExtLaunchKernelGGL(kB, dim3(256), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, log, counter, 256u); }
            if (mode == 2) { kA<<<256, 256, 0, s0>>>(log, counter, ticks); kB<<<256, 256, 0, s1>>>(log, counter, 256); }
            if (mode == 3) {
                hipGraph_t g; hipGraphExec_t ge;
                CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
                CHECK(hipEventRecord(ev, s0)); CHECK(hipStreamWaitEvent(s1, ev, 0));          // fork
                kA<<<256, 256, 0, s0>>>(log, counter, ticks);
                kB<<<256, 256, 0, s1>>>(log, counter, 256);
                CHECK(hipEventRecord(ev2, s1)); CHECK(hipStreamWaitEvent(s0, ev2, 0));        // join
                CHECK(hipStreamEndCapture(s0, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CHECK(hipGraphLaunch(ge, s0));
                CHECK(hipStreamSynchronize(s0));
                CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
            }
            CHECK(hipStreamSynchronize(s0)); CHECK(hipStreamSynchronize(s1));
            u64 h[8]; CHECK(hipMemcpy(h, log, 64, hipMemcpyDeviceToHost));
            printf("mode %d: A ran %.2f us; B started %+.2f us relative to A's end, waited %.2f us (%llu polls)\n", mode, (h[1] - h[0]) / 100.0, ((double) h[2] - (double) h[1]) / 100.0, (h[3] - h[2]) / 100.0, h[4]);
        }
    }
    return 0;
}
