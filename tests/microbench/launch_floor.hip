// Microbenchmark: cost of a dependent kernel boundary on this box (eager vs hipGraph), and the shader clock a
// short-kernel workload actually gets. Build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void tiny(float * p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void spin(float * p, int iters) {   // pure ALU: iters dependent fmas per lane
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.000001f + 0.5f;
    p[threadIdx.x] = v;
}
__global__ void clocks(unsigned long long * out) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = 1.f;
    for (int i = 0; i < 200000; i++) v = v * 1.000001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long) v; }
}

int main() {
    float * d; CK(hipMalloc(&d, 1 << 20));
    unsigned long long * dc; CK(hipMalloc(&dc, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 2000;
    for (int wgs : {1, 4, 256}) {
        // eager
        for (int i = 0; i < 100; i++) tiny<<<wgs, 256, 0, s>>>(d, wgs * 256);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < N; i++) tiny<<<wgs, 256, 0, s>>>(d, wgs * 256);
        CK(hipStreamSynchronize(s));
        double eager = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) tiny<<<wgs, 256, 0, s>>>(d, wgs * 256);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double graph = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (5.0 * N);
        printf("tiny kernel, %3d WGs: eager %.2f us/launch, hipGraph %.2f us/launch\n", wgs, eager, graph);
    }
    // clock under a short-kernel workload vs after sustained load
    clocks<<<1, 64, 0, s>>>(dc); CK(hipStreamSynchronize(s));
    unsigned long long h[3]; CK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
    printf("cold: shader clock ~ %.0f MHz\n", (double) h[0] / (double) h[1] * 100.0);
    for (int i = 0; i < 200; i++) spin<<<1024, 256, 0, s>>>(d, 200000);
    CK(hipStreamSynchronize(s));
    clocks<<<1, 64, 0, s>>>(dc); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
    printf("after sustained load: shader clock ~ %.0f MHz\n", (double) h[0] / (double) h[1] * 100.0);
    // tiny kernels again right after the load
    {
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < N; i++) tiny<<<4, 256, 0, s>>>(d, 1024);
        CK(hipStreamSynchronize(s));
        printf("tiny eager right after load: %.2f us/launch\n", std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N);
    }
    return 0;
}
