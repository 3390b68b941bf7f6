// Experiment: does prefetching the NEXT mat-vec's weights into the Infinity Cache (MALL) from a forked graph branch, while the
// current mat-vec runs, shorten a dependent chain of weight-streaming kernels?  Chain = 16 x (4096 -> 12288 Q4_K), 28 MB each.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../moshi.cpp_amd/csrc/hip_kernels_fused.hip"
extern "C" void ggml_abort(const char * file, int line, const char * fmt, ...) { va_list ap; va_start(ap, fmt); fprintf(stderr, "%s:%d: ", file, line); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); abort(); }
#include <algorithm>
#include <chrono>
#include <vector>

__global__ void __launch_bounds__(256) prefetch_kernel(const u32x4 * src, size_t n16, unsigned * sink) {
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) { const u32x4 v = src[i]; acc += v.x ^ v.w; }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

int main(int argc, char ** argv) {
    const int N = 16, pf_wgs = argc > 1 ? atoi(argv[1]) : 128;
    const int64_t K = 4096, M = 12288, row_bytes = K / 256 * 144;
    const size_t wbytes = (size_t) (M * row_bytes);
    std::vector<char *> w(N);
    std::vector<uint8_t> hw(wbytes);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < wbytes; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(&hw[i], &s, 8); }
    for (size_t b = 0; b < wbytes; b += 144) { hw[b] = 0; hw[b + 1] = 0x18; hw[b + 2] = 0; hw[b + 3] = 0x18; }
    for (int i = 0; i < N; i++) { HIP_CHECK(hipMalloc(&w[i], wbytes)); HIP_CHECK(hipMemcpy(w[i], hw.data(), wbytes, hipMemcpyHostToDevice)); }
    float * x, * alpha, * y; unsigned * sink;
    HIP_CHECK(hipMalloc(&x, K * 4)); HIP_CHECK(hipMalloc(&alpha, K * 4)); HIP_CHECK(hipMalloc(&y, M * 4 * N)); HIP_CHECK(hipMalloc(&sink, 64));
    std::vector<float> hx((size_t) K, 0.5f);
    HIP_CHECK(hipMemcpy(x, hx.data(), K * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(alpha, hx.data(), K * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2; HIP_CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto mv = [&](int i, hipStream_t st) {
        mv_args a; memset(&a, 0, sizeof(a));
        a.wtype = GGML_TYPE_Q4_K; a.w = w[i]; a.row_bytes = row_bytes; a.K = K; a.M = M; a.prologue = MV_RMSNORM; a.x = x; a.alpha = alpha; a.eps = 1e-8f;
        a.ncols = 1; a.y = y + (size_t) i * M;
        k_matvec(st, a);
    };
    for (int variant = 0; variant < 2; variant++) {
        hipGraph_t g; hipGraphExec_t ge;
        std::vector<hipEvent_t> ev(2 * N + 2);
        for (auto & e : ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) {
            if (variant == 1 && i + 1 < N) {   // fork: prefetch W[i+1] while mat-vec i runs
                HIP_CHECK(hipEventRecord(ev[2 * i], s1));
                HIP_CHECK(hipStreamWaitEvent(s2, ev[2 * i], 0));
                prefetch_kernel<<<pf_wgs, 256, 0, s2>>>((const u32x4 *) w[i + 1], wbytes / 16, sink);
            }
            mv(i, s1);
        }
        if (variant == 1) { HIP_CHECK(hipEventRecord(ev[2 * N], s2)); HIP_CHECK(hipStreamWaitEvent(s1, ev[2 * N], 0)); }
        HIP_CHECK(hipStreamEndCapture(s1, &g));
        HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<double> t;
        for (int it = 0; it < 8; it++) {
            HIP_CHECK(hipStreamSynchronize(s1));
            auto t0 = std::chrono::high_resolution_clock::now();
            HIP_CHECK(hipGraphLaunch(ge, s1));
            HIP_CHECK(hipStreamSynchronize(s1));
            t.push_back(std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count());
        }
        std::sort(t.begin(), t.end());
        printf("%s: %d x 28 MB chain: median %.1f us total = %.2f us per mat-vec -> %.0f GB/s\n", variant ? "with MALL prefetch branch" : "plain chain           ", N,
               t[4], t[4] / N, (double) wbytes * N / t[4] / 1e3);
    }
    return 0;
}
