// Diagnostic microbenchmark of the Q4_K mat-vec kernel on the moshika shapes: kernel time (HIP events) and per-phase
// s_memtime stamps of wave 0 (build with -DMV_STAMPS). Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMV_STAMPS -I../../include -I../../moshi.cpp_amd/csrc \
//         mv_bench.hip -o mv_bench      (stand-alone: linked against the product library as well, the run takes the LIBRARY's kernels and records no stamps)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../moshi.cpp_amd/csrc/hip_kernels_fused.hip"
extern "C" void ggml_abort(const char * file, int line, const char * fmt, ...) { va_list ap; va_start(ap, fmt); fprintf(stderr, "%s:%d: ", file, line); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); abort(); }
#include <algorithm>
#include <cstdio>
#include <vector>

static void run(const char * name, int64_t K, int64_t M, int pro) {
    const int64_t row_bytes = K / 256 * 144;
    const size_t wbytes = (size_t) (M * row_bytes);
    char * w; float * x, * alpha, * y, * res;
    HIP_CHECK(hipMalloc(&w, wbytes)); HIP_CHECK(hipMalloc(&x, K * 8)); HIP_CHECK(hipMalloc(&alpha, K * 4)); HIP_CHECK(hipMalloc(&y, M * 4)); HIP_CHECK(hipMalloc(&res, M * 4));
    std::vector<uint8_t> hw(wbytes);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < wbytes; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(&hw[i], &s, 8); }
    for (size_t b = 0; b < wbytes; b += 144) { hw[b] = 0; hw[b + 1] = 0x18; hw[b + 2] = 0; hw[b + 3] = 0x18; }   // sane f16 d/dmin
    HIP_CHECK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t) K * 2, 0.37f);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = (float) ((i * 2654435761u) % 1000) / 500.f - 1.f;
    HIP_CHECK(hipMemcpy(x, hx.data(), K * 8, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(alpha, hx.data(), K * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(res, 0, M * 4));
    mv_args a; memset(&a, 0, sizeof(a));
    a.wtype = GGML_TYPE_Q4_K; a.w = w; a.row_bytes = row_bytes; a.K = K; a.M = M; a.prologue = pro; a.x = x; a.alpha = alpha; a.eps = 1e-8f;
    a.ncols = 1; a.residual = res; a.y = y;
    hipStream_t st; HIP_CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    // flush caches between runs with a big memset so weights come from HBM
    char * junk; HIP_CHECK(hipMalloc(&junk, 512u << 20));
    { static unsigned long long z[4096][8]; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mv_stamps), z, sizeof(z))); }
    std::vector<float> times;
    for (int it = 0; it < 12; it++) {
        HIP_CHECK(hipMemsetAsync(junk, it, 512u << 20, st));
        HIP_CHECK(hipEventRecord(e0, st));
        k_matvec(st, a);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipStreamSynchronize(st));
        float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        times.push_back(ms * 1e3f);
    }
    std::sort(times.begin(), times.end());
    {   // a digest of y: variants of the kernel built with -DMV_EXP_* must print the same one
        std::vector<float> hy((size_t) M);
        HIP_CHECK(hipMemcpy(hy.data(), y, M * 4, hipMemcpyDeviceToHost));
        uint64_t h = 1469598103934665603ull;
        for (float f : hy) { uint32_t u; memcpy(&u, &f, 4); h = (h ^ u) * 1099511628211ull; }
        printf("[y %016llx] ", (unsigned long long) h);
    }
    printf("%-10s K=%5ld M=%6ld pro=%d: median %.2f us (min %.2f) -> %.0f GB/s", name, (long) K, (long) M, pro, times[6], times[0], wbytes / times[6] / 1e3);
#ifdef MV_STAMPS
    static unsigned long long hs[4096][8];
    HIP_CHECK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_mv_stamps), sizeof(hs)));
    // per-phase offsets averaged over all workgroups, plus the distribution of start / end times
    const int nwg = (int) std::min<int64_t>(4096, (M + 15) / 16);
    double ph[8] = { 0 }; int n = 0;
    unsigned long long t0min = ~0ull, t0max = 0, t7max = 0;
    for (int b = 0; b < nwg; b++) { if (!hs[b][0]) continue; n++; for (int i = 0; i < 8; i++) ph[i] += (double) (hs[b][i] - hs[b][0]); t0min = std::min(t0min, hs[b][0]); t0max = std::max(t0max, hs[b][0]); t7max = std::max(t7max, hs[b][7]); }
    static unsigned long long hr[4096][2];
    HIP_CHECK(hipMemcpyFromSymbol(hr, HIP_SYMBOL(g_mv_real), sizeof(hr)));
    unsigned long long r0 = ~0ull, r1 = 0; std::vector<double> ends, starts;
    for (int b = 0; b < nwg; b++) { if (!hr[b][0]) continue; r0 = std::min(r0, hr[b][0]); r1 = std::max(r1, hr[b][1]); }
    for (int b = 0; b < nwg; b++) { if (!hr[b][0]) continue; starts.push_back((hr[b][0] - r0) / 100.0); ends.push_back((hr[b][1] - r0) / 100.0); }
    if (starts.empty()) { printf(" | no stamps recorded (hs[0] = %llu %llu %llu, hr[0] = %llu %llu, nwg %d)\n", hs[0][0], hs[0][1], hs[0][7], hr[0][0], hr[0][1], nwg); return; }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    printf(" | realtime us: start p50 %.2f p99 %.2f max %.2f; end p10 %.2f p50 %.2f p90 %.2f max %.2f", starts[starts.size() / 2], starts[starts.size() * 99 / 100], starts.back(),
           ends[ends.size() / 10], ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back());
    printf(" | cyc avg of %d WGs: ld-issued %.0f, prologue %.0f, sync %.0f, staged %.0f, tiles %.0f, sync %.0f, end %.0f | x", n,
           ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, ph[6] / n, ph[7] / n);
#endif
    printf("\n");
    hipFree(w); hipFree(x); hipFree(alpha); hipFree(y); hipFree(res); hipFree(junk);
}

int main() {
    run("in_proj", 4096, 12288, MV_RMSNORM);
    run("out_proj", 4096, 4096, MV_PLAIN);
    run("linear_in", 4096, 22528, MV_RMSNORM);
    run("linear_out", 11264, 4096, MV_GATE_SILU);
    run("text_lin", 4096, 32000, MV_PLAIN);
    run("dep_in", 1024, 3072, MV_RMSNORM);
    run("dep_out", 1024, 1024, MV_PLAIN);
    run("dep_lout", 2816, 1024, MV_GATE_SILU);
    return 0;
}
