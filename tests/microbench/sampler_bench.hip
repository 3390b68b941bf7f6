// Diagnostic microbenchmark of the fused sampler (sample_topk_kernel): device time per call on the two shapes of the moshika frame - the audio heads
// (n = 2 048, k = 250, temperature 0.8) and the text head (n = 32 000, k = 25, temperature 0.7) - back to back in one stream. Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../moshi.cpp_amd/csrc sampler_bench.hip -o sampler_bench
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../moshi.cpp_amd/csrc/hip_kernels_fused.hip"
extern "C" void ggml_abort(const char * file, int line, const char * fmt, ...) { va_list ap; va_start(ap, fmt); fprintf(stderr, "%s:%d: ", file, line); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); abort(); }
#include <cmath>
#include <random>
#include <vector>

static void run(int n, int k, float temp, float sigma) {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, sigma);
    std::exponential_distribution<float> ed(1.f);
    const int reps = 200;
    std::vector<float> hl((size_t) n * reps), hn((size_t) k);
    for (auto & v : hl) v = nd(rng);
    for (auto & v : hn) v = ed(rng);
    float * l, * nz; int32_t * out;
    HIP_CHECK(hipMalloc(&l, hl.size() * 4)); HIP_CHECK(hipMalloc(&nz, hn.size() * 4)); HIP_CHECK(hipMalloc(&out, reps * 4));
    HIP_CHECK(hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(nz, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    hipStream_t s; HIP_CHECK(hipStreamCreate(&s));
    for (int pass = 0; pass < 2; pass++) {
        HIP_CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) { sample_args a = { l + (size_t) r * n, n, 1.f / temp, k, nz, out + r, nullptr }; k_sample_topk(s, a); }
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int32_t> ho(reps); HIP_CHECK(hipMemcpy(ho.data(), out, reps * 4, hipMemcpyDeviceToHost));
    long long chk = 0; for (int r = 0; r < reps; r++) chk = chk * 31 + ho[r];
#if defined(SMP_LOG)
    unsigned long long lg[16]; HIP_CHECK(hipMemcpyFromSymbol(lg, HIP_SYMBOL(g_smp_log), sizeof(lg)));
    printf("   soft_max, us since entry: logits + clear issued %.2f | wave max %.2f | barrier %.2f | exp + f64 sums %.2f | wave sum + barrier %.2f\n", (lg[6] - lg[0]) / 100.0, (lg[7] - lg[0]) / 100.0, (lg[8] - lg[0]) / 100.0, (lg[9] - lg[0]) / 100.0, (lg[10] - lg[0]) / 100.0);
    printf("   last call, us since entry: soft_max %.2f | select %.2f | collect %.2f | rank sort %.2f | end %.2f\n", (lg[1] - lg[0]) / 100.0, (lg[2] - lg[0]) / 100.0, (lg[3] - lg[0]) / 100.0, (lg[4] - lg[0]) / 100.0, (lg[5] - lg[0]) / 100.0);
#endif
    printf("n %6d k %3d temp %.1f sigma %.1f: %7.2f us per call (incl. the launch boundary), token checksum %lld\n", n, k, temp, sigma, 1e3 * ms / reps, chk);
}
int main() {
    for (float sigma : { 1.0f, 3.0f }) { run(2048, 250, 0.8f, sigma); run(32000, 25, 0.7f, sigma); }
    return 0;
}
