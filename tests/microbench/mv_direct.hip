// Experiment: Q4_K mat-vec with weights loaded straight into registers (no LDS transposition, no per-tile barriers):
// 8 lanes per super-block, each lane owns one 16-byte chunk of the nibbles (coalesced) plus the shared 16-byte header;
// every load of a group of 8 (row, iteration) units is requested before the first is consumed.
// Compared against the product kernel (matvec_q4k_kernel, MV_PREQ8K so both start from pre-quantised activations).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../moshi.cpp_amd/csrc mv_direct.hip -o mv_direct
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../moshi.cpp_amd/csrc/hip_kernels_fused.hip"
extern "C" void ggml_abort(const char * file, int line, const char * fmt, ...) { va_list ap; va_start(ap, fmt); fprintf(stderr, "%s:%d: ", file, line); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); abort(); }
#include <algorithm>
#include <vector>

template <int NW, int G>
__global__ void __launch_bounds__(NW * 64) mv_q4k_direct(const char * __restrict__ w, int64_t row_bytes, int nb, int M, int rows_per_wg, const xblk * __restrict__ xq, float * __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    xblk * xs = (xblk *) smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * rows_per_wg;
    const int rows = min(rows_per_wg, M - row0);
    const int iters = (nb + 7) / 8;
    const int my_rows = rows > wave ? (rows - wave + NW - 1) / NW : 0;
    const int units = my_rows * iters;
    const int c = lane & 7, sbl = lane >> 3;        // chunk within the super-block, super-block within the wave-iteration
    const int g = c >> 1, h = c & 1;
    auto unit_ptr = [&](int u, bool & valid) {
        const int r = u / iters, it = u - r * iters;
        const int sb = it * 8 + sbl;
        valid = u < units && sb < nb;
        const int rr = u < units ? r : (my_rows > 0 ? my_rows - 1 : 0);
        const int sbb = sb < nb ? sb : nb - 1;
        return w + (int64_t) (row0 + wave + rr * NW) * row_bytes + (int64_t) sbb * 144;
    };
    u32x4 qv[G], hv[G];
    bool ok[G];
    int u0 = 0;
    auto issue = [&](int base) {
#pragma unroll
        for (int i = 0; i < G; i++) {
            const char * p = unit_ptr(base + i, ok[i]);
            hv[i] = __builtin_nontemporal_load((const u32x4 *) p);
            qv[i] = __builtin_nontemporal_load((const u32x4 *) (p + 16 + c * 16));
        }
    };
    if (my_rows > 0 || true) issue(0);
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < nb * (XBLK_BYTES / 16); i += NW * 64) ((u32x4 *) xs)[i] = ((const u32x4 *) xq)[i];
    lds_barrier();
    float acc = 0.f;
    for (; u0 < units; u0 += G) {
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int u = u0 + i;
            if (u < units) {
                const int r = u / iters, it = u - r * iters;
                const int sb = it * 8 + sbl;
                float term = 0.f;
                if (ok[i]) {
                    const uint32_t u0w = hv[i][1], u1w = hv[i][2], u2w = hv[i][3];
                    const uint32_t scw = g < 2 ? (u0w & 0x3f3f3f3fu) : ((u2w & 0x0f0f0f0fu) | (((u0w >> 6) & 0x03030303u) << 4));
                    const uint32_t mnw = g < 2 ? (u1w & 0x3f3f3f3fu) : (((u2w >> 4) & 0x0f0f0f0fu) | (((u1w >> 6) & 0x03030303u) << 4));
                    const int sh = ((2 * g) & 3) * 8;
                    const int sc_lo = (scw >> sh) & 0xff, sc_hi = (scw >> (sh + 8)) & 0xff, m_lo = (mnw >> sh) & 0xff, m_hi = (mnw >> (sh + 8)) & 0xff;
                    const xblk * xb = xs + sb;
                    const int4 ylo = *(const int4 *) (xb->q + 64 * g + 16 * h), yhi = *(const int4 *) (xb->q + 64 * g + 32 + 16 * h);
                    const int yl[4] = { ylo.x, ylo.y, ylo.z, ylo.w }, yh[4] = { yhi.x, yhi.y, yhi.z, yhi.w };
                    int ilo = 0, ihi = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const uint32_t q = qv[i][t];
                        ilo = dot4_i8((int) (q & 0x0F0F0F0Fu), yl[t], ilo);
                        ihi = dot4_i8((int) ((q >> 4) & 0x0F0F0F0Fu), yh[t], ihi);
                    }
                    const int isum = __mul24(sc_lo, ilo) + __mul24(sc_hi, ihi);
                    const int msum = __mul24(m_lo, (int) xb->bsums[4 * g + h]) + __mul24(m_hi, (int) xb->bsums[4 * g + 2 + h]);
                    const float d = h2f((uint16_t) (hv[i][0] & 0xffff)) * xb->d, dmin = h2f((uint16_t) (hv[i][0] >> 16)) * xb->d;
                    term = d * (float) isum - dmin * (float) msum;
                }
                acc += term;
                if (it == iters - 1) {   // row complete (wave-uniform)
                    float s = row16_allsum_f32(acc);
                    s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16)) +
                        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 48));
                    if (lane == 0) y[row0 + wave + r * NW] = s;
                    acc = 0.f;
                }
            }
        }
        if (u0 + G < units) issue(u0 + G);
    }
}

static void run(const char * name, int64_t K, int64_t M) {
    const int nb = (int) (K / 256);
    const int64_t row_bytes = (int64_t) nb * 144;
    const size_t wbytes = (size_t) (M * row_bytes);
    char * w; float * x, * y0, * y1; void * xq;
    HIP_CHECK(hipMalloc(&w, wbytes)); HIP_CHECK(hipMalloc(&x, K * 8)); HIP_CHECK(hipMalloc(&y0, M * 4)); HIP_CHECK(hipMalloc(&y1, M * 4)); HIP_CHECK(hipMalloc(&xq, (size_t) nb * XBLK_BYTES));
    std::vector<uint8_t> hw(wbytes);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < wbytes; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(&hw[i], &s, 8); }
    for (size_t b = 0; b < wbytes; b += 144) { hw[b] = 0; hw[b + 1] = 0x18; hw[b + 2] = 0; hw[b + 3] = 0x18; }
    HIP_CHECK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t) K * 2);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = (float) ((i * 2654435761u) % 1000) / 500.f - 1.f;
    HIP_CHECK(hipMemcpy(x, hx.data(), K * 8, hipMemcpyHostToDevice));
    hipStream_t st; HIP_CHECK(hipStreamCreate(&st));
    // pre-quantised activations: silu(l)*r of the 2K vector, through the product's gate kernel
    k_gate_quant_q8k(st, x, K, xq, GGML_TYPE_Q4_K);
    mv_args a; memset(&a, 0, sizeof(a));
    a.wtype = GGML_TYPE_Q4_K; a.w = w; a.row_bytes = row_bytes; a.K = K; a.M = M; a.prologue = MV_PREQ8K; a.x = (const float *) xq; a.ncols = 1; a.y = y0;
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    char * junk; HIP_CHECK(hipMalloc(&junk, 512u << 20));
    auto time_it = [&](auto && launch) {
        std::vector<float> t;
        for (int it = 0; it < 12; it++) {
            HIP_CHECK(hipMemsetAsync(junk, it, 512u << 20, st));
            HIP_CHECK(hipEventRecord(e0, st)); launch(); HIP_CHECK(hipEventRecord(e1, st)); HIP_CHECK(hipStreamSynchronize(st));
            float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end()); return t[6];
    };
    const float t_prod = time_it([&] { k_matvec(st, a); });
    float best = 1e9f; const char * best_name = "";
    auto try_direct = [&](auto kern, int nw, int grid, const char * nm) {
        const int rows_per_wg = (int) ((M + grid - 1) / grid);
        const size_t smem = (size_t) nb * XBLK_BYTES;
        const float t = time_it([&] { kern<<<grid, nw * 64, smem, st>>>(w, row_bytes, nb, (int) M, rows_per_wg, (const xblk *) xq, y1); });
        printf("    direct %-14s %7.2f us\n", nm, t);
        if (t < best) { best = t; best_name = nm; }
    };
    try_direct(mv_q4k_direct<12, 8>, 12, 256, "12w g8 x256");
    try_direct(mv_q4k_direct<8, 8>, 8, 256, "8w g8 x256");
    try_direct(mv_q4k_direct<8, 8>, 8, 512, "8w g8 x512");
    try_direct(mv_q4k_direct<4, 8>, 4, 1024, "4w g8 x1024");
    try_direct(mv_q4k_direct<16, 4>, 16, 256, "16w g4 x256");
    try_direct(mv_q4k_direct<12, 4>, 12, 256, "12w g4 x256");
    std::vector<float> r0(M), r1(M);
    HIP_CHECK(hipMemcpy(r0.data(), y0, M * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(r1.data(), y1, M * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; for (int64_t i = 0; i < M; i++) { maxd = std::max(maxd, (double) fabsf(r0[i] - r1[i])); maxv = std::max(maxv, (double) fabsf(r0[i])); }
    printf("%-10s K=%5ld M=%6ld: product %.2f us (%.0f GB/s) | best direct %s %.2f us (%.0f GB/s) | max rel diff %.2e\n", name, (long) K, (long) M, t_prod, wbytes / t_prod / 1e3, best_name, best, wbytes / best / 1e3, maxd / maxv);
    hipFree(w); hipFree(x); hipFree(y0); hipFree(y1); hipFree(xq); hipFree(junk);
}

int main() {
    run("in_proj", 4096, 12288);
    run("out_proj", 4096, 4096);
    run("linear_in", 4096, 22528);
    run("linear_out", 11264, 4096);
    run("text_lin", 4096, 32000);
    run("dep_in", 1024, 3072);
    return 0;
}
