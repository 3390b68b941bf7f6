// hip_stream.hip — the persistent stream engine (EXPERIMENT, off by default: MI355X_STREAM=1 or backend flag 128 switch it on): the run of LARGE
// Q4_K mat-vecs between two attention launches of the Temporal transformer (/root/reference/src/moshi/modules/transformer.h:300-420,
// StreamingTransformerLayer: out_proj + residual -> norm2 + gated linear_in -> linear_out + residual -> next layer's norm1 + in_proj; 116 MB of
// weights per layer at 4096 / 11264) executed by ONE launch.
//
// Idea: as four launches a layer's mat-vecs ramp their HBM stream up from nothing and drain it four times, although the NEXT matrix's bytes never
// depended on anything. Here 256 workgroups stay resident across the run:
//   * 8 STREAMER waves per workgroup each keep a ring of eight passes (a pass = 8 super-blocks of 144 B, 8 lanes per super-block - the WS = 1
//     arithmetic of matvec_q4k_kernel) of weight requests in flight that runs straight THROUGH the phase boundaries. Wave w owns rows
//     [w rows / 8, (w + 1) rows / 8) of the workgroup's rows of every matrix, so row sums, epilogue and publication need no workgroup barrier;
//   * 4 GATHERER waves per workgroup have nothing but hand-off polls in their vector-memory queue: they sweep the previous phase's output (the chain
//     engine's 8-byte {tag, value} granules, hip_chain.hip), run the RMS norm in matvec_q4k_kernel's summation order, quantise to Q8_K into LDS and
//     release the streamers through an LDS counter. No workgroup barrier after the table copy, no grid barrier, no fence;
//   * arithmetic is the unchained kernels' to the bit: tests/test_stream_engine.py compares the two plans bit for bit.
//
// Measured (profiles/r03_stream_engine_stamps.txt): inside a phase the ring streams at 5.6 - 6 TB/s, but a phase boundary costs 5 - 9 us - wave skew
// inside the workgroup (1 - 3.5 us: 5 vs 6 row pairs per wave in linear_in), one sweep of the granules (2 us per 6 blocks: with ~72 KB per CU of
// weight requests queued in the fabric a poll's round trip IS the ring's depth, whichever wave issues it), norm + quantisation (1.7 - 2.6 us on
// four waves) - against ~2 us of kernel boundary plus ~3.5 us of ramp for a separate launch. Net: Temporal 2 050 us per frame with the engine, 1 600 us
// without. The lesson matches the chain engine's: an all-to-all hand-off between 256 workgroups under full HBM load is not cheaper than a kernel
// boundary on this chip; a deeper ring buys throughput and pays it back in poll latency.
// The LDS-staged variant was built and measured too (not kept: this file is the register-ring form): every streamer wave's weight bytes as one stream
// of 1 KB chunks dropped by LDS-DMA (global_load_lds_dwordx4) into a private 14 KB LDS ring, requests paced to 5 chunks in flight per wave
// (s_waitcnt vmcnt(4) in front of every request) and issued also while the wave waits for the next phase's blocks, eight gatherer waves. Bit-identical
// as well; LM step 3 130 us against 2 889 for this form and 2 413 for one launch per mat-vec: the paced stream reaches 4.1 TB/s inside a phase, and the
// hand-off is no shorter with the fabric kept busy through it (poll sweep 2.3 - 4.3 us, wave skew up to 6.5 us in linear_in, norm + quantisation
// 1.6 - 3 us) - longer than the 5 us of stream a 14 KB ring per wave can absorb.
#include "hip_common.h"
#include "hip_device.h"
#include "hip_mv_device.h"

#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef unsigned long long u64;

#define ST_NW        8                    // streamer waves: weights -> dots -> rows -> publication
#define ST_NG        4                    // gatherer waves: hand-off polls -> norm -> Q8_K blocks
#define ST_THREADS   ((ST_NW + ST_NG) * 64)
#define ST_GRID      256
#define ST_NB_MAX    44                   // K <= 11264
#define ST_K_MAX     (ST_NB_MAX * 256)
#define ST_PASS_MAX  24                   // passes per wave and phase (linear_in: 6 row pairs x 16 super-blocks x 2 / 8)
#define ST_SBW       (ST_PASS_MAX * 8)    // super-blocks per wave and phase
#define ST_RW_MAX    12                   // rows per wave and phase (one half of a paired phase)
#define ST_PH_MAX    8                    // phases per launch
#ifndef ST_RH
#define ST_RH        4                    // passes per half of the ring (a round)
#endif
#define ST_RND_MAX   128                  // rounds per launch
#define ST_SPIN_MAX  (1u << 22)

struct st_phase {
    const char * w; long long row_bytes;
    const float * x; const float * alpha; const float * residual; float * y;
    long long pair_F;            // > 0: paired gate - rows [0, F) and [F, 2 F) of W, y receives silu(l) * r (F values)
    int K, nb, rows, rows_prev;  // rows: per workgroup (paired: per half); rows_prev: rows per workgroup of the phase whose output is x (sentinels)
    int prologue, x_chain, res, res_src;   // res: 0 none, 1 rows kept in LDS by phase res_src, 2 memory
    int n_pub, n_pass; float eps, inv_nb;
    int pad_[6];
};
static_assert(sizeof(st_phase) % 16 == 0, "descriptors are copied to LDS by 16-byte lanes");

struct st_params {
    const st_phase * phases; const int2 * rounds;   // rounds[r] = (phase, first pass of the round inside the phase)
    int n_phases, n_rounds;
    u64 * gbuf;                 // [2][ST_K_MAX] granules
    u64 * flags;                // [2][ST_GRID] one granule per workgroup and phase parity: "this workgroup has published its rows" (a hint: the data granules carry the proof)
    unsigned * launch_seq;
    unsigned * err;
};

struct st_ctl { unsigned failed; unsigned blocks_ready; unsigned norm_count; unsigned pub_count; unsigned go; unsigned pad[3]; double sumsq[ST_NW]; };

#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ GLOBAL_AS T * gp(T * p) { return (GLOBAL_AS T *) p; }
template <typename T> __device__ __forceinline__ const GLOBAL_AS T * gp(const T * p) { return (const GLOBAL_AS T *) p; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void * p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int) bytes, 0x00020000); }
__device__ __forceinline__ u32x4 st_ld16_agent(__amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int) byte_off, 0, 16); }
__device__ __forceinline__ void st_publish(u64 * p, unsigned tag, unsigned value) { __hip_atomic_store(p, ((u64) tag << 32) | (u64) value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned st_lds_load(unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T> __device__ __forceinline__ T st_uniform(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
    if constexpr (sizeof(T) == 4) { unsigned u; __builtin_memcpy(&u, &v, 4); u = __builtin_amdgcn_readfirstlane(u); __builtin_memcpy(&v, &u, 4); return v; }
    else { unsigned u[2]; __builtin_memcpy(u, &v, 8); u[0] = __builtin_amdgcn_readfirstlane(u[0]); u[1] = __builtin_amdgcn_readfirstlane(u[1]); __builtin_memcpy(&v, u, 8); return v; }
}

#if defined(ST_LOG)
// [0] workgroup 0, [1] last workgroup; per phase: streamer wave 0: 0 phase start, 1 blocks ready, 2 dots done, 3 published; gatherer 0: 4 start, 5 sentinels, 6 gathered, 7 blocks written
__device__ u64 g_st_log[2][64][8];
#define ST_STAMP(i) do { if (lane == 0 && (wave == 0 || wave == ST_NW) && (wg == 0 || wg == (int) gridDim.x - 1) && p < 64) { u64 t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    g_st_log[wg ? 1 : 0][p][i] = t_; } } while (0)
extern "C" __attribute__((visibility("default"))) void mi355x_stream_log_read(unsigned long long * dst) { (void) hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_st_log), sizeof(g_st_log)); }
#else
#define ST_STAMP(i) do {} while (0)
#endif

__global__ void __launch_bounds__(ST_THREADS) matvec_stream_kernel(st_params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    xblk * xs = (xblk *) smem;                                  // [2][ST_NB_MAX] Q8_K blocks of the current / next activation vector
    float * part = (float *) (xs + 2 * ST_NB_MAX);              // [ST_NW][ST_SBW] super-block partial sums, private to the streamer wave that owns the rows
    float * ysave = part + ST_NW * ST_SBW;                      // [ST_PH_MAX][ST_NW][ST_RW_MAX] a wave's output rows, per phase (residuals of later phases)
    st_ctl * ctl = (st_ctl *) (ysave + ST_PH_MAX * ST_NW * ST_RW_MAX);
    st_phase * phl = (st_phase *) (ctl + 1);                    // the descriptors
    int2 * rnd = (int2 *) (phl + ST_PH_MAX);                    // [n_rounds]: (phase, first pass)

    {   // tables -> LDS
        const int n16 = P.n_phases * (int) (sizeof(st_phase) / 16);
        for (int i = tid; i < n16; i += ST_THREADS) ((u32x4 *) phl)[i] = ((const GLOBAL_AS u32x4 *) P.phases)[i];
        for (int i = tid; i < P.n_rounds; i += ST_THREADS) ((u64 *) rnd)[i] = ((const GLOBAL_AS u64 *) P.rounds)[i];
        if (tid == 0) { ctl->failed = 0; ctl->blocks_ready = 0; ctl->norm_count = 0; ctl->pub_count = 0; ctl->go = 0; }
    }
    __syncthreads();   // the only workgroup-wide barrier: from here on streamers and gatherers meet through LDS counters
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    auto give_up = [&]() { if (lane == 0) { __hip_atomic_store(&ctl->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); *gp(P.err) = 3u; } };

    if (wave >= ST_NW) {
        // ================================================= gatherers =========================================================================
        // Their vector-memory queue holds nothing but hand-off polls, so a poll returns in one fabric round trip (a streamer's would return behind its
        // sixteen outstanding weight requests: ~ the ring's depth, 3.5 us). Gatherer g takes blocks g, g + 4, ...: polls their granules, runs the norm in
        // matvec_q4k_kernel's summation order (virtual wave w = blocks w, w + 8: gatherer g holds virtual waves g and g + 4 whole), quantises to Q8_K.
        const int gw = wave - ST_NW;
        const __amdgpu_buffer_rsrc_t gb = st_rsrc(P.gbuf, 2u * ST_K_MAX * 8u);
        unsigned norm_epoch = 0;
        for (int p = 0; p < P.n_phases; p++) {
            const st_phase * d = phl + p;
            const int nb = st_uniform(d->nb), K = st_uniform(d->K);
            const int prologue = st_uniform(d->prologue), x_chain = st_uniform(d->x_chain);
            const unsigned tag_in = tag_base | (unsigned) p;
            const unsigned in_base = (unsigned) ((p - 1) & 1) * (ST_K_MAX * 8u);
            xblk * xcur = xs + (p & 1) * ST_NB_MAX;
            const float * xp = st_uniform(d->x); const float * ap = st_uniform(d->alpha);
            ST_STAMP(4);
            if (x_chain) {
                // ... and wait (LDS) until this workgroup's own streamers have published the previous phase: the workgroups run in step, so that is when
                // the others' rows start to appear - sweeping the data granules any earlier only loads the fabric for the whole length of the dot phase
                unsigned spins = 0;
                while (st_lds_load(&ctl->pub_count) < (unsigned) (ST_NW * p)) {
                    if (++spins > ST_SPIN_MAX || st_lds_load(&ctl->failed)) { give_up(); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            ST_STAMP(5);
            auto chunk = [&](auto xb_tag, int i0) {   // blocks gw + 4 (i0 + i), i < XB
                constexpr int XB = decltype(xb_tag)::value;
                float v[XB][4];
                f32x4 al[XB];
#pragma unroll
                for (int i = 0; i < XB; i++) {
                    const int b = gw + ST_NG * (i0 + i), bc = b < nb ? b : nb - 1;
                    al[i] = (f32x4) { 1.f, 1.f, 1.f, 1.f };
                    if (prologue == MV_RMSNORM) al[i] = *(const GLOBAL_AS f32x4 *) (gp(ap) + bc * 256 + lane * 4);
                }
                if (x_chain) {
                    u32x4 g[XB][2];
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int i = 0; i < XB; i++) {
                            const int b = gw + ST_NG * (i0 + i), bc = b < nb ? b : nb - 1;
                            const unsigned o = in_base + ((unsigned) bc * 256u + (unsigned) lane * 4u) * 8u;
                            g[i][0] = st_ld16_agent(gb, o); g[i][1] = st_ld16_agent(gb, o + 16u);
                        }
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < XB; i++) ok = ok && g[i][0].y == tag_in && g[i][0].w == tag_in && g[i][1].y == tag_in && g[i][1].w == tag_in;
                        if (__all(ok)) break;
                        if (++spins > ST_SPIN_MAX || st_lds_load(&ctl->failed)) { give_up(); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int i = 0; i < XB; i++) {
                        v[i][0] = __uint_as_float(g[i][0].x); v[i][1] = __uint_as_float(g[i][0].z); v[i][2] = __uint_as_float(g[i][1].x); v[i][3] = __uint_as_float(g[i][1].z);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < XB; i++) {
                        const int b = gw + ST_NG * (i0 + i), bc = b < nb ? b : nb - 1;
                        const f32x4 t = *(const GLOBAL_AS f32x4 *) (gp(xp) + bc * 256 + lane * 4);
                        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
                    }
                }
                ST_STAMP(6);
                if (XB == 4 && prologue == MV_RMSNORM) {
                    // matvec_q4k_kernel's order: thread (w, lane) adds the squares of its values of block w, then of block w + 8, in double; wave butterfly;
                    // the eight waves' sums are added in index order. (K <= 4096: blocks i = 0 .. 3 of this gatherer are w = gw: {0, 2}, w = gw + 4: {1, 3}.)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        double acc = 0;
#pragma unroll
                        for (int jj = 0; jj < 2; jj++) {
                            const int i = h + 2 * jj;
                            if (gw + ST_NG * i < nb)
#pragma unroll
                                for (int k = 0; k < 4; k++) acc += (double) (v[i][k] * v[i][k]);
                        }
                        acc = wave_allsum_f64(acc);
                        if (lane == 0) ctl->sumsq[gw + ST_NG * h] = acc;
                    }
                    norm_epoch++;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_fetch_add(&ctl->norm_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    unsigned spins = 0;
                    while (st_lds_load(&ctl->norm_count) < ST_NG * norm_epoch) {
                        if (++spins > ST_SPIN_MAX || st_lds_load(&ctl->failed)) { give_up(); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    double tot = 0;
#pragma unroll
                    for (int w = 0; w < ST_NW; w++) tot += ctl->sumsq[w];
                    const float mean = (float) (tot / (double) K);
                    const float scale = 1.0f / sqrtf(mean + st_uniform(d->eps));
#pragma unroll
                    for (int i = 0; i < XB; i++) {
                        const float a4[4] = { al[i].x, al[i].y, al[i].z, al[i].w };
#pragma unroll
                        for (int k = 0; k < 4; k++) v[i][k] = a4[k] * (v[i][k] * scale);
                    }
                }
#pragma unroll
                for (int i = 0; i < XB; i++) {
                    const int b = gw + ST_NG * (i0 + i);
                    if (b < nb) quantize_block_q8k(xcur + b, v[i], lane);
                }
            };
            if (nb <= 4 * ST_NG) chunk(std::integral_constant<int, 4>(), 0);
            else for (int i0 = 0; gw + ST_NG * i0 < nb; i0 += 6) chunk(std::integral_constant<int, 6>(), i0);
            ST_STAMP(7);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&ctl->blocks_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (st_lds_load(&ctl->failed)) return;
        }
        return;
    }

    // ===================================================== streamers ===========================================================================
    // ---- the ring: slot s holds the header (d, dmin, 6-bit scales) and this lane's 16-byte nibble chunk of one super-block
    u32x4 wh[2 * ST_RH], wq[2 * ST_RH];
    // Wave w owns rows [w rows / 8, (w + 1) rows / 8) of the workgroup's rows (paired: of both halves): every super-block of a row is its own, so the
    // row sums need no workgroup barrier. Its super-blocks, half by half and row by row, are dealt eight to a pass.
    struct wave_rows { int r_lo, nrow, nseg, nsb; };
    auto rows_of = [&](int rows, int nb, bool paired) {
        wave_rows r;
        r.r_lo = (wave * rows) >> 3; r.nrow = (((wave + 1) * rows) >> 3) - r.r_lo; r.nseg = r.nrow * nb; r.nsb = paired ? 2 * r.nseg : r.nseg;
        return r;
    };
    // request pass (round R, slot i of the round) into ring slot H * 4 + i. Beyond the last round every lane asks for the same 16 bytes (one request, unused).
    auto request_round = [&](int R, auto half_tag) {
        constexpr int H = decltype(half_tag)::value;
        const bool real = R < P.n_rounds;
        const int2 ri = rnd[real ? R : 0];
        const int q = __builtin_amdgcn_readfirstlane(ri.x), j0 = __builtin_amdgcn_readfirstlane(ri.y);
        const st_phase * d = phl + q;
        const char * w = st_uniform(d->w);
        const long long row_bytes = st_uniform(d->row_bytes), pair_F = st_uniform(d->pair_F);
        const int nb = st_uniform(d->nb), rows = st_uniform(d->rows);
        const float inv_nb = st_uniform(d->inv_nb);
        const wave_rows wr = rows_of(rows, nb, pair_F > 0);
        const long long row0 = (long long) wg * rows + wr.r_lo;
#pragma unroll
        for (int i = 0; i < ST_RH; i++) {
            const int sb = (j0 + i) * 8 + (lane >> 3);
            const bool use = real && sb < wr.nsb;
            const int sbc = sb < wr.nsb ? sb : wr.nsb - 1;
            const int up = sbc >= wr.nseg ? 1 : 0;                   // second half of a paired phase
            const int sl = sbc - up * wr.nseg;
            const int r = (int) (((float) sl + 0.5f) * inv_nb), b = sl - r * nb;
            const GLOBAL_AS u32x4 * src = (const GLOBAL_AS u32x4 *) (gp(w) + (row0 + r + (up ? pair_F : 0ll)) * row_bytes) + b * 9;
            const GLOBAL_AS u32x4 * hsrc = use ? src : (const GLOBAL_AS u32x4 *) gp(w);
            const GLOBAL_AS u32x4 * qsrc = use ? src + 1 + (lane & 7) : (const GLOBAL_AS u32x4 *) gp(w);
            wh[H * ST_RH + i] = __builtin_nontemporal_load(hsrc);
            wq[H * ST_RH + i] = __builtin_nontemporal_load(qsrc);
        }
    };
    typedef std::integral_constant<int, 0> half0;
    typedef std::integral_constant<int, 1> half1;

    request_round(0, half0());
    request_round(1, half1());

    float * wpart = part + wave * ST_SBW;
    int R = 0;   // rounds consumed so far (whole launch)
    for (int p = 0; p < P.n_phases; p++) {
        const st_phase * d = phl + p;
        const int nb = st_uniform(d->nb), rows = st_uniform(d->rows);
        const long long pair_F = st_uniform(d->pair_F);
        const bool paired = pair_F > 0;
        const unsigned tag_out = tag_base | (unsigned) (p + 1);
        const xblk * xcur = xs + (p & 1) * ST_NB_MAX;
        const int res = st_uniform(d->res), n_pub = st_uniform(d->n_pub);
        float * y = st_uniform(d->y);
        const wave_rows wr = rows_of(rows, nb, paired);
        const long long row0 = (long long) wg * rows + wr.r_lo;
        const float inv_nb = st_uniform(d->inv_nb);
        ST_STAMP(0);
        // a residual that comes from memory is asked for now (its round trip used to sit, exposed, at the very end of the phase): lane group g of
        // sweep k adds it to row 4 k + g
        float res_pre[3];
        {
            const float * rp = res == 2 ? st_uniform(d->residual) : (const float *) y;   // (no branch around a load) y is valid memory of the same extent
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int rr = (lane >> 4) + 4 * k;
                res_pre[k] = gp(rp)[row0 + (rr < wr.nrow ? rr : 0)];
            }
        }
        // ---- wait for this phase's Q8_K blocks (the gatherers write them)
        {
            unsigned spins = 0;
            while (st_lds_load(&ctl->blocks_ready) < (unsigned) (ST_NG * (p + 1))) {
                if (++spins > ST_SPIN_MAX || st_lds_load(&ctl->failed)) { give_up(); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (st_lds_load(&ctl->failed)) return;
        }
        ST_STAMP(1);

        // ---- the dots: rounds of four passes out of one half of the ring; a slot is asked for again (two rounds ahead) as soon as it has been used
        const int n_rounds_p = st_uniform(d->n_pass) / ST_RH;
        auto consume = [&](const u32x4 whs, const u32x4 wqs, int pass) {
            if (pass * 8 >= wr.nsb) return;   // (uniform) a padding pass
            const int sb = pass * 8 + (lane >> 3);
            const int sbc = sb < wr.nsb ? sb : wr.nsb - 1;
            const int sl = sbc >= wr.nseg ? sbc - wr.nseg : sbc;
            const int r = (int) (((float) sl + 0.5f) * inv_nb), b = sl - r * nb;
            const int j8 = lane & 7, g32 = j8 >> 1, hf = j8 & 1;
            const xblk * xb = xcur + b;
            const uint32_t hw[4] = { whs.x, whs.y, whs.z, whs.w };
            uint32_t sc[2], mn[2];
            q4k_unpack_scales_w(hw[1], hw[2], hw[3], sc, mn);
            const u32x4 ylo = *(const u32x4 *) (xb->q + 64 * g32 + 16 * hf), yhi = *(const u32x4 *) (xb->q + 64 * g32 + 32 + 16 * hf);
            const uint32_t qw[4] = { wqs.x, wqs.y, wqs.z, wqs.w }, yl[4] = { ylo.x, ylo.y, ylo.z, ylo.w }, yh[4] = { yhi.x, yhi.y, yhi.z, yhi.w };
            int lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo = dot4_i8((int) (qw[k] & 0x0F0F0F0Fu), (int) yl[k], lo);
                hi = dot4_i8((int) ((qw[k] >> 4) & 0x0F0F0F0Fu), (int) yh[k], hi);
            }
            const int i0 = 2 * g32, i1 = 2 * g32 + 1;
            const int s0 = (int) ((sc[i0 >> 2] >> (8 * (i0 & 3))) & 0xff), s1 = (int) ((sc[i1 >> 2] >> (8 * (i1 & 3))) & 0xff);
            int isum = __mul24(s0, lo) + __mul24(s1, hi);
            const uint32_t bs2 = *(const uint32_t *) (xb->bsums + 2 * j8);
            const int bs = (int) (int16_t) (bs2 & 0xffff) + (int) (int16_t) (bs2 >> 16);
            int msum = __mul24((int) ((mn[j8 >> 2] >> (8 * (j8 & 3))) & 0xff), bs);
            isum += dpp_i32<DPP_QUAD_XOR1>(isum); msum += dpp_i32<DPP_QUAD_XOR1>(msum);
            isum += dpp_i32<DPP_QUAD_XOR2>(isum); msum += dpp_i32<DPP_QUAD_XOR2>(msum);
            isum += dpp_i32<DPP_HALF_MIRROR>(isum); msum += dpp_i32<DPP_HALF_MIRROR>(msum);
            if (j8 == 0 && sb < wr.nsb) {
                const float dd = h2f((uint16_t) (hw[0] & 0xffff)) * xb->d, dmin = h2f((uint16_t) (hw[0] >> 16)) * xb->d;
                wpart[sb] = dd * (float) isum - dmin * (float) msum;
            }
        };
        auto round = [&](int r, auto half_tag) {
            constexpr int H = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < ST_RH; i++) consume(wh[H * ST_RH + i], wq[H * ST_RH + i], r * ST_RH + i);
            request_round(R + 2, half_tag);
            R++;
        };
        {
            int r = 0;
            if (R & 1) {
                for (; r + 2 <= n_rounds_p; r += 2) { round(r, half1()); round(r + 1, half0()); }
                if (r < n_rounds_p) round(r, half1());
            } else {
                for (; r + 2 <= n_rounds_p; r += 2) { round(r, half0()); round(r + 1, half1()); }
                if (r < n_rounds_p) round(r, half0());
            }
        }
        ST_STAMP(2);
        // the wave's own partial sums: LDS operations of one wave execute in order; the fence only keeps the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- fixed-order row sums (16 lanes per row, strided partials, butterfly - matvec_q4k_kernel's), epilogue, publication
        u64 * pub = P.gbuf + (size_t) (p & 1) * ST_K_MAX;
        float * ysv = ysave + (p * ST_NW + wave) * ST_RW_MAX;
        const float * ysrc = ysave + (st_uniform(d->res_src) * ST_NW + wave) * ST_RW_MAX;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int rr = (lane >> 4) + 4 * k;
            if (4 * k >= wr.nrow) break;   // (uniform)
            const int rc = rr < wr.nrow ? rr : wr.nrow - 1;
            float sl = 0.f, sr = 0.f;
            for (int j = lane & 15; j < nb; j += 16) { sl += wpart[rc * nb + j]; if (paired) sr += wpart[wr.nseg + rc * nb + j]; }
            sl = row16_allsum_f32(sl);
            if (paired) sr = row16_allsum_f32(sr);
            if ((lane & 15) == 0 && rr < wr.nrow) {
                const long long row = row0 + rr;
                float out;
                if (paired) out = (sl / (1.0f + expf(-sl))) * sr;
                else {
                    out = sl;
                    if (res == 1) out = ysrc[rr] + out;
                    else if (res == 2) out = res_pre[k] + out;
                    ysv[rr] = out;
                }
                if (n_pub) st_publish(pub + row, tag_out, __float_as_uint(out));
                gp(y)[row] = out;
            }
        }
        // (the gatherers start sweeping the next phase's granules when their own workgroup's streamers are through)
        if (lane == 0) __hip_atomic_fetch_add(&ctl->pub_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ST_STAMP(3);
    }
    if (wg == 0 && tid == 0) *gp(P.launch_seq) = launch + 1u;
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
static int st_env(const char * name, int def) { const char * v = getenv(name); return v ? atoi(v) : def; }
bool k_stream_default_on() { static const int on = st_env("MI355X_STREAM", 0); return on != 0; }

struct stream_plan {
    st_params P;
    size_t smem;
    std::vector<st_phase> phases;
    std::vector<int2> rounds;
};

static bool st_overlaps(const void * a, size_t an, const void * b, size_t bn) {
    return a && b && (const char *) a < (const char *) b + bn && (const char *) b < (const char *) a + an;
}

// Greedy prefix of the n consecutive mat-vecs (launch order) one stream launch can execute; fills the descriptors. 0: none.
#define ST_WHY(code) do { if (st_env("MI355X_STREAM_DEBUG", 0)) fprintf(stderr, "stream_analyse: run of %d cut at phase %d, rule %d (K %lld M %lld pro %d pair %lld)\n", n, i, code, (long long) a.K, (long long) a.M, a.prologue, (long long) a.pair_F); } while (0)
static int stream_analyse(const mv_args * mv, int n, std::vector<st_phase> & out, std::vector<int2> & rounds) {
    out.clear(); rounds.clear();
    int n_rounds = 0;
    static const int max_len = st_env("MI355X_STREAM_MAX", ST_PH_MAX), min_pass = st_env("MI355X_STREAM_MIN_PASSES", 2);
    for (int i = 0; i < n && i < ST_PH_MAX && i < max_len; i++) {
        const mv_args & a = mv[i];
        if (a.wtype != GGML_TYPE_Q4_K || a.ncols != 1 || a.K % 256 != 0 || a.K / 256 > ST_NB_MAX || a.row_bytes != (a.K / 256) * 144) { ST_WHY(1); break; }
        if (a.prologue != MV_PLAIN && a.prologue != MV_RMSNORM) { ST_WHY(2); break; }
        if (a.prologue == MV_RMSNORM && (a.K > 4096 || a.beta)) { ST_WHY(3); break; }
        if (a.out_act || a.out_scale || a.res_embed.table || a.ticket || a.argmax_out[0] || a.argmax_out[1] || a.attn || a.x_out) { ST_WHY(4); break; }
        const int nb = (int) (a.K / 256);
        const int64_t rows_total = a.pair_F > 0 ? a.pair_F : a.M;
        if (a.pair_F > 0 && (a.M != 2 * a.pair_F || a.residual)) { ST_WHY(5); break; }
        if (rows_total % ST_GRID != 0) { ST_WHY(6); break; }
        const int rows = (int) (rows_total / ST_GRID);
        const int rows_w = (rows + ST_NW - 1) / ST_NW;                              // most rows a streamer wave owns
        const int passes = (rows_w * nb * (a.pair_F > 0 ? 2 : 1) + 7) / 8;          // its passes (8 super-blocks each)
        if (rows < ST_NW || rows_w > ST_RW_MAX || passes > ST_PASS_MAX || passes < min_pass) { ST_WHY(7); break; }   // (anything smaller is the chain engine's)
        st_phase ph;
        memset(&ph, 0, sizeof(ph));
        ph.w = a.w; ph.row_bytes = a.row_bytes; ph.x = a.x; ph.alpha = a.alpha; ph.residual = a.residual; ph.y = a.y; ph.pair_F = a.pair_F;
        ph.K = (int) a.K; ph.nb = nb; ph.rows = rows; ph.rows_prev = i > 0 ? out[(size_t) i - 1].rows : 0; ph.prologue = a.prologue; ph.eps = a.eps; ph.inv_nb = 1.0f / (float) nb;
        const size_t ybytes = (size_t) rows_total * 4, xbytes = (size_t) a.K * 4;
        // what this phase reads against what earlier phases of the run write
        ph.x_chain = 0; ph.res = a.residual ? 2 : 0; ph.res_src = 0;
        bool cut = false;
        for (int q = 0; q < i && !cut; q++) {
            const st_phase & e = out[(size_t) q];
            const size_t eb = (size_t) (e.pair_F > 0 ? e.pair_F : (long long) e.rows * ST_GRID) * 4;
            if (st_overlaps(a.x, xbytes, e.y, eb)) {
                if (q == i - 1 && a.x == e.y && xbytes == eb) ph.x_chain = 1; else cut = true;
            }
            if (a.residual && st_overlaps(a.residual, ybytes, e.y, eb)) {
                if (a.residual == e.y && ybytes == eb && e.pair_F == 0 && e.rows == rows) { ph.res = 1; ph.res_src = q; } else cut = true;
            }
            if (a.alpha && st_overlaps(a.alpha, xbytes, e.y, eb)) cut = true;
            // what this phase writes against what earlier phases read from memory (they are done with it: a phase starts after its predecessor's
            // outputs, which every workgroup contributes to) and against their outputs (a later graph node may read those)
            if (st_overlaps(a.y, ybytes, e.y, eb)) cut = true;
        }
        if (cut) { ST_WHY(9); break; }
        if (i > 0 && !ph.x_chain) { ST_WHY(10); break; }   // every phase but the first waits on its predecessor: the two hand-off buffers alternate by phase parity
        // this phase's own operands: y may be its residual in place (row-wise, same thread), nothing else
        if (st_overlaps(a.y, ybytes, a.x, xbytes) || (a.alpha && st_overlaps(a.y, ybytes, a.alpha, xbytes))) { ST_WHY(11); break; }
        if (a.residual && a.residual != a.y && st_overlaps(a.y, ybytes, a.residual, ybytes)) { ST_WHY(12); break; }
        if (st_overlaps(a.y, ybytes, a.w, (size_t) a.M * a.row_bytes)) { ST_WHY(13); break; }
        ph.n_pass = (passes + ST_RH - 1) / ST_RH * ST_RH;
        if (n_rounds + ph.n_pass / ST_RH > ST_RND_MAX) { ST_WHY(14); break; }
        for (int r = 0; r < ph.n_pass / ST_RH; r++) rounds.push_back(make_int2(i, r * ST_RH));
        n_rounds += ph.n_pass / ST_RH;
        ph.n_pub = 0;
        if (i > 0) out[(size_t) i - 1].n_pub = (int) a.K;
        out.push_back(ph);
    }
    return (int) out.size();
}

int k_stream_accept(const mv_args * mv, int n) {
    static const int min_len = st_env("MI355X_STREAM_MIN", 2);
    if (n < min_len) return 0;
    std::vector<st_phase> ph; std::vector<int2> rd;
    const int len = stream_analyse(mv, n, ph, rd);
    return len >= min_len ? len : 0;
}
static size_t st_tables_bytes() { return GGML_PAD((size_t) ST_PH_MAX * sizeof(st_phase), 256) + GGML_PAD((size_t) ST_RND_MAX * sizeof(int2), 256); }
static size_t st_state_bytes() { return 256 + 2 * (size_t) ST_K_MAX * 8 + 2 * (size_t) ST_GRID * 8; }
size_t k_stream_ws_size(const mv_args *, int) { return st_tables_bytes() + st_state_bytes(); }

stream_plan * k_stream_create(hipStream_t s, const mv_args * mv, int n, void * ws, unsigned * err) {
    stream_plan * c = new stream_plan;
    const int len = stream_analyse(mv, n, c->phases, c->rounds);
    GGML_ASSERT(len == n && "k_stream_create: pass exactly the run k_stream_accept took");
    char * base = (char *) ws;
    st_phase * d_ph = (st_phase *) base;
    int2 * d_rd = (int2 *) (base + GGML_PAD((size_t) ST_PH_MAX * sizeof(st_phase), 256));
    char * state = base + st_tables_bytes();
    HIP_CHECK(hipMemcpyAsync(d_ph, c->phases.data(), (size_t) n * sizeof(st_phase), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_rd, c->rounds.data(), c->rounds.size() * sizeof(int2), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(state, 0, st_state_bytes(), s));
    HIP_CHECK(hipStreamSynchronize(s));
    c->P.phases = d_ph; c->P.rounds = d_rd; c->P.n_phases = n; c->P.n_rounds = (int) c->rounds.size();
    c->P.launch_seq = (unsigned *) state;
    c->P.gbuf = (u64 *) (state + 256);
    c->P.flags = c->P.gbuf + 2 * ST_K_MAX;
    c->P.err = err;
    c->smem = 2 * ST_NB_MAX * XBLK_BYTES + (size_t) (ST_NW * ST_SBW + ST_PH_MAX * ST_NW * ST_RW_MAX) * 4 + sizeof(st_ctl) + ST_PH_MAX * sizeof(st_phase) + (ST_RND_MAX + 2) * sizeof(int2);
    GGML_ASSERT(c->smem <= 64 * 1024);
    return c;
}
void k_stream_free(stream_plan * c) { delete c; }
int k_stream_length(const stream_plan * c) { return c->P.n_phases; }
int64_t k_stream_weight_bytes(const stream_plan * c) {
    int64_t b = 0;
    for (auto & ph : c->phases) b += (int64_t) ph.rows * ST_GRID * (ph.pair_F > 0 ? 2 : 1) * ph.row_bytes;
    return b;
}
void k_stream_launch(hipStream_t s, const stream_plan * c) { matvec_stream_kernel<<<ST_GRID, ST_THREADS, c->smem, s>>>(c->P); }
