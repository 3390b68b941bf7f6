// hip_stream.hip — the persistent stream engine: the run of LARGE Q4_K mat-vecs between two attention launches of the Temporal transformer
// (/root/reference/src/moshi/modules/transformer.h:300-420, StreamingTransformerLayer: out_proj + residual -> norm2 + gated linear_in ->
// linear_out + residual -> next layer's norm1 + in_proj; 116 MB of weights per layer at 4096 / 11264) executed by ONE launch.
//
// Why: as four launches a layer's mat-vecs take 5.6 + 14 + 9.7 + 10 us in the kernel (1.7 - 3.7 TB/s, profiles/r03_bench_kernel_trace_summary.txt)
// plus four boundaries of 1.5 - 2.3 us: every launch ramps its HBM stream up from nothing and drains it before the next one may start, although
// the NEXT matrix's bytes never depended on anything. Here 256 workgroups of 8 waves stay resident across the run and every wave keeps a RING of
// eight passes (8 x 1 152 B per wave, 72 KB per CU) of weight requests in flight that runs straight THROUGH the phase boundaries: while a phase's
// rows are summed, published and the next activation vector is gathered, normed and quantised, the next matrix is already on its way.
//   * work unit = super-block (144 B), 8 lanes per super-block, a pass = 8 super-blocks per wave (the WS = 1 arithmetic of matvec_q4k_kernel);
//     workgroup g owns rows [g M / 256, (g + 1) M / 256) of every matrix (paired gate: the same rows of both halves), its super-blocks are dealt to
//     the waves pass by pass; the ring is two halves of four passes, refilled slot by slot right after a slot is consumed;
//   * hand-offs are the chain engine's (hip_chain.hip): every output float is published as one 8-byte {tag, value} granule by an agent-scope store,
//     consumers poll the granules they need until the tags are this launch's and this phase's - no grid barrier, no fence, no counter;
//   * arithmetic is the unchained kernels' to the bit (Q8_K rounding, per-super-block float expression, 16-lane strided row sums, the RMS norm's
//     summation order), so tests/test_stream_engine.py compares the two plans bit for bit.
#include "hip_common.h"
#include "hip_device.h"
#include "hip_mv_device.h"

#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef unsigned long long u64;

#define ST_NW        8
#define ST_THREADS   (ST_NW * 64)
#define ST_GRID      256
#define ST_NB_MAX    44                   // K <= 11264
#define ST_K_MAX     (ST_NB_MAX * 256)
#define ST_SB_MAX    1408                 // super-blocks per workgroup and phase (linear_in: 2 x 44 rows x 16)
#define ST_ROWS_MAX  96                   // rows per workgroup and phase (both halves of a paired phase)
#define ST_PH_MAX    8                    // phases per launch
#define ST_RND_MAX   64                   // rounds (4 passes per wave) per launch
#define ST_SPIN_MAX  (1u << 22)

struct st_phase {
    const char * w; long long row_bytes;
    const float * x; const float * alpha; const float * residual; float * y;
    long long pair_F;            // > 0: paired gate - rows [0, F) and [F, 2 F) of W, y receives silu(l) * r (F values)
    int K, nb, rows, nsb;        // rows: per workgroup (paired: per half); nsb: super-blocks per workgroup (both halves)
    int prologue, x_chain, res, res_src;   // res: 0 none, 1 rows kept in LDS by phase res_src, 2 memory
    int n_pub, n_pass; float eps, inv_nb;
    int pad_[6];
};
static_assert(sizeof(st_phase) % 16 == 0, "descriptors are copied to LDS by 16-byte lanes");

struct st_params {
    const st_phase * phases; const int2 * rounds;   // rounds[r] = (phase, first pass of the round inside the phase)
    int n_phases, n_rounds;
    u64 * gbuf;                 // [2][ST_K_MAX] granules
    unsigned * launch_seq;
    unsigned * err;
};

struct st_ctl { unsigned failed; unsigned pad[3]; double sumsq[ST_NW]; };

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
__device__ u64 g_st_log[2][64][8];   // [0] workgroup 0, [1] last workgroup; wave 0: per phase: 0 prologue start, 1 x ready, 2 blocks ready, 3 dots done, 4 epilogue done
#define ST_STAMP(i) do { if (wave == 0 && lane == 0 && (wg == 0 || wg == (int) gridDim.x - 1) && p < 64) { u64 t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
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
    float * part = (float *) (xs + 2 * ST_NB_MAX);              // [ST_SB_MAX] super-block partial sums of the current phase
    float * ysave = part + ST_SB_MAX;                           // [ST_PH_MAX][ST_ROWS_MAX] this workgroup's output rows, per phase (residuals of later phases)
    st_ctl * ctl = (st_ctl *) (ysave + ST_PH_MAX * ST_ROWS_MAX);
    st_phase * phl = (st_phase *) (ctl + 1);                    // the descriptors
    int2 * rnd = (int2 *) (phl + ST_PH_MAX);                    // [n_rounds + 2]: (phase, first pass)

    {   // tables -> LDS
        const int n16 = P.n_phases * (int) (sizeof(st_phase) / 16);
        for (int i = tid; i < n16; i += ST_THREADS) ((u32x4 *) phl)[i] = ((const GLOBAL_AS u32x4 *) P.phases)[i];
        for (int i = tid; i < P.n_rounds; i += ST_THREADS) ((u64 *) rnd)[i] = ((const GLOBAL_AS u64 *) P.rounds)[i];
        if (tid == 0) ctl->failed = 0;
    }
    __syncthreads();
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    const __amdgpu_buffer_rsrc_t gb = st_rsrc(P.gbuf, 2u * ST_K_MAX * 8u);
    auto give_up = [&]() { if (lane == 0) { __hip_atomic_store(&ctl->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); *gp(P.err) = 3u; } };
    auto barrier_ok = [&]() { lds_barrier(); return st_lds_load(&ctl->failed) == 0u; };

    // ---- the ring: slot s holds the header (d, dmin, 6-bit scales) and this lane's 16-byte nibble chunk of one super-block
    u32x4 wh[8], wq[8];
    // request pass (round R, slot i of the round) into ring slot H * 4 + i. Beyond the last round every lane asks for the same 16 bytes (one request, unused).
    auto request_round = [&](int R, auto half_tag) {
        constexpr int H = decltype(half_tag)::value;
        const bool real = R < P.n_rounds;
        const int2 ri = rnd[real ? R : 0];
        const int q = __builtin_amdgcn_readfirstlane(ri.x), j0 = __builtin_amdgcn_readfirstlane(ri.y);
        const st_phase * d = phl + q;
        const char * w = st_uniform(d->w);
        const long long row_bytes = st_uniform(d->row_bytes), pair_F = st_uniform(d->pair_F);
        const int nb = st_uniform(d->nb), rows = st_uniform(d->rows), nsb = st_uniform(d->nsb);
        const float inv_nb = st_uniform(d->inv_nb);
        const long long row0 = (long long) wg * rows;
        const int nseg = rows * nb;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sb = ((j0 + i) * ST_NW + wave) * 8 + (lane >> 3);
            const bool use = real && sb < nsb;
            const int sbc = sb < nsb ? sb : nsb - 1;
            const int up = sbc >= nseg ? 1 : 0;                      // second half of a paired phase
            const int sl = sbc - up * nseg;
            const int r = (int) (((float) sl + 0.5f) * inv_nb), b = sl - r * nb;
            const GLOBAL_AS u32x4 * src = (const GLOBAL_AS u32x4 *) (gp(w) + (row0 + r + (up ? pair_F : 0ll)) * row_bytes) + b * 9;
            const GLOBAL_AS u32x4 * hsrc = use ? src : (const GLOBAL_AS u32x4 *) gp(w);
            const GLOBAL_AS u32x4 * qsrc = use ? src + 1 + (lane & 7) : (const GLOBAL_AS u32x4 *) gp(w);
            wh[H * 4 + i] = __builtin_nontemporal_load(hsrc);
            wq[H * 4 + i] = __builtin_nontemporal_load(qsrc);
        }
    };
    typedef std::integral_constant<int, 0> half0;
    typedef std::integral_constant<int, 1> half1;

    request_round(0, half0());
    request_round(1, half1());

    int R = 0;   // rounds consumed so far (whole launch)
    bool alive = true;
    for (int p = 0; p < P.n_phases && alive; p++) {
        const st_phase * d = phl + p;
        const int nb = st_uniform(d->nb), K = st_uniform(d->K), rows = st_uniform(d->rows), nsb = st_uniform(d->nsb);
        const long long pair_F = st_uniform(d->pair_F);
        const bool paired = pair_F > 0;
        const int prologue = st_uniform(d->prologue), x_chain = st_uniform(d->x_chain);
        const unsigned tag_in = tag_base | (unsigned) p, tag_out = tag_base | (unsigned) (p + 1);
        xblk * xcur = xs + (p & 1) * ST_NB_MAX;
        const int res = st_uniform(d->res), n_pub = st_uniform(d->n_pub);
        float * y = st_uniform(d->y);
        const long long row0 = (long long) wg * rows;
        ST_STAMP(0);
        // a residual that comes from memory is asked for now (its round trip used to sit, exposed, at the very end of the phase)
        float res_pre[3];
        {
            const float * rp = res == 2 ? st_uniform(d->residual) : (const float *) y;   // (no branch around a load) y is valid memory of the same extent
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int rr = (tid >> 4) + k * (ST_NW * 4);
                res_pre[k] = gp(rp)[row0 + (rr < rows ? rr : 0)];
            }
        }

        // ---- prologue: the activation vector -> Q8_K blocks. Wave w takes blocks w, w + 8, ... (matvec_q4k_kernel's assignment: the norm's sums match).
        auto gather = [&](auto xj_tag) -> bool {
            constexpr int XJ = decltype(xj_tag)::value;
            float v[XJ][4];
            f32x4 al[XJ];
            const float * xp = st_uniform(d->x); const float * ap = st_uniform(d->alpha);
#pragma unroll
            for (int j = 0; j < XJ; j++) {
                const int b = j * ST_NW + wave, bc = b < nb ? b : nb - 1;
                al[j] = (f32x4) { 1.f, 1.f, 1.f, 1.f };
                if (prologue == MV_RMSNORM) al[j] = *(const GLOBAL_AS f32x4 *) (gp(ap) + bc * 256 + lane * 4);
            }
            if (x_chain) {
                const unsigned in_base = (unsigned) ((p - 1) & 1) * (ST_K_MAX * 8u);
                u32x4 g[XJ][2];
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int j = 0; j < XJ; j++) {
                        const int b = j * ST_NW + wave, bc = b < nb ? b : nb - 1;
                        const unsigned o = in_base + ((unsigned) bc * 256u + (unsigned) lane * 4u) * 8u;
                        g[j][0] = st_ld16_agent(gb, o); g[j][1] = st_ld16_agent(gb, o + 16u);
                    }
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < XJ; j++) ok = ok && g[j][0].y == tag_in && g[j][0].w == tag_in && g[j][1].y == tag_in && g[j][1].w == tag_in;
                    if (__all(ok)) break;
                    if (++spins > ST_SPIN_MAX || st_lds_load(&ctl->failed)) { give_up(); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int j = 0; j < XJ; j++) {
                    v[j][0] = __uint_as_float(g[j][0].x); v[j][1] = __uint_as_float(g[j][0].z); v[j][2] = __uint_as_float(g[j][1].x); v[j][3] = __uint_as_float(g[j][1].z);
                }
            } else {
#pragma unroll
                for (int j = 0; j < XJ; j++) {
                    const int b = j * ST_NW + wave, bc = b < nb ? b : nb - 1;
                    const f32x4 t = *(const GLOBAL_AS f32x4 *) (gp(xp) + bc * 256 + lane * 4);
                    v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
                }
            }
            ST_STAMP(1);
            if (prologue == MV_RMSNORM) {
                // matvec_q4k_kernel's order: per-thread squares in double (blocks ascending), wave butterfly, waves added in index order
                double acc = 0;
#pragma unroll
                for (int j = 0; j < XJ; j++)
                    if (j * ST_NW + wave < nb)
#pragma unroll
                        for (int k = 0; k < 4; k++) acc += (double) (v[j][k] * v[j][k]);
                acc = wave_allsum_f64(acc);
                if (lane == 0) ctl->sumsq[wave] = acc;
                if (!barrier_ok()) return false;
                double tot = 0;
#pragma unroll
                for (int w = 0; w < ST_NW; w++) tot += ctl->sumsq[w];
                const float mean = (float) (tot / (double) K);
                const float scale = 1.0f / sqrtf(mean + st_uniform(d->eps));
#pragma unroll
                for (int j = 0; j < XJ; j++) {
                    const float a4[4] = { al[j].x, al[j].y, al[j].z, al[j].w };
#pragma unroll
                    for (int k = 0; k < 4; k++) v[j][k] = a4[k] * (v[j][k] * scale);
                }
            }
#pragma unroll
            for (int j = 0; j < XJ; j++) {
                const int b = j * ST_NW + wave;
                if (b < nb) quantize_block_q8k(xcur + b, v[j], lane);
            }
            return true;
        };
        if (nb <= 2 * ST_NW) { if (!gather(std::integral_constant<int, 2>())) return; }
        else                 { if (!gather(std::integral_constant<int, 6>())) return; }
        if (!barrier_ok()) return;
        ST_STAMP(2);

        // ---- the dots: rounds of four passes out of one half of the ring; a slot is asked for again (two rounds ahead) as soon as it has been used
        const int n_rounds_p = st_uniform(d->n_pass) >> 2;
        auto consume = [&](const u32x4 whs, const u32x4 wqs, int pass) {
            const int sb = (pass * ST_NW + wave) * 8 + (lane >> 3);
            if (pass * (ST_NW * 8) >= nsb) return;   // (uniform) a padding pass
            const int sbc = sb < nsb ? sb : nsb - 1;
            const int nseg = rows * nb;
            const int sl = sbc >= nseg ? sbc - nseg : sbc;
            const int r = (int) (((float) sl + 0.5f) * st_uniform(d->inv_nb)), b = sl - r * nb;
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
            if (j8 == 0 && sb < nsb) {
                const float dd = h2f((uint16_t) (hw[0] & 0xffff)) * xb->d, dmin = h2f((uint16_t) (hw[0] >> 16)) * xb->d;
                part[sb] = dd * (float) isum - dmin * (float) msum;
            }
        };
        auto round = [&](int r, auto half_tag) {
            constexpr int H = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < 4; i++) consume(wh[H * 4 + i], wq[H * 4 + i], r * 4 + i);
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
        ST_STAMP(3);
        if (!barrier_ok()) return;

        // ---- fixed-order row sums (16 lanes per row, strided partials, butterfly), epilogue, publication
        u64 * pub = P.gbuf + (size_t) (p & 1) * ST_K_MAX;
        if (paired) {
            for (int rr = tid >> 4; rr < rows; rr += ST_NW * 4) {
                float sl = 0.f, sr = 0.f;
                for (int j = tid & 15; j < nb; j += 16) { sl += part[rr * nb + j]; sr += part[(rows + rr) * nb + j]; }
                sl = row16_allsum_f32(sl); sr = row16_allsum_f32(sr);
                if ((tid & 15) == 0) {
                    const float g = (sl / (1.0f + expf(-sl))) * sr;
                    if (n_pub) st_publish(pub + row0 + rr, tag_out, __float_as_uint(g));
                    gp(y)[row0 + rr] = g;
                }
            }
        } else {
            const float * ys = ysave + st_uniform(d->res_src) * ST_ROWS_MAX;
            int kq = 0;
            for (int rr = tid >> 4; rr < rows; rr += ST_NW * 4, kq++) {
                float sum = 0.f;
                for (int j = tid & 15; j < nb; j += 16) sum += part[rr * nb + j];
                sum = row16_allsum_f32(sum);
                if ((tid & 15) == 0) {
                    const long long row = row0 + rr;
                    if (res == 1) sum = ys[rr] + sum;
                    else if (res == 2) sum = (kq == 0 ? res_pre[0] : kq == 1 ? res_pre[1] : res_pre[2]) + sum;
                    ysave[p * ST_ROWS_MAX + rr] = sum;
                    if (n_pub) st_publish(pub + row, tag_out, __float_as_uint(sum));
                    gp(y)[row] = sum;
                }
            }
        }
        ST_STAMP(4);
        // (part and ysave are next written behind the next phase's "blocks ready" barrier)
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
    for (int i = 0; i < n && i < ST_PH_MAX; i++) {
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
        const int nsb = rows * nb * (a.pair_F > 0 ? 2 : 1);
        if (nsb > ST_SB_MAX || rows * (a.pair_F > 0 ? 2 : 1) > ST_ROWS_MAX || nsb < ST_NW * 8 * 2) { ST_WHY(7); break; }   // (at least two passes per wave: anything smaller is the chain engine's)
        st_phase ph;
        memset(&ph, 0, sizeof(ph));
        ph.w = a.w; ph.row_bytes = a.row_bytes; ph.x = a.x; ph.alpha = a.alpha; ph.residual = a.residual; ph.y = a.y; ph.pair_F = a.pair_F;
        ph.K = (int) a.K; ph.nb = nb; ph.rows = rows; ph.nsb = nsb; ph.prologue = a.prologue; ph.eps = a.eps; ph.inv_nb = 1.0f / (float) nb;
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
        const int passes = (nsb + ST_NW * 8 - 1) / (ST_NW * 8);
        ph.n_pass = (passes + 3) & ~3;
        if (n_rounds + ph.n_pass / 4 > ST_RND_MAX) { ST_WHY(14); break; }
        for (int r = 0; r < ph.n_pass / 4; r++) rounds.push_back(make_int2(i, r * 4));
        n_rounds += ph.n_pass / 4;
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
static size_t st_state_bytes() { return 256 + 2 * (size_t) ST_K_MAX * 8; }
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
    c->P.err = err;
    c->smem = 2 * ST_NB_MAX * XBLK_BYTES + (size_t) (ST_SB_MAX + ST_PH_MAX * ST_ROWS_MAX) * 4 + sizeof(st_ctl) + ST_PH_MAX * sizeof(st_phase) + (ST_RND_MAX + 2) * sizeof(int2);
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
