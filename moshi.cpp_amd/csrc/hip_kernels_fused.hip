// hip_kernels_fused.hip — the per-frame hot path of moshi.cpp's streaming decode as hand-written gfx950
// kernels: block-quantised mat-vec with fused prologue / epilogue,
// single-token ring-cache attention, and the 17-stream embedding sum.
//
// Roofline: every kernel here is HBM-bound (batch-1 mat-vec = 2 flop per 0.5625 B of Q4_K weights), so the
// design rules are the bandwidth ones: 16-byte fully coalesced loads of the raw block stream, enough
// bytes in flight per CU, weights read exactly once, no materialised intermediates.
//
// Q4_K mat-vec data flow (matvec_q4k_kernel):
//   HBM  --16 B/lane coalesced, 9 loads = one 64-super-block tile (9216 B) per wave-->  VGPR
//   VGPR --ds_write_b128, linear image-->  LDS (wave-private 9216 B)
//   LDS  --lane l reads super-block l (9 x ds_read_b128 at a 144 B stride: conflict-free)--> VGPR
//   activations: quantised once per workgroup to Q8_K in LDS (padded 304 B blocks: conflict-free),
//   integer sub-block dots with v_dot4_i32_i8, one float term per super-block, fixed-order row sums.
// The activation rounding (Q8_K) and the per-super-block arithmetic are ggml's CPU semantics, so results
// agree with the oracle to float-summation-order noise.
#include "hip_common.h"
#include "hip_device.h"
#include "hip_mv_device.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <map>

// ---------------------------------------------------------------------------------------------------
// activation prologues
// ---------------------------------------------------------------------------------------------------
struct x_src {
    int prologue;
    const float * x;
    const float * alpha;
    float scale;      // rms scale (RMSNORM)
    int64_t K;
};

__device__ __forceinline__ float x_value(const x_src & s, int64_t i) {
    switch (s.prologue) {
        case MV_RMSNORM:   return s.alpha[i] * (s.x[i] * s.scale);
        case MV_GATE_SILU: { const float l = s.x[i], r = s.x[s.K + i]; return (l / (1.0f + expf(-l))) * r; }
        default:           return s.x[i];
    }
}

// 1/sqrt(mean(x^2) + eps) over K elements, all threads of the block participate (double accumulation)
__device__ float block_rms_scale(const float * x, int64_t K, float eps, double * sh) {
    double acc = 0;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) { const float v = x[i]; acc += (double) (v * v); }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    double tot = 0;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) tot += sh[w];
    const float mean = (float) (tot / (double) K);
    return 1.0f / sqrtf(mean + eps);
}

// ---------------------------------------------------------------------------------------------------
// Q4_K mat-vec
// ---------------------------------------------------------------------------------------------------
#define TILE_BYTES 9216  // 64 super-blocks of 144 B = 9 wave-wide 16-byte loads

// diagnostic builds only. -DMV_STAMPS (tests/microbench/mv_bench.hip): per-phase s_memtime stamps of wave 0 of each workgroup.
// -DMV_LOG (tests/microbench/frame_stamps.py): one record per LAUNCH, taken by workgroup 0 / wave 0 - phase stamps, start / end in
// s_memrealtime ticks (10 ns), shape - so that a graph-replayed frame shows every mat-vec's in-kernel phases and the gaps between kernels.
#if defined(MV_LOG)
__device__ unsigned long long g_mv_log[8192][24];
__device__ unsigned g_mv_launch;
#define MV_STAMP(i) do { if (blockIdx.x == 0 && tid == 0) { if ((i) == 0) mv_log_id = atomicAdd(&g_mv_launch, 1u) & 8191u; unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mv_log[mv_log_id][i] = t_; \
    if ((i) == 0 || (i) == 7) { unsigned long long r_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory"); g_mv_log[mv_log_id][16 + ((i) == 7)] = r_; } \
    if ((i) == 0) { g_mv_log[mv_log_id][18] = (unsigned long long) a.K | ((unsigned long long) a.M << 32); g_mv_log[mv_log_id][19] = (unsigned long long) PRO | ((unsigned long long) WS << 8) | ((unsigned long long) gridDim.x << 32); } } } while (0)
extern "C" __attribute__((visibility("default"))) int mi355x_mv_log_read(unsigned long long * out, int max_records) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_mv_launch), 4) != hipSuccess) return -1;
    const int m = (int) (n < 8192u ? n : 8192u) < max_records ? (int) (n < 8192u ? n : 8192u) : max_records;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mv_log), (size_t) m * 24 * 8) != hipSuccess) return -1;
    const unsigned z = 0; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_mv_launch), &z, 4);
    return (int) n;
}
#elif defined(MV_STAMPS)
__device__ unsigned long long g_mv_stamps[4096][8];
__device__ unsigned long long g_mv_real[4096][2];
#define MV_STAMP(i) do { if (lane == 0 && wave == 0 && blockIdx.x < 4096) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mv_stamps[blockIdx.x][i] = t_; \
    if ((i) == 0 || (i) == 7) { unsigned long long r_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory"); g_mv_real[blockIdx.x][(i) == 7] = r_; } } } while (0)
#else
#define MV_STAMP(i) do {} while (0)
#endif


// Weight formats of the block mat-vec. A lane always owns 256 consecutive weights of one row: one Q4_K super-block, or eight
// consecutive Q8_0 / Q4_0 blocks (K % 256 == 0 for every linear of the models). Activations are quantised the way ggml's CPU
// backend does for that weight type: Q8_K (256-wide) for Q4_K, Q8_0 (32-wide, F16 scale) for Q8_0 and Q4_0; both kinds are kept
// in the same 304-byte LDS record.
#define MVF_Q4K 0
#define MVF_Q80 1
#define MVF_Q40 2
template <int FMT> struct mvfmt;
template <> struct mvfmt<MVF_Q4K> { static constexpr int SB = 144, NLOAD = 9; };    // f16 d, f16 dmin, 12 B scales, 128 B nibbles
template <> struct mvfmt<MVF_Q80> { static constexpr int SB = 272, NLOAD = 17; };   // 8 x (f16 d, 32 int8)
template <> struct mvfmt<MVF_Q40> { static constexpr int SB = 144, NLOAD = 9; };    // 8 x (f16 d, 16 B nibbles)
// dword `i` of a byte stream that starts 2 bytes into a 4-byte-aligned image (D = the aligned dwords)
__device__ __forceinline__ uint32_t dword_at2(const uint32_t * D, int byte_off) {
    const int i = byte_off >> 2;
    return (byte_off & 2) ? __builtin_amdgcn_alignbyte(D[i + 1], D[i], 2) : D[i];
}
// eight Q8_0 blocks (272 B, 16-byte aligned) against 256 Q8_0-quantised activations: sum_j sumi_j * (d_w * d_x), in block order
// (vec_dot_q8_0_q8_0)
__device__ __forceinline__ float q80_q80_sb_dot(const char * wb, const xblk80 * xb) {
    const uint32_t * D = (const uint32_t *) wb;
    const int * y = (const int *) xb->q;
    float sumf = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int off = 34 * j;
        const float dw = h2f((uint16_t) (D[off >> 2] >> ((off & 2) * 8)));
        int sumi = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) sumi = dot4_i8((int) dword_at2(D, off + 2 + 4 * t), y[j * 8 + t], sumi);
        sumf += (float) sumi * (dw * xb->d[j]);
    }
    return sumf;
}
// eight Q4_0 blocks (144 B) against 256 Q8_0-quantised activations: sum_j (sumi_j * d_w) * d_x with sumi = sum (q - 8) * y
// (vec_dot_q4_0_q8_0); the -8 offset is applied through the per-block activation sums
__device__ __forceinline__ float q40_q80_sb_dot(const char * wb, const xblk80 * xb) {
    const uint32_t * D = (const uint32_t *) wb;
    const int * y = (const int *) xb->q;
    float sumf = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int off = 18 * j;
        const float dw = h2f((uint16_t) (D[off >> 2] >> ((off & 2) * 8)));
        int sumi = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t q = dword_at2(D, off + 2 + 4 * t);
            sumi = dot4_i8((int) (q & 0x0F0F0F0Fu), y[j * 8 + t], sumi);
            sumi = dot4_i8((int) ((q >> 4) & 0x0F0F0F0Fu), y[j * 8 + 4 + t], sumi);
        }
        sumi -= 8 * (int) xb->bsums[j];
        sumf += ((float) sumi * dw) * xb->d[j];
    }
    return sumf;
}


// Attention of NH (2 or 4) consecutive heads by ONE wave, for a single new token over a short ring (C <= 8 slots of D = 64): the Depth
// transformer's shape. Same arithmetic, in the same order, as attn_decode_kernel below restricted to the one wave that has work
// there (lane = (slot, 8-dim chunk)). Every global load of all four heads (q/k/v, ring rows) is requested before the first use, so
// the whole thing costs about one memory round trip. NH x 64 outputs land in xa (LDS); `wbuf` is >= 1792 floats (7 KB) of wave-private LDS.
template <int NH>
__device__ __forceinline__ void attn_small_wave(const attn_args & a, int h0, int lane, float * wbuf, float * xa, bool write_cache) {
    constexpr int LPS = 8, half = 32;
    const int C = a.C;
    const int sub = lane / LPS, dl = (lane % LPS) * 8;
    const int c = sub, cc = c < C ? c : C - 1;
    const int j = lane, p = j < half ? j : j - half;
    // ---- loads
    const int slot = a.index[0];
    const float m = a.mask[cc];
    float rc = 1.f, rs = 0.f;
    if (a.rot) { rc = a.rot[p]; rs = a.rot[half + p]; }
    float qr[NH], qi[NH], kr[NH], ki[NH], vv[NH];
    uint4 kq[NH], vq[NH];
#pragma unroll
    for (int hh = 0; hh < NH; hh++) {
        const int h = h0 + hh;
        const float * q = a.q + (int64_t) h * a.q_hs, * k = a.k + (int64_t) h * a.k_hs, * v = a.v + (int64_t) h * a.v_hs;
        if (a.rot) { qr[hh] = q[2 * p]; qi[hh] = q[2 * p + 1]; kr[hh] = k[2 * p]; ki[hh] = k[2 * p + 1]; }
        else { qr[hh] = q[j]; qi[hh] = 0.f; kr[hh] = k[j]; ki[hh] = 0.f; }
        vv[hh] = v[j];
        kq[hh] = *(const uint4 *) (a.kcache + (int64_t) h * a.k_nb2 + (int64_t) cc * a.k_nb1 + dl * 2);
        vq[hh] = *(const uint4 *) (a.vcache + (int64_t) h * a.v_nb2 + (int64_t) cc * a.v_nb1 + dl * 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool live = c < C && m > -INFINITY;
    const bool fresh = slot == c;
    // ---- RoPE, BF16 rounding, ring write
#pragma unroll
    for (int hh = 0; hh < NH; hh++) {
        float qo, ko;
        if (a.rot) {
            if (j < half) { qo = qr[hh] * rc - qi[hh] * rs; ko = kr[hh] * rc - ki[hh] * rs; }
            else          { qo = qr[hh] * rs + qi[hh] * rc; ko = kr[hh] * rs + ki[hh] * rc; }
        } else { qo = qr[hh]; ko = kr[hh]; }
        const uint16_t kb = f2bf(ko), vb = f2bf(vv[hh]);
        float * wb = wbuf + hh * 192;
        wb[j] = bf2f(f2bf(qo)); wb[64 + j] = bf2f(kb); wb[128 + j] = bf2f(vb);
        if (write_cache && slot >= 0 && slot < C) {
            const int h = h0 + hh;
            ((uint16_t *) (a.kcache + (int64_t) h * a.k_nb2 + (int64_t) slot * a.k_nb1))[j] = kb;
            ((uint16_t *) (a.vcache + (int64_t) h * a.v_nb2 + (int64_t) slot * a.v_nb1))[j] = vb;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // P x V: every lane forms the 8 products of its (slot, 8-dim chunk) - exact in float, both factors being BF16 values - and parks them in
    // LDS as [slot][dim]; lane j then adds the 8 slots of dim j in double, in slot order: the oracle's accumulation order, one LDS round trip
    // per pair of heads (the previous form tree-summed 8 doubles per lane through LDS in two rounds per head).
    float * prod = wbuf + 768;   // 2 heads x 8 slots x 64 dims floats = 4 KB behind the q / k / v scratch
#pragma unroll
    for (int hp = 0; hp < NH; hp += 2) {
#pragma unroll
        for (int hq = 0; hq < 2; hq++) {
            const int hh = hp + hq;
            const float * qf = wbuf + hh * 192, * knew = qf + 64, * vnew = qf + 128;
            float qv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) qv[i] = qf[dl + i];
            double acc = 0;
            if (live) {
                if (fresh) {
#pragma unroll
                    for (int i = 0; i < 8; i++) acc += (double) (knew[dl + i] * qv[i]);
                } else {
                    const uint32_t kw[4] = { kq[hh].x, kq[hh].y, kq[hh].z, kq[hh].w };
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        acc += (double) (bf2f((uint16_t) (kw[i] & 0xffff)) * qv[2 * i]);
                        acc += (double) (bf2f((uint16_t) (kw[i] >> 16)) * qv[2 * i + 1]);
                    }
                }
            }
            acc = group_allsum_f64(acc, LPS);
            const float sv = live ? (float) acc * a.scale + m : -INFINITY;
            const float gmax = wave_allmax_f32(sv);
            const float e = sv > -INFINITY ? expf(sv - gmax) : 0.f;
            double lsum = (lane % LPS) == 0 ? (double) e : 0.0;   // one representative per slot
            lsum = wave_allsum_f64(lsum);
            const float inv = (float) (1.0 / lsum);
            const float pr = bf2f(f2bf(e * inv));
            float pf[8];
#pragma unroll
            for (int i = 0; i < 8; i++) pf[i] = 0.f;
            if (pr != 0.f) {
                if (fresh) {
#pragma unroll
                    for (int i = 0; i < 8; i++) pf[i] = vnew[dl + i] * pr;
                } else {
                    const uint32_t vw[4] = { vq[hh].x, vq[hh].y, vq[hh].z, vq[hh].w };
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        pf[2 * i]     = bf2f((uint16_t) (vw[i] & 0xffff)) * pr;
                        pf[2 * i + 1] = bf2f((uint16_t) (vw[i] >> 16)) * pr;
                    }
                }
            }
            float * dst = prod + hq * 512 + sub * 64 + dl;
            *(float4 *) dst = make_float4(pf[0], pf[1], pf[2], pf[3]);
            *(float4 *) (dst + 4) = make_float4(pf[4], pf[5], pf[6], pf[7]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int hq = 0; hq < 2; hq++) {
            double tot = 0;
#pragma unroll
            for (int c2 = 0; c2 < 8; c2++) tot += (double) prod[hq * 512 + c2 * 64 + lane];
            xa[(hp + hq) * 64 + lane] = (float) tot;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// One workgroup = 4 waves = `rows_per_wg` output rows.
//  phase 1: every wave puts its first weight tile in flight (9 x 16 B per lane, nontemporal)
//  phase 2: the activation vector is produced (prologue) and quantised to Q8_K in LDS; all loads of a
//           16-block batch are issued before any of them is used, so the phase costs ~one L2 round trip
//  phase 3: tiles stream registers -> LDS image -> one super-block per lane (next tile prefetched first)
//  phase 4: fixed-order row sums (+ residual)
//
// WS = 1 (small matrices, Q4_K): no LDS tiles. Eight lanes share a super-block - every lane requests the 16-byte header (d, dmin, 6-bit scales;
// one fetch for the group) and ONE 16-byte nibble chunk, all passes up front - so a wave covers 8 super-blocks per pass with 1/8 of the integer
// work per lane. The Depth mat-vecs are a few tiles in all: with a super-block per lane 192 waves did ~2 000 cycles of serial dot4 / unpack each
// while 3/4 of the chip's SIMDs idled; this way the dot phase is ~300 cycles. Same integers, same per-super-block float expression.
#define MVD_PMAX 4   // passes (super-blocks per lane group) a workgroup can hold in registers: rows * nb <= MVD_PMAX * NW * 8
// XB: activation batches of NW * 256 values a load_batch requests per thread (4: K up to NW * 1024 per round; 2: K <= NW * 512 in ONE round); RN: residual values
// pre-loaded per thread (row (tid >> 4) + k * NW * 4 for k < RN: 4 covers any workgroup, 1 covers rows <= NW * 4, 0 = no residual). Until round 6 every launch
// issued four x + four alpha / gate loads and four residual loads per thread whatever it needed: at K = 4096 with 8 waves (the Temporal linear_in, the text head)
// two of the four batches were clamped duplicates and there is no residual - 8 of a wave's 21 vector-memory instructions, ~26 cycles of issue each, in front of
// the weight tile and of the workgroup barrier of the norm; out_proj / linear_out (16 rows per workgroup) used one of their four residual loads.
template <int PRO, int NW, int FMT = MVF_Q4K, int WS = 0, int XB = 4, int RN = 4>
__global__ void __launch_bounds__(NW * 64) matvec_q4k_kernel(mv_args a, int rows_per_wg, attn_args at) {
    constexpr int SB = mvfmt<FMT>::SB, NLOAD = mvfmt<FMT>::NLOAD, TILE = 64 * SB;   // bytes per lane-chunk / 16-byte loads per lane per tile
    static_assert(WS == 0 || FMT == MVF_Q4K, "the direct layout is written for Q4_K super-blocks");
    auto quantize_block = [&](xblk * dst, const float v[4]) {
        if (FMT == MVF_Q4K) quantize_block_q8k(dst, v, threadIdx.x & 63); else quantize_block_q80((xblk80 *) dst, v, threadIdx.x & 63);
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double sh_red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if defined(MV_LOG)
    unsigned mv_log_id = 0;
#endif
    constexpr int nwaves = NW;
    const int nb = (int) (a.K / 256);
    constexpr int STAGE = (WS == 0 || PRO == MV_ATTN) ? TILE : 0;   // wave-private staging area (weight tiles; scratch of the attention prologue)
    xblk * xs = (xblk *) smem;
    char * stage = smem + nb * XBLK_BYTES + wave * STAGE;
    float * part = (float *) (smem + nb * XBLK_BYTES + nwaves * STAGE);

    // paired gate (a.pair_F): rows_per_wg = 2 h rows, the first h from W_l at [blockIdx * h, +h), the second h from W_r at F + the same range
    const bool paired = WS == 0 && a.pair_F > 0;
    const int half_rows = rows_per_wg >> 1;
    const int64_t row0 = paired ? (int64_t) blockIdx.x * half_rows : (int64_t) blockIdx.x * rows_per_wg;
    const int rows = paired ? rows_per_wg : (int) (a.M - row0 < rows_per_wg ? a.M - row0 : rows_per_wg);
    const int nblk = rows * nb;
    const int ntiles = (nblk + 63) >> 6;
    const int nchunks = nblk * (SB / 16);
    const u32x4 * wsrc = (const u32x4 *) (a.w + row0 * a.row_bytes);
    // (paired: tiles of the second half start half_rows * nb / 64 tiles in - whole tiles, checked on the host - and come from the W_r rows)
    const int tiles_half = paired ? (half_rows * nb) >> 6 : 0x7fffffff;
    const u32x4 * wsrc_r = paired ? (const u32x4 *) (a.w + (a.pair_F + row0) * a.row_bytes) - (int64_t) tiles_half * (NLOAD * 64) : wsrc;

    MV_STAMP(0);
    const int K = (int) a.K;
    // activation loads of the first 16-block batch go out BEFORE the weight tile: vector-memory loads return in order,
    // so this lets the prologue finish while the (much larger, HBM-bound) weight tile is still in flight
    float4 xv[XB], aux[XB];
    bool ok[XB];
    auto load_batch = [&](int base) {
#pragma unroll
        for (int j = 0; j < XB; j++) {
            // loads are unconditional (index clamped, result discarded through ok[]): a branch around a load makes hipcc
            // fall back to s_waitcnt vmcnt(0)-style waits, which would serialise the prologue behind the weight tile
            const int e0 = base + j * (NW * 256) + tid * 4;
            ok[j] = e0 < K;
            const int e = ok[j] ? e0 : K - 4;
            xv[j] = *(const float4 *) (a.x + e);
            aux[j] = xv[j];
            if (PRO == MV_RMSNORM) aux[j] = *(const float4 *) (a.alpha + e);
            if (PRO == MV_GATE_SILU) aux[j] = *(const float4 *) (a.x + K + e);
        }
    };
    // a LATER round (K > XB * NW * 256: the Temporal linear_out, K = 11 264) is waited for at once, so here a wave simply skips the batches past the end of x
    // (whole waves fall on either side: K is a multiple of 256) instead of issuing clamped duplicates - 2 of round two's 4 loads at K = 11 264
    auto load_batch_tail = [&](int base) {
#pragma unroll
        for (int j = 0; j < XB; j++) {
            const int e0 = base + j * (NW * 256) + tid * 4;
            ok[j] = e0 < K;
            xv[j] = make_float4(0.f, 0.f, 0.f, 0.f); aux[j] = xv[j];
            if (__builtin_amdgcn_readfirstlane(e0 - lane * 4) < K) {
                xv[j] = *(const float4 *) (a.x + e0);
                aux[j] = xv[j];
                if (PRO == MV_RMSNORM) aux[j] = *(const float4 *) (a.alpha + e0);
                if (PRO == MV_GATE_SILU) aux[j] = *(const float4 *) (a.x + K + e0);
            }
        }
    };
    if (PRO != MV_PREQ8K && PRO != MV_ATTN) load_batch(0);
    // MV_PREQ8K: the activation arrives quantised (norm_quant / gate_quant kernels): nb padded Q8_K blocks of 304 B = nb * 19 chunks of 16 B,
    // at most 3 per thread (K <= 16384 at 256 threads); requested ahead of the weight tile for the same reason
    constexpr int PQ_MAX = 4;
    u32x4 pq[PQ_MAX];
    if (PRO == MV_PREQ8K) {
        const int nchunk = nb * (XBLK_BYTES / 16);
#pragma unroll
        for (int j = 0; j < PQ_MAX; j++) {
            const int i = tid + j * NW * 64;
            pq[j] = ((const u32x4 *) a.x)[i < nchunk ? i : nchunk - 1];
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the activation loads ahead of the weight tile in program (= return) order
#if defined(MV_HEAD_EXP) && MV_HEAD_EXP == 1
    __builtin_amdgcn_s_barrier();
#elif defined(MV_HEAD_EXP) && MV_HEAD_EXP == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif

    u32x4 r[WS == 0 ? NLOAD : 1];
    u32x4 dh[WS == 1 ? MVD_PMAX : 1], dq[WS == 1 ? MVD_PMAX : 1];   // WS = 1: header / nibble chunk of this lane group's super-block, per pass
    int t = wave;
    constexpr bool first_is_last = false;
    if (WS == 0 && first_is_last) {
    } else
    if (WS == 0) {   // unconditional (no branch around a load, see above): chunks past the end re-read the last valid chunk, and a wave without
        // a tile reads chunk 0 in every lane - one 16-byte request instead of a 9 KB tile in the CU's load queue; neither is consumed
        const bool has_tile = t < ntiles;
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            const int g = t * (NLOAD * 64) + i * 64 + lane;
            r[i] = __builtin_nontemporal_load((t >= tiles_half ? wsrc_r : wsrc) + (has_tile ? (g < nchunks ? g : nchunks - 1) : 0));
        }
    } else {
#pragma unroll
        for (int p = 0; p < MVD_PMAX; p++) {
            const int sb = p * (NW * 8) + wave * 8 + (lane >> 3);
            const u32x4 * src = wsrc + (sb < nblk ? sb : nblk - 1) * 9;
            dh[p] = __builtin_nontemporal_load(src);
            dq[p] = __builtin_nontemporal_load(src + 1 + (lane & 7));
        }
    }
    // the residual of this thread's rows (phase 4) rides behind the first weight loads: its round trip used to sit, exposed, at the very end
    float res_pre[4] = { 0.f, 0.f, 0.f, 0.f };
    if (RN > 0) {
        const float * rp = a.residual ? a.residual : a.y;   // (no branch around a load) y is valid memory of the same extent
#pragma unroll
        for (int k = 0; k < RN; k++) {
            const int rr = (tid >> 4) + k * (NW * 4);
            res_pre[k] = rp[row0 + (rr < (paired ? half_rows : rows) ? rr : 0)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    MV_STAMP(1);

    if (PRO == MV_ATTN) {
        // x = attention output (H heads of 64, K = NW * 256): wave w computes heads 4w..4w+3 = its own Q8_K block, entirely in
        // its (still unused) tile-staging LDS; workgroup 0 also performs the ring write the attention node implies
        // heads are spread over all NW waves (H / NW = 2 or 4 each: the serial arithmetic of one head is ~0.5 us), the outputs meet
        // in a buffer carved from wave 0's staging area, then the first K / 256 waves quantise one block each
        float * wbuf = (float *) stage;
        float * xa_all = part + rows_per_wg * nb;   // 4 KB behind the partial sums (the waves' own scratch fills 7 of their 9 KB)
        const int hpw = at.H / NW;
        if (hpw == 4) attn_small_wave<4>(at, wave * 4, lane, wbuf, xa_all + wave * 256, blockIdx.x == 0);
        else          attn_small_wave<2>(at, wave * 2, lane, wbuf, xa_all + wave * 128, blockIdx.x == 0);
        lds_barrier();
        if (wave < nb) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = xa_all[wave * 256 + lane * 4 + k];
            if (a.x_out != nullptr && blockIdx.x == 0) *(float4 *) (a.x_out + wave * 256 + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
            quantize_block(xs + wave, v);
        }
    } else if (PRO == MV_PREQ8K) {
        // activations were quantised once by norm_quant / gate_quant: the padded Q8_K blocks (304 B each) go to LDS as they are
        const int nchunk = nb * (XBLK_BYTES / 16);
#pragma unroll
        for (int j = 0; j < PQ_MAX; j++) {
            const int i = tid + j * NW * 64;
            if (i < nchunk) ((u32x4 *) xs)[i] = pq[j];
        }
    } else {
        for (int base = 0; base < K; base += NW * 256 * XB) {
            if (base > 0) load_batch_tail(base);
            float v[XB][4];
#pragma unroll
            for (int j = 0; j < XB; j++) {
                const float z = ok[j] ? 1.f : 0.f;   // clamped (out-of-range) chunks contribute nothing to the norm
                v[j][0] = xv[j].x * z; v[j][1] = xv[j].y * z; v[j][2] = xv[j].z * z; v[j][3] = xv[j].w * z;
            }
#if defined(MV_LOG)
            asm volatile("" :: "v"(v[0][0]), "v"(v[XB - 1][3]));   // (diagnostic build: stamp 8 = this wave's x values have arrived)
            MV_STAMP(8);
#endif
            if (PRO == MV_RMSNORM) {   // K <= 4096 (checked on the host): the whole vector is in registers
                double acc = 0;
#pragma unroll
                for (int j = 0; j < XB; j++)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc += (double) (v[j][k] * v[j][k]);
                acc = wave_allsum_f64(acc);
                if (lane == 0) sh_red[wave] = acc;
                __syncthreads();
                MV_STAMP(9);    // (every wave's sum of squares is in)
                double tot = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) tot += sh_red[w];
                const float mean = (float) (tot / (double) K);
                const float scale = 1.0f / sqrtf(mean + a.eps);
#pragma unroll
                for (int j = 0; j < XB; j++) {
                    const float al[4] = { aux[j].x, aux[j].y, aux[j].z, aux[j].w };
#pragma unroll
                    for (int k = 0; k < 4; k++) v[j][k] = al[k] * (v[j][k] * scale);
                }
            }
            if (PRO == MV_GATE_SILU) {
#pragma unroll
                for (int j = 0; j < XB; j++) {
                    const float rr[4] = { aux[j].x, aux[j].y, aux[j].z, aux[j].w };
#pragma unroll
                    for (int k = 0; k < 4; k++) { const float l = v[j][k]; v[j][k] = (l / (1.0f + expf(-l))) * rr[k]; }
                }
            }
#if defined(MV_LOG)
            asm volatile("" :: "v"(v[0][0]), "v"(v[XB - 1][3]));   // (stamp 10 = normalised / gated values ready, the quantiser starts)
            MV_STAMP(10);
#endif
#pragma unroll
            for (int j = 0; j < XB; j++) {
                if (!ok[j]) continue;   // wave-uniform: a wave always holds one whole block. (Passing ok[j] into a branch-free quantiser so that a wave's blocks
                // interleave measured SLOWER: Temporal 1 400 -> 1 415 us, profiles/r06_ab_branch_free_quantiser.txt)
                const int b = base / 256 + j * NW + wave;
                if (a.x_out != nullptr && blockIdx.x == 0) *(float4 *) (a.x_out + base + j * (NW * 256) + tid * 4) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                quantize_block(xs + b, v[j]);
            }
        }
    }
    MV_STAMP(2);
    __syncthreads();
    MV_STAMP(3);

    // phase 3
    if (WS == 1) {
#pragma unroll
        for (int p = 0; p < MVD_PMAX; p++) {
            const int sb = p * (NW * 8) + wave * 8 + (lane >> 3);
            if (p * (NW * 8) >= nblk) break;
            const int j8 = lane & 7, g32 = j8 >> 1, hf = j8 & 1;
            const xblk * xb = xs + ((sb < nblk ? sb : nblk - 1) % nb);
            const uint32_t hw[4] = { dh[p].x, dh[p].y, dh[p].z, dh[p].w };
            uint32_t sc[2], mn[2];
            q4k_unpack_scales_w(hw[1], hw[2], hw[3], sc, mn);
            const u32x4 ylo = *(const u32x4 *) (xb->q + 64 * g32 + 16 * hf), yhi = *(const u32x4 *) (xb->q + 64 * g32 + 32 + 16 * hf);
            const uint32_t qw[4] = { dq[p].x, dq[p].y, dq[p].z, dq[p].w }, yl[4] = { ylo.x, ylo.y, ylo.z, ylo.w }, yh[4] = { yhi.x, yhi.y, yhi.z, yhi.w };
            int lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo = dot4_i8((int) (qw[k] & 0x0F0F0F0Fu), (int) yl[k], lo);
                hi = dot4_i8((int) ((qw[k] >> 4) & 0x0F0F0F0Fu), (int) yh[k], hi);
            }
            const int i0 = 2 * g32, i1 = 2 * g32 + 1;
            const int s0 = (int) ((sc[i0 >> 2] >> (8 * (i0 & 3))) & 0xff), s1 = (int) ((sc[i1 >> 2] >> (8 * (i1 & 3))) & 0xff);
            int isum = __mul24(s0, lo) + __mul24(s1, hi);
            // mins: lane j8 takes sub-block j8
            const uint32_t bs2 = *(const uint32_t *) (xb->bsums + 2 * j8);
            const int bs = (int) (int16_t) (bs2 & 0xffff) + (int) (int16_t) (bs2 >> 16);
            int msum = __mul24((int) ((mn[j8 >> 2] >> (8 * (j8 & 3))) & 0xff), bs);
            isum += dpp_i32<DPP_QUAD_XOR1>(isum); msum += dpp_i32<DPP_QUAD_XOR1>(msum);
            isum += dpp_i32<DPP_QUAD_XOR2>(isum); msum += dpp_i32<DPP_QUAD_XOR2>(msum);
            isum += dpp_i32<DPP_HALF_MIRROR>(isum); msum += dpp_i32<DPP_HALF_MIRROR>(msum);
            if (j8 == 0 && sb < nblk) {
                const float d = h2f((uint16_t) (hw[0] & 0xffff)) * xb->d, dmin = h2f((uint16_t) (hw[0] >> 16)) * xb->d;
                part[sb] = d * (float) isum - dmin * (float) msum;
            }
        }
    }
#define MV_LOOP_MORE true
    for (; WS == 0 && t < ntiles && MV_LOOP_MORE; t += nwaves) {
        {
#pragma unroll
            for (int i = 0; i < NLOAD; i++) ((u32x4 *) stage)[i * 64 + lane] = r[i];
        }
        MV_STAMP(4);
        {
            const int tn = t + nwaves < ntiles ? t + nwaves : ntiles - 1;
#pragma unroll
            for (int i = 0; i < NLOAD; i++) {
                const int g = tn * (NLOAD * 64) + i * 64 + lane;
                r[i] = __builtin_nontemporal_load((tn >= tiles_half ? wsrc_r : wsrc) + (g < nchunks ? g : nchunks - 1));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int bi = t * 64 + lane;
        if (bi < nblk) {
            const char * wb = stage + lane * SB;
            const xblk * xb = xs + (bi % nb);
            if (FMT == MVF_Q4K) part[bi] = q4k_q8k_block_dot((const block_q4_K *) wb, xb->q, xb->bsums, xb->d);
            else if (FMT == MVF_Q80) part[bi] = q80_q80_sb_dot(wb, (const xblk80 *) xb);
            else part[bi] = q40_q80_sb_dot(wb, (const xblk80 *) xb);
        }
        __builtin_amdgcn_wave_barrier();
    }
    MV_STAMP(5);
    __syncthreads();
    MV_STAMP(6);

    // phase 4: row sums (+ residual): 16 lanes per row, strided partials then a 4-step butterfly
    float best = -INFINITY; int bi = -1;   // fused greedy sampling: this thread's LAST maximum among its rows (ggml_vec_argmax_f32 keeps the last)
    if (paired) {
        // g[row] = silu(l) * r from the two row sums this workgroup holds (same 16-lane fixed-order sums as below)
        for (int rr = tid >> 4; rr < half_rows; rr += NW * 4) {
            float sl = 0.f, sr = 0.f;
            for (int j = tid & 15; j < nb; j += 16) { sl += part[rr * nb + j]; sr += part[(half_rows + rr) * nb + j]; }
            sl = row16_allsum_f32(sl); sr = row16_allsum_f32(sr);
            if ((tid & 15) == 0) a.y[row0 + rr] = (sl / (1.0f + expf(-sl))) * sr;
        }
        MV_STAMP(7);
        return;
    }
    int kq = 0;
    for (int rr = tid >> 4; rr < rows; rr += NW * 4, kq++) {
        float sum = 0.f;
        for (int j = tid & 15; j < nb; j += 16) sum += part[rr * nb + j];
        sum = row16_allsum_f32(sum);
        if ((tid & 15) == 0) {
            const int64_t row = row0 + rr;
            if (a.residual) sum = (kq < RN ? (kq == 0 ? res_pre[0] : kq == 1 ? res_pre[1] : kq == 2 ? res_pre[2] : res_pre[3]) : a.residual[row]) + sum;
            else if (a.res_embed.table) {
                int64_t r = *a.res_embed.index;
                if (r < 0 || r >= a.res_embed.n_rows) r = 0;
                float e = dequant_elem(a.res_embed.table + r * a.res_embed.row_bytes, a.res_embed.type, row);
                if (a.res_embed.scale) e = e * *a.res_embed.scale;
                sum = sum + e;
            }
            a.y[row] = sum;
            if (row < a.M && sum >= best) { best = sum; bi = (int) row; }   // rows ascend per thread: '>=' keeps the last maximum
        }
    }
    MV_STAMP(7);
    if (a.ticket) {
        // greedy sampling fused in (ggml's CPU argmax: the LAST maximum): every workgroup reduces its own rows to one candidate and publishes it
        // with returning agent-scope exchanges (complete at the coherence point once the old value is back - a plain store followed by the
        // counter increment was observed to lose that race once per ~1e5 hand-offs); the last workgroup to arrive merges the candidates
        __shared__ int s_last;
        __shared__ float am_v[NW];
        __shared__ int am_i[NW];
        float * cand_v = (float *) (a.ticket + 64);          // workspace: counter | 256 B | values[grid] | indices[grid]
        int * cand_i = (int *) (cand_v + gridDim.x);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi > bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { am_v[wave] = best; am_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; w++) if (am_v[w] > best || (am_v[w] == best && am_i[w] > bi)) { best = am_v[w]; bi = am_i[w]; }
            xchg_agent_wait(cand_v + blockIdx.x, best);
            xchg_agent_wait(cand_i + blockIdx.x, bi);
            s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;
        }
        __syncthreads();
        if (!s_last) return;
        best = -INFINITY; bi = -1;
        for (int i = tid; i < (int) gridDim.x; i += NW * 64) {
            const float v = ld_agent(cand_v + i); const int ci = ld_agent(cand_i + i);
            if (v > best || (v == best && ci > bi)) { best = v; bi = ci; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi > bi)) { best = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { am_v[wave] = best; am_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; w++) if (am_v[w] > best || (am_v[w] == best && am_i[w] > bi)) { best = am_v[w]; bi = am_i[w]; }
            const int code = bi < 0 ? 0 : bi;
            if (a.argmax_out[0]) *a.argmax_out[0] = code;
            if (a.argmax_out[1]) *a.argmax_out[1] = code;
            *a.ticket = 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// float-weight mat-vec (F32 / F16 / BF16 weights; 1..MV_MAX_COLS activation columns)
// ---------------------------------------------------------------------------------------------------
// y[:, c] = (W x[:, c]) (* out_scale) (+ residual[:, c]) for up to MV_MAX_COLS activation columns; the weights are
// streamed once (16 B per lane), the prologue (rms / layer norm, silu gate, gelu) is applied while staging x in LDS
template <int WT>
__global__ void __launch_bounds__(256) matvec_f_kernel(mv_args a, int rows_per_wave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float * xs = (float *) smem;   // [ncols][K]
    __shared__ double sh_cols[4][MV_MAX_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = (int) a.K, nc = a.ncols;
    float mean[MV_MAX_COLS], scale[MV_MAX_COLS];
#pragma unroll
    for (int c = 0; c < MV_MAX_COLS; c++) { mean[c] = 0.f; scale[c] = 1.f; }
    if (a.prologue == MV_RMSNORM || a.prologue == MV_LAYERNORM) {
        // statistics of all columns in one reduction each (double accumulators, wave all-reduce on the DPP path)
        auto block_sum_cols = [&](double v[MV_MAX_COLS]) {
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) if (c < nc) v[c] = wave_allsum_f64(v[c]);
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < MV_MAX_COLS; c++) sh_cols[wave][c] = v[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) v[c] = sh_cols[0][c] + sh_cols[1][c] + sh_cols[2][c] + sh_cols[3][c];
        };
        double acc[MV_MAX_COLS];
        if (a.prologue == MV_LAYERNORM) {
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) acc[c] = 0;
            for (int i = tid; i < K; i += 256)
#pragma unroll
                for (int c = 0; c < MV_MAX_COLS; c++) if (c < nc) acc[c] += (double) a.x[(int64_t) c * a.x_cs + i];
            block_sum_cols(acc);
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) mean[c] = (float) (acc[c] / (double) K);
        }
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) acc[c] = 0;
        for (int i = tid; i < K; i += 256)
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) if (c < nc) { const float v = a.x[(int64_t) c * a.x_cs + i] - mean[c]; acc[c] += (double) (v * v); }
        block_sum_cols(acc);
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) scale[c] = 1.0f / sqrtf((float) (acc[c] / (double) K) + a.eps);
    }
#pragma unroll
    for (int c = 0; c < MV_MAX_COLS; c++) {
        if (c >= nc) break;
        const float * x = a.x + (int64_t) c * a.x_cs;
        for (int i = tid; i < K; i += 256) {
            float v;
            switch (a.prologue) {
                case MV_RMSNORM:   v = a.alpha[i] * (x[i] * scale[c]); break;
                case MV_LAYERNORM: v = ((x[i] - mean[c]) * scale[c]) * a.alpha[i]; if (a.beta) v = v + a.beta[i]; break;
                case MV_GATE_SILU: { const float l = x[i], r = x[K + i]; v = (l / (1.0f + expf(-l))) * r; } break;
                case MV_GELU:      v = gelu_table(x[i]); break;
                default:           v = x[i]; break;
            }
            if (a.x_out != nullptr && blockIdx.x == 0) a.x_out[(int64_t) c * K + i] = v;
            if (WT == GGML_TYPE_F16) v = h2f(f2h(v));
            if (WT == GGML_TYPE_BF16) v = bf2f(f2bf(v));
            xs[c * K + i] = v;
        }
    }
    __syncthreads();
    constexpr int VEC = WT == GGML_TYPE_F32 ? 4 : 8;
    const int64_t rbase = ((int64_t) blockIdx.x * 4 + wave) * rows_per_wave;
    for (int rr = 0; rr < rows_per_wave; rr++) {
        const int64_t row = rbase + rr;
        if (row >= a.M) break;
        const char * w = a.w + row * a.row_bytes;
        double acc[MV_MAX_COLS];
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) acc[c] = 0;
        for (int k = lane * VEC; k < K; k += 64 * VEC) {
            float wf[VEC];
            if (WT == GGML_TYPE_F32) {
                const float4 wv = *(const float4 *) (w + (int64_t) k * 4);
                wf[0] = wv.x; wf[1] = wv.y; wf[2] = wv.z; wf[3] = wv.w;
            } else {
                const uint4 wv = *(const uint4 *) (w + (int64_t) k * 2);
                const uint32_t ww[4] = { wv.x, wv.y, wv.z, wv.w };
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint16_t lo = (uint16_t) (ww[j] & 0xffff), hi = (uint16_t) (ww[j] >> 16);
                    wf[2 * j]     = WT == GGML_TYPE_F16 ? h2f(lo) : bf2f(lo);
                    wf[2 * j + 1] = WT == GGML_TYPE_F16 ? h2f(hi) : bf2f(hi);
                }
            }
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) {
                if (c < nc) {
#pragma unroll
                    for (int j = 0; j < VEC; j++) acc[c] += (double) (wf[j] * xs[c * K + k + j]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) {
            if (c < nc) {
                const double tot = wave_allsum_f64(acc[c]);
                if (lane == 0) {
                    float sres = (float) tot;
                    if (a.out_act == 1) sres = gelu_table(sres);
                    if (a.out_scale) sres = sres * a.out_scale[row];
                    a.y[(int64_t) c * a.y_cs + row] = a.residual ? a.residual[(int64_t) c * a.r_cs + row] + sres : sres;
                }
            }
        }
    }
}

// Register-resident variant for the codec transformer's F32 linears (K = 256 * KV, KV <= 8): the weight rows of this wave are
// requested before anything else, the activation columns are read from global memory exactly once (statistics, normalisation
// and staging all work from registers), so the only serial latencies left are one weight fetch and two block reductions.
template <int KV, int RPW>
__global__ void __launch_bounds__(256) matvec_f32_reg_kernel(mv_args a) {
    __shared__ __attribute__((aligned(16))) float xs[MV_MAX_COLS * KV * 256];
    __shared__ double sh_cols[2][4][MV_MAX_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int K = KV * 256;
    const int nc = a.ncols;
    const int64_t rbase = ((int64_t) blockIdx.x * 4 + wave) * RPW;
    f32x4 wreg[RPW][KV];
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        int64_t row = rbase + rr;
        if (row >= a.M) row = a.M - 1;
        const char * w = a.w + row * a.row_bytes;
#pragma unroll
        for (int it = 0; it < KV; it++) wreg[rr][it] = __builtin_nontemporal_load((const f32x4 *) (w + ((int64_t) it * 256 + lane * 4) * 4));
    }
    float xr[MV_MAX_COLS][KV], al[KV], be[KV];
#pragma unroll
    for (int c = 0; c < MV_MAX_COLS; c++)
#pragma unroll
        for (int j = 0; j < KV; j++) xr[c][j] = a.x[(int64_t) (c < nc ? c : 0) * a.x_cs + tid + 256 * j];
    if (a.prologue == MV_GATE_SILU) {
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++)
#pragma unroll
            for (int j = 0; j < KV; j++) { const float l = xr[c][j], r = a.x[(int64_t) (c < nc ? c : 0) * a.x_cs + K + tid + 256 * j]; xr[c][j] = (l / (1.0f + expf(-l))) * r; }
    }
    if (a.prologue == MV_RMSNORM || a.prologue == MV_LAYERNORM) {
#pragma unroll
        for (int j = 0; j < KV; j++) { al[j] = a.alpha[tid + 256 * j]; be[j] = a.beta ? a.beta[tid + 256 * j] : 0.f; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.prologue == MV_RMSNORM || a.prologue == MV_LAYERNORM) {
        auto block_sum_cols = [&](double v[MV_MAX_COLS], int slot) {
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) if (c < nc) v[c] = wave_allsum_f64(v[c]);
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < MV_MAX_COLS; c++) sh_cols[slot][wave][c] = v[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) v[c] = sh_cols[slot][0][c] + sh_cols[slot][1][c] + sh_cols[slot][2][c] + sh_cols[slot][3][c];
        };
        float mean[MV_MAX_COLS];
        double acc[MV_MAX_COLS];
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) mean[c] = 0.f;
        if (a.prologue == MV_LAYERNORM) {
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) { acc[c] = 0; 
#pragma unroll
                for (int j = 0; j < KV; j++) acc[c] += (double) xr[c][j]; }
            block_sum_cols(acc, 0);
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) mean[c] = (float) (acc[c] / (double) K);
        }
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) { acc[c] = 0;
#pragma unroll
            for (int j = 0; j < KV; j++) { const float v = xr[c][j] - mean[c]; acc[c] += (double) (v * v); } }
        block_sum_cols(acc, 1);
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) {
            const float scale = 1.0f / sqrtf((float) (acc[c] / (double) K) + a.eps);
#pragma unroll
            for (int j = 0; j < KV; j++) {
                if (a.prologue == MV_RMSNORM) xr[c][j] = al[j] * (xr[c][j] * scale);
                else { float v = ((xr[c][j] - mean[c]) * scale) * al[j]; if (a.beta) v = v + be[j]; xr[c][j] = v; }
            }
        }
    } else if (a.prologue == MV_GELU) {
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++)
#pragma unroll
            for (int j = 0; j < KV; j++) xr[c][j] = gelu_table(xr[c][j]);
    }
#pragma unroll
    for (int c = 0; c < MV_MAX_COLS; c++) {
        if (c < nc) {
#pragma unroll
            for (int j = 0; j < KV; j++) {
                xs[c * K + tid + 256 * j] = xr[c][j];
                if (a.x_out != nullptr && blockIdx.x == 0) a.x_out[(int64_t) c * K + tid + 256 * j] = xr[c][j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        const int64_t row = rbase + rr;
        double acc[MV_MAX_COLS];
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) acc[c] = 0;
#pragma unroll
        for (int it = 0; it < KV; it++) {
            const int k = it * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < MV_MAX_COLS; c++) {
                if (c < nc) {
                    const float4 xv = *(const float4 *) (xs + c * K + k);
                    acc[c] += (double) (wreg[rr][it][0] * xv.x);
                    acc[c] += (double) (wreg[rr][it][1] * xv.y);
                    acc[c] += (double) (wreg[rr][it][2] * xv.z);
                    acc[c] += (double) (wreg[rr][it][3] * xv.w);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MV_MAX_COLS; c++) {
            if (c < nc) {
                const double tot = wave_allsum_f64(acc[c]);
                if (lane == 0 && row < a.M) {
                    float sres = (float) tot;
                    if (a.out_act == 1) sres = gelu_table(sres);
                    if (a.out_scale) sres = sres * a.out_scale[row];
                    a.y[(int64_t) c * a.y_cs + row] = a.residual ? a.residual[(int64_t) c * a.r_cs + row] + sres : sres;
                }
            }
        }
    }
}

// can linear_in [K, 2 F] run in the paired-gate form? (whole 64-super-block tiles per half and workgroup, one workgroup per CU)
bool k_matvec_pair_ok(int wtype, int64_t K, int64_t F) {
    if (!(wtype == GGML_TYPE_Q4_K || wtype == GGML_TYPE_Q4_0) || K % 256 != 0 || F < 2048) return false;
    const int nb = (int) (K / 256);
    int h = (int) ((F + 255) / 256);
    while ((h * nb) % 64 != 0) h++;
    return F % h == 0 && 2 * h * nb <= 4096 && F / h >= 128;
}
bool k_matvec_supported(int wtype, int64_t K, int64_t M) {
    if (M <= 0) return false;
    switch (wtype) {
        case GGML_TYPE_Q4_K: case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_0: return K % 256 == 0 && K <= 16384;
        case GGML_TYPE_F32:  return K % 4 == 0 && K <= 16384;
        case GGML_TYPE_F16: case GGML_TYPE_BF16: return K % 8 == 0 && K <= 16384;
        default: return false;
    }
}

// silu(left) * right of the gated FFN, quantised to padded Q8_K blocks in global memory: one wave per 256-element block.
// Used for long rows (K > 4096), where redoing this in every mat-vec workgroup would dominate the kernel.
template <int FMT>
__global__ void __launch_bounds__(64) gate_quant_kernel(const float * h, int K, xblk * out, float * g_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int e = b * 256 + lane * 4;
    const float4 l4 = *(const float4 *) (h + e), r4 = *(const float4 *) (h + K + e);
    const float l[4] = { l4.x, l4.y, l4.z, l4.w }, r[4] = { r4.x, r4.y, r4.z, r4.w };
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (l[k] / (1.0f + expf(-l[k]))) * r[k];
    if (g_out) *(float4 *) (g_out + e) = make_float4(v[0], v[1], v[2], v[3]);
    if (FMT == MVF_Q4K) quantize_block_q8k(out + b, v, lane); else quantize_block_q80((xblk80 *) (out + b), v, lane);
}
// alpha * rms_norm(x) quantised ONCE to padded Q8_K / Q8_0 blocks (K <= 4096: one workgroup of K / 4 threads, wave w owns block w; the
// whole vector lives in registers between the statistics and the quantiser). The mat-vecs that consume it (MV_PREQ8K) then start with a
// 4.8 KB copy instead of re-reading 32 KB and redoing the norm in each of their 256 workgroups.
template <int FMT>
__global__ void __launch_bounds__(1024) norm_quant_kernel(const float * x, const float * alpha, float eps, int K, xblk * out, float * n_out) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float4 x4 = *(const float4 *) (x + tid * 4), a4 = *(const float4 *) (alpha + tid * 4);
    float v[4] = { x4.x, x4.y, x4.z, x4.w };
    const float al[4] = { a4.x, a4.y, a4.z, a4.w };
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) acc += (double) (v[k] * v[k]);
    acc = wave_allsum_f64(acc);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    double tot = 0;
    for (int w = 0; w < nw; w++) tot += sh[w];
    const float scale = 1.0f / sqrtf((float) (tot / (double) K) + eps);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = al[k] * (v[k] * scale);
    if (n_out) *(float4 *) (n_out + tid * 4) = make_float4(v[0], v[1], v[2], v[3]);
    if (FMT == MVF_Q4K) quantize_block_q8k(out + wave, v, lane); else quantize_block_q80((xblk80 *) (out + wave), v, lane);
}
void k_norm_quant_q8k(hipStream_t s, const float * x, const float * alpha, float eps, int64_t K, void * out_blocks, int wtype, float * n_out) {
    GGML_ASSERT(K % 256 == 0 && K <= 4096);
    if (wtype == GGML_TYPE_Q4_K) norm_quant_kernel<MVF_Q4K><<<1, (int) (K / 4), 0, s>>>(x, alpha, eps, (int) K, (xblk *) out_blocks, n_out);
    else norm_quant_kernel<MVF_Q80><<<1, (int) (K / 4), 0, s>>>(x, alpha, eps, (int) K, (xblk *) out_blocks, n_out);
}

void k_gate_quant_q8k(hipStream_t s, const float * h, int64_t K, void * out_blocks, int wtype) {
    if (wtype == GGML_TYPE_Q4_K) gate_quant_kernel<MVF_Q4K><<<(int) (K / 256), 64, 0, s>>>(h, (int) K, (xblk *) out_blocks, nullptr);
    else gate_quant_kernel<MVF_Q80><<<(int) (K / 256), 64, 0, s>>>(h, (int) K, (xblk *) out_blocks, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// batched quantised mat-mul (prompt prefill: T = 2..64 activation rows against Q4_K weights)
// ---------------------------------------------------------------------------------------------------
// Each activation row is quantised to Q8_K exactly as for T = 1 (ggml quantises src1 row by row), then the 4-bit x 8-bit dot products
// of one 32-wide sub-block for 16 weight rows x 16 activation rows are ONE v_mfma_i32_16x16x32_i8: integer, hence exact. Operand maps
// (checked with integer data, tests/microbench/mfma_i8_layout.hip): lane l supplies A[row l&15][k = 8(l>>4)+j] and
// B[k = 8(l>>4)+j][col l&15], j = byte 0..7; it receives C[row 4(l>>4)+q][col l&15] in register q. A Q4_K 32-byte group holds sub-block 2j
// in its low nibbles and 2j+1 in its high nibbles, so the 8 bytes at 32j + 8(l>>4) of a row ARE this lane's A fragment of both sub-blocks.
// The 6-bit sub-block scale multiplies the int32 tile (24-bit multiplier, exact), mins meet the Q8_K block sums on the VALU, and the float
// combination per super-block is the T = 1 kernel's: d_w d_x isum - dmin_w d_x msum, summed over super-blocks in order.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

// activation record of the batched path: the Q8_K block (same rounding as quantize_block_q8k), its eight 32-wide sums split as
// 64 * hi + lo (hi in [-64, 63], lo in [0, 63]: both int8, so the mins product runs on the MFMA too), the block scale
struct xblkb { int8_t q[256]; int8_t bs_hi[8]; int8_t bs_lo[8]; float d; float pad[3]; };
static_assert(sizeof(xblkb) == 288, "xblkb layout");

__global__ void __launch_bounds__(64) quant_rows_q8k_kernel(const float * x, int64_t x_cs, int nb, xblkb * out) {
    __shared__ xblk tmp;
    const int b = blockIdx.x % nb, t = blockIdx.x / nb, lane = threadIdx.x;
    const float4 v4 = *(const float4 *) (x + (int64_t) t * x_cs + b * 256 + lane * 4);
    const float v[4] = { v4.x, v4.y, v4.z, v4.w };
    quantize_block_q8k(&tmp, v, lane);
    __syncthreads();
    xblkb * o = out + (int64_t) t * nb + b;
    ((uint32_t *) o->q)[lane] = ((const uint32_t *) tmp.q)[lane];
    if (lane < 8) {
        const int bs = (int) tmp.bsums[2 * lane] + (int) tmp.bsums[2 * lane + 1];
        o->bs_hi[lane] = (int8_t) (bs >> 6);
        o->bs_lo[lane] = (int8_t) (bs & 63);
    }
    if (lane == 0) o->d = tmp.d;
}

__device__ __forceinline__ void store_xblkb(xblkb * o, const xblk * tmp, int lane) {
    ((uint32_t *) o->q)[lane] = ((const uint32_t *) tmp->q)[lane];
    if (lane < 8) {
        const int bs = (int) tmp->bsums[2 * lane] + (int) tmp->bsums[2 * lane + 1];
        o->bs_hi[lane] = (int8_t) (bs >> 6);
        o->bs_lo[lane] = (int8_t) (bs & 63);
    }
    if (lane == 0) o->d = tmp->d;
}

// alpha * rms_norm(x) of one activation row, quantised (norm_kernel's arithmetic: float squares summed in double, 1 / sqrtf(mean + eps))
__global__ void __launch_bounds__(256) rms_quant_rows_q8k_kernel(const float * x, int64_t x_cs, const float * alpha, float eps, int K, int nb, xblkb * out) {
    __shared__ double sh[4];
    __shared__ xblk tmp[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float * row = x + (int64_t) t * x_cs;
    double acc = 0;
    for (int i = tid; i < K; i += 256) { const float v = row[i]; acc += (double) (v * v); }
    acc = wave_allsum_f64(acc);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    const float var = (float) ((sh[0] + sh[1] + sh[2] + sh[3]) / (double) K);
    const float scale = 1.0f / sqrtf(var + eps);
    for (int b = wave; b < nb; b += 4) {
        const int e = b * 256 + lane * 4;
        const float4 x4 = *(const float4 *) (row + e), a4 = *(const float4 *) (alpha + e);
        const float v[4] = { (x4.x * scale) * a4.x, (x4.y * scale) * a4.y, (x4.z * scale) * a4.z, (x4.w * scale) * a4.w };
        quantize_block_q8k(&tmp[wave], v, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        store_xblkb(out + (int64_t) t * nb + b, &tmp[wave], lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// silu(h[:n]) * h[n:] of one activation row, quantised (gate_quant_kernel's arithmetic)
__global__ void __launch_bounds__(64) gate_quant_rows_q8k_kernel(const float * h, int64_t h_cs, int n, int nb, xblkb * out) {
    __shared__ xblk tmp;
    const int b = blockIdx.x % nb, t = blockIdx.x / nb, lane = threadIdx.x;
    const int e = b * 256 + lane * 4;
    const float4 l4 = *(const float4 *) (h + (int64_t) t * h_cs + e), r4 = *(const float4 *) (h + (int64_t) t * h_cs + n + e);
    const float l[4] = { l4.x, l4.y, l4.z, l4.w }, r[4] = { r4.x, r4.y, r4.z, r4.w };
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (l[k] / (1.0f + expf(-l[k]))) * r[k];
    quantize_block_q8k(&tmp, v, lane);
    __syncthreads();
    store_xblkb(out + (int64_t) t * nb + b, &tmp, lane);
}

#define MMQ_NW 4          // waves per workgroup; tiles of 4 consecutive super-blocks are dealt round-robin to the waves (split K)
#define MMQ_TSB 4         // super-blocks per tile: 16 rows x 4 x 144 B = 9216 B = 9 coalesced 16-byte loads per lane
#define MMQ_ROW 608       // LDS bytes per staged row (576 used): 152 dwords = 24 mod 64, so the 16 rows of one ds_read_b64 spread over all banks
// The activation rows take the MFMA's M side and the weight rows its N side: lane l holds C[t = 4(l>>4)+q][row l&15], ONE weight row per
// lane. Its 6-bit sub-block scale s = 8 sh + sl is folded into the 4-bit weights before the MFMA (q * sh and q * sl stay below 128, a
// packed dword times a small integer never carries between bytes), so the eight sub-blocks of a super-block ACCUMULATE in the matrix
// core: isum = 8 * sum(x . q sh) + sum(x . q sl), no per-tile scaling on the VALU. The mins meet the split block sums the same way.
template <int NT>
__global__ void __launch_bounds__(64 * MMQ_NW) mm_q4k_mfma_kernel(const char * w, int64_t row_bytes, int nb, int M, int T, const xblkb * xq,
                                                                   float * y, int64_t y_cs, const float * residual, int64_t r_cs) {
    __shared__ __attribute__((aligned(16))) char stage_all[MMQ_NW][16 * MMQ_ROW];
    __shared__ float red[MMQ_NW - 1][NT * 4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 16;
    char * stage = stage_all[wave];
    const char * xa[NT]; const char * xo[NT][4];   // A operand: activation row nt*16 + r; results: activation rows nt*16 + 4g + q
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = nt * 16 + r < T ? nt * 16 + r : T - 1;
        xa[nt] = (const char *) (xq + (int64_t) col * nb);
#pragma unroll
        for (int q = 0; q < 4; q++) { const int co = nt * 16 + 4 * g + q < T ? nt * 16 + 4 * g + q : T - 1; xo[nt][q] = (const char *) (xq + (int64_t) co * nb); }
    }
    float acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[nt][q] = 0.f;
    const int ntiles = (nb + MMQ_TSB - 1) / MMQ_TSB;
    // the tile of 16 rows x 4 super-blocks as 576 chunks of 16 B: chunk c -> row c / 36, byte (c % 36) * 16 of that row's 576 B
    int grow[9], goff[9], loff[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int c = i * 64 + lane, rr = c / 36, o = (c - rr * 36) * 16;
        grow[i] = row0 + rr < M ? row0 + rr : M - 1;
        goff[i] = o;
        loff[i] = rr * MMQ_ROW + o;
    }
    auto load_tile = [&](int tile, u32x4 * dst) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int64_t off = (int64_t) tile * (MMQ_TSB * 144) + goff[i];
            if (off > row_bytes - 16) off = row_bytes - 16;   // ragged last tile: stay inside the row
            dst[i] = __builtin_nontemporal_load((const u32x4 *) (w + (int64_t) grow[i] * row_bytes + off));
        }
    };
    u32x4 rn[9];
    if (wave < ntiles) load_tile(wave, rn);
    for (int tile = wave; tile < ntiles; tile += MMQ_NW) {
#pragma unroll
        for (int i = 0; i < 9; i++) *(u32x4 *) (stage + loff[i]) = rn[i];
        if (tile + MMQ_NW < ntiles) load_tile(tile + MMQ_NW, rn);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nsb = nb - tile * MMQ_TSB < MMQ_TSB ? nb - tile * MMQ_TSB : MMQ_TSB;
        for (int sb = 0; sb < nsb; sb++) {
            const int64_t xoff = (int64_t) (tile * MMQ_TSB + sb) * (int64_t) sizeof(xblkb);
            const char * wrow = stage + r * MMQ_ROW + sb * 144;
            const uint4 hdr = *(const uint4 *) wrow;   // d, dmin (f16 x 2) + 12 scale bytes of this lane's weight row
            uint32_t sc[2], mn[2];
            const uint32_t scl[3] = { hdr.y, hdr.z, hdr.w };
            q4k_unpack_scales((const uint8_t *) scl, sc, mn);
            const float dw = h2f((uint16_t) (hdr.x & 0xffff)), dm = h2f((uint16_t) (hdr.x >> 16));
            const uint32_t shi[2] = { (sc[0] >> 3) & 0x07070707u, (sc[1] >> 3) & 0x07070707u }, slo[2] = { sc[0] & 0x07070707u, sc[1] & 0x07070707u };
            i32x4_t ihi[NT], ilo[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) { ihi[nt] = i32x4_t{ 0, 0, 0, 0 }; ilo[nt] = i32x4_t{ 0, 0, 0, 0 }; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint2 q8 = *(const uint2 *) (wrow + 16 + 32 * j + 8 * g);
                const uint32_t l0 = q8.x & 0x0F0F0F0Fu, l1 = q8.y & 0x0F0F0F0Fu, h0 = (q8.x >> 4) & 0x0F0F0F0Fu, h1 = (q8.y >> 4) & 0x0F0F0F0Fu;
                const uint32_t sh0 = (shi[j >> 1] >> (16 * (j & 1))) & 0xff, sh1 = (shi[j >> 1] >> (16 * (j & 1) + 8)) & 0xff;
                const uint32_t sl0 = (slo[j >> 1] >> (16 * (j & 1))) & 0xff, sl1 = (slo[j >> 1] >> (16 * (j & 1) + 8)) & 0xff;
                // packed bytes (<= 15) times a scale half (<= 7): <= 105 per byte, so a packed 16-bit multiply (two bytes per half-word,
                // <= 0x6969) never carries; v_pk_mul_lo_u16 is full rate, a 32-bit v_mul_lo_u32 a quarter
                auto pkmul = [](uint32_t v, uint32_t k) { const u16x2_t a = __builtin_bit_cast(u16x2_t, v), b = { (unsigned short) k, (unsigned short) k }; return __builtin_bit_cast(uint32_t, (u16x2_t) (a * b)); };
                const long w_h0 = (long) ((uint64_t) pkmul(l0, sh0) | ((uint64_t) pkmul(l1, sh0) << 32));
                const long w_l0 = (long) ((uint64_t) pkmul(l0, sl0) | ((uint64_t) pkmul(l1, sl0) << 32));
                const long w_h1 = (long) ((uint64_t) pkmul(h0, sh1) | ((uint64_t) pkmul(h1, sh1) << 32));
                const long w_l1 = (long) ((uint64_t) pkmul(h0, sl1) | ((uint64_t) pkmul(h1, sl1) << 32));
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const char * xqp = xa[nt] + xoff + 64 * j + 8 * g;
                    const long x_lo = *(const long *) xqp, x_hi = *(const long *) (xqp + 32);
                    ihi[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_lo, w_h0, ihi[nt], 0, 0, 0);
                    ilo[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_lo, w_l0, ilo[nt], 0, 0, 0);
                    ihi[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_hi, w_h1, ihi[nt], 0, 0, 0);
                    ilo[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_hi, w_l1, ilo[nt], 0, 0, 0);
                }
            }
            // mins x block sums: K = 8 of the 32 (lanes of group 0 carry them, the rest zeros)
            const long mins_b = g == 0 ? (long) ((uint64_t) mn[0] | ((uint64_t) mn[1] << 32)) : 0L;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                long a_hi = 0, a_lo = 0;
                if (g == 0) { a_hi = *(const long *) (xa[nt] + xoff + 256); a_lo = *(const long *) (xa[nt] + xoff + 264); }
                const i32x4_t z = { 0, 0, 0, 0 };
                const i32x4_t mh = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_hi, mins_b, z, 0, 0, 0);
                const i32x4_t ml = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_lo, mins_b, z, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int isum = 8 * ihi[nt][q] + ilo[nt][q], msum = 64 * mh[q] + ml[q];
                    const float d8 = *(const float *) (xo[nt][q] + xoff + 272);
                    const float d = dw * d8, dmin = dm * d8;
                    acc[nt][q] += d * (float) isum - dmin * (float) msum;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // every lane is done with the staged tile before it is overwritten
    }
    if (wave > 0) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[wave - 1][nt * 4 + q][lane] = acc[nt][q];
    }
    __syncthreads();
    if (wave > 0) return;
    const int row = row0 + r;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v = acc[nt][q];
#pragma unroll
            for (int ww = 0; ww < MMQ_NW - 1; ww++) v += red[ww][nt * 4 + q][lane];
            const int col = nt * 16 + 4 * g + q;
            if (col >= T || row >= M) continue;
            if (residual) v = residual[(int64_t) col * r_cs + row] + v;
            y[(int64_t) col * y_cs + row] = v;
        }
}

// Large-M variant: the four waves of a workgroup take four 16-row tiles (64 rows) and walk K TOGETHER, so the activation tile of
// the current 4 super-blocks (T x 4 x 288 B) is fetched from L2 once per workgroup into LDS instead of once per wave through L1
// (with K split over the waves the activation traffic was ~4x the weight bytes). Weights: per-wave tile as above. Both tiles of the
// next step are requested into registers before the current one is multiplied.
#define MMQ_XCOL 1184     // LDS bytes per activation column (4 x 288 used): 296 dwords = 40 mod 64 -> 16 columns of a ds_read_b64 spread over the banks
template <int NT>
__global__ void __launch_bounds__(256) mm_q4k_mfma_rows_kernel(const char * w, int64_t row_bytes, int nb, int M, int T, const xblkb * xq,
                                                               float * y, int64_t y_cs, const float * residual, int64_t r_cs) {
    __shared__ __attribute__((aligned(16))) char stage_all[4][16 * MMQ_ROW];
    __shared__ __attribute__((aligned(16))) char xs[NT * 16 * MMQ_XCOL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 64 + wave * 16;
    char * stage = stage_all[wave];
    float acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[nt][q] = 0.f;
    const int ntiles = (nb + MMQ_TSB - 1) / MMQ_TSB;
    int grow[9], goff[9], loff[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int c = i * 64 + lane, rr = c / 36, o = (c - rr * 36) * 16;
        grow[i] = row0 + rr < M ? row0 + rr : M - 1;
        goff[i] = o;
        loff[i] = rr * MMQ_ROW + o;
    }
    auto load_tile = [&](int tile, u32x4 * dst) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int64_t off = (int64_t) tile * (MMQ_TSB * 144) + goff[i];
            if (off > row_bytes - 16) off = row_bytes - 16;
            dst[i] = __builtin_nontemporal_load((const u32x4 *) (w + (int64_t) grow[i] * row_bytes + off));
        }
    };
    // activation tile: NT*16 columns x 1152 B = NT*16*72 chunks of 16 B over 256 threads
    constexpr int XCH = NT * 16 * 72 / 256;      // 4.5 for NT = 1 -> handled with a guard below
    constexpr int XN = (NT * 16 * 72 + 255) / 256;
    auto load_x = [&](int tile, u32x4 * dst) {
#pragma unroll
        for (int i = 0; i < XN; i++) {
            const int c = i * 256 + tid, col = c / 72, o = (c - col * 72) * 16;
            const int cc = col < T ? col : T - 1;
            int64_t sbo = (int64_t) tile * (MMQ_TSB * 288) + o;
            if (sbo > (int64_t) nb * 288 - 16) sbo = (int64_t) nb * 288 - 16;
            if (c < NT * 16 * 72) dst[i] = *(const u32x4 *) ((const char *) xq + (int64_t) cc * nb * 288 + sbo);
        }
    };
    (void) XCH;
    u32x4 rn[9], xn[XN];
    load_tile(0, rn);
    load_x(0, xn);
    for (int tile = 0; tile < ntiles; tile++) {
        __syncthreads();                          // everyone is done with the previous activation tile
#pragma unroll
        for (int i = 0; i < 9; i++) *(u32x4 *) (stage + loff[i]) = rn[i];
#pragma unroll
        for (int i = 0; i < XN; i++) {
            const int c = i * 256 + tid, col = c / 72, o = (c - col * 72) * 16;
            if (c < NT * 16 * 72) *(u32x4 *) (xs + col * MMQ_XCOL + o) = xn[i];
        }
        if (tile + 1 < ntiles) { load_tile(tile + 1, rn); load_x(tile + 1, xn); }
        __syncthreads();
        const int nsb = nb - tile * MMQ_TSB < MMQ_TSB ? nb - tile * MMQ_TSB : MMQ_TSB;
        for (int sb = 0; sb < nsb; sb++) {
            const char * wrow = stage + r * MMQ_ROW + sb * 144;
            const uint4 hdr = *(const uint4 *) wrow;
            uint32_t sc[2], mn[2];
            const uint32_t scl[3] = { hdr.y, hdr.z, hdr.w };
            q4k_unpack_scales((const uint8_t *) scl, sc, mn);
            const float dw = h2f((uint16_t) (hdr.x & 0xffff)), dm = h2f((uint16_t) (hdr.x >> 16));
            const uint32_t shi[2] = { (sc[0] >> 3) & 0x07070707u, (sc[1] >> 3) & 0x07070707u }, slo[2] = { sc[0] & 0x07070707u, sc[1] & 0x07070707u };
            i32x4_t ihi[NT], ilo[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) { ihi[nt] = i32x4_t{ 0, 0, 0, 0 }; ilo[nt] = i32x4_t{ 0, 0, 0, 0 }; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint2 q8 = *(const uint2 *) (wrow + 16 + 32 * j + 8 * g);
                const uint32_t l0 = q8.x & 0x0F0F0F0Fu, l1 = q8.y & 0x0F0F0F0Fu, h0 = (q8.x >> 4) & 0x0F0F0F0Fu, h1 = (q8.y >> 4) & 0x0F0F0F0Fu;
                const uint32_t sh0 = (shi[j >> 1] >> (16 * (j & 1))) & 0xff, sh1 = (shi[j >> 1] >> (16 * (j & 1) + 8)) & 0xff;
                const uint32_t sl0 = (slo[j >> 1] >> (16 * (j & 1))) & 0xff, sl1 = (slo[j >> 1] >> (16 * (j & 1) + 8)) & 0xff;
                auto pkmul = [](uint32_t v, uint32_t k) { const u16x2_t a = __builtin_bit_cast(u16x2_t, v), b = { (unsigned short) k, (unsigned short) k }; return __builtin_bit_cast(uint32_t, (u16x2_t) (a * b)); };
                const long w_h0 = (long) ((uint64_t) pkmul(l0, sh0) | ((uint64_t) pkmul(l1, sh0) << 32));
                const long w_l0 = (long) ((uint64_t) pkmul(l0, sl0) | ((uint64_t) pkmul(l1, sl0) << 32));
                const long w_h1 = (long) ((uint64_t) pkmul(h0, sh1) | ((uint64_t) pkmul(h1, sh1) << 32));
                const long w_l1 = (long) ((uint64_t) pkmul(h0, sl1) | ((uint64_t) pkmul(h1, sl1) << 32));
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const char * xqp = xs + (nt * 16 + r) * MMQ_XCOL + sb * 288 + 64 * j + 8 * g;
                    const long x_lo = *(const long *) xqp, x_hi = *(const long *) (xqp + 32);
                    ihi[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_lo, w_h0, ihi[nt], 0, 0, 0);
                    ilo[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_lo, w_l0, ilo[nt], 0, 0, 0);
                    ihi[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_hi, w_h1, ihi[nt], 0, 0, 0);
                    ilo[nt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x_hi, w_l1, ilo[nt], 0, 0, 0);
                }
            }
            const long mins_b = g == 0 ? (long) ((uint64_t) mn[0] | ((uint64_t) mn[1] << 32)) : 0L;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const char * xb = xs + (nt * 16 + r) * MMQ_XCOL + sb * 288;
                const long a_hi = g == 0 ? *(const long *) (xb + 256) : 0L, a_lo = g == 0 ? *(const long *) (xb + 264) : 0L;
                const i32x4_t z = { 0, 0, 0, 0 };
                const i32x4_t mh = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_hi, mins_b, z, 0, 0, 0);
                const i32x4_t ml = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_lo, mins_b, z, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int isum = 8 * ihi[nt][q] + ilo[nt][q], msum = 64 * mh[q] + ml[q];
                    const float d8 = *(const float *) (xs + (nt * 16 + 4 * g + q) * MMQ_XCOL + sb * 288 + 272);
                    const float d = dw * d8, dmin = dm * d8;
                    acc[nt][q] += d * (float) isum - dmin * (float) msum;
                }
            }
        }
    }
    const int row = row0 + r;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = nt * 16 + 4 * g + q;
            if (col >= T || row >= M) continue;
            float v = acc[nt][q];
            if (residual) v = residual[(int64_t) col * r_cs + row] + v;
            y[(int64_t) col * y_cs + row] = v;
        }
}

// ---- the same batched mat-mul for Q8_0 / Q4_0 weights (tts / stt checkpoints: `-q q8_0`, the loader's Q4_K -> Q4_0 fall-back) ---------------
// Activation rows are quantised to Q8_0 (32-wide blocks, F16 scale - what ggml's CPU backend does for these weight types); a 32-wide
// block is exactly the K = 32 of one v_mfma_i32_16x16x32_i8, so every block is ONE matrix instruction per 16 x 16 tile whose int32 sums
// are scaled by d_w[row] * d_x[t] on the way into the float accumulators (vec_dot_q8_0_q8_0 / vec_dot_q4_0_q8_0 per block).
struct xblk80b { int8_t q[256]; float d[8]; };   // eight consecutive blocks of one activation row; same 288 B as xblkb
static_assert(sizeof(xblk80b) == sizeof(xblkb), "the two batched activation records share the workspace sizing");

__device__ __forceinline__ void store_xblk80b(xblk80b * o, const xblk80 * tmp, int lane) {
    ((uint32_t *) o->q)[lane] = ((const uint32_t *) tmp->q)[lane];
    if (lane < 8) o->d[lane] = tmp->d[lane];
}
__global__ void __launch_bounds__(64) quant_rows_q80_kernel(const float * x, int64_t x_cs, int nb, xblk80b * out) {
    __shared__ xblk80 tmp;
    const int b = blockIdx.x % nb, t = blockIdx.x / nb, lane = threadIdx.x;
    const float4 v4 = *(const float4 *) (x + (int64_t) t * x_cs + b * 256 + lane * 4);
    const float v[4] = { v4.x, v4.y, v4.z, v4.w };
    quantize_block_q80(&tmp, v, lane);
    __syncthreads();
    store_xblk80b(out + (int64_t) t * nb + b, &tmp, lane);
}
__global__ void __launch_bounds__(256) rms_quant_rows_q80_kernel(const float * x, int64_t x_cs, const float * alpha, float eps, int K, int nb, xblk80b * out) {
    __shared__ double sh[4];
    __shared__ xblk80 tmp[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float * row = x + (int64_t) t * x_cs;
    double acc = 0;
    for (int i = tid; i < K; i += 256) { const float v = row[i]; acc += (double) (v * v); }
    acc = wave_allsum_f64(acc);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    const float var = (float) ((sh[0] + sh[1] + sh[2] + sh[3]) / (double) K);
    const float scale = 1.0f / sqrtf(var + eps);
    for (int b = wave; b < nb; b += 4) {
        const int e = b * 256 + lane * 4;
        const float4 x4 = *(const float4 *) (row + e), a4 = *(const float4 *) (alpha + e);
        const float v[4] = { (x4.x * scale) * a4.x, (x4.y * scale) * a4.y, (x4.z * scale) * a4.z, (x4.w * scale) * a4.w };
        quantize_block_q80(&tmp[wave], v, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        store_xblk80b(out + (int64_t) t * nb + b, &tmp[wave], lane);
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void __launch_bounds__(64) gate_quant_rows_q80_kernel(const float * h, int64_t h_cs, int n, int nb, xblk80b * out) {
    __shared__ xblk80 tmp;
    const int b = blockIdx.x % nb, t = blockIdx.x / nb, lane = threadIdx.x;
    const int e = b * 256 + lane * 4;
    const float4 l4 = *(const float4 *) (h + (int64_t) t * h_cs + e), r4 = *(const float4 *) (h + (int64_t) t * h_cs + n + e);
    const float l[4] = { l4.x, l4.y, l4.z, l4.w }, r[4] = { r4.x, r4.y, r4.z, r4.w };
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (l[k] / (1.0f + expf(-l[k]))) * r[k];
    quantize_block_q80(&tmp, v, lane);
    __syncthreads();
    store_xblk80b(out + (int64_t) t * nb + b, &tmp, lane);
}

// Tile = 16 weight rows x TSB groups of eight blocks (Q8_0: 2 x 272 B, Q4_0: 4 x 144 B per row; <= 576 B, staged at the MMQ_ROW stride),
// dealt round-robin to the four waves (split K) exactly as in mm_q4k_mfma_kernel. The activation rows take the MFMA's M side, the
// weight rows its N side: lane l = (r = l & 15, g = l >> 4) feeds weights 8 g .. 8 g + 7 of block j of row r and receives
// C[t = 4 g + q][row r]. Q4_0: weights 0..15 of a block are the low nibbles of its 16 bytes, 16..31 the high ones, so g picks the
// byte half (g & 1) and the nibble (g >> 1); (nibble - 8) as int8 without a per-byte borrow: ((nibble ^ 8) + 0x78) ^ 0x78.
template <int NT, int FMT>
__global__ void __launch_bounds__(64 * MMQ_NW) mm_q80_mfma_kernel(const char * w, int64_t row_bytes, int nb, int M, int T, const xblk80b * xq,
                                                                   float * y, int64_t y_cs, const float * residual, int64_t r_cs) {
    constexpr int SBB = FMT == MVF_Q80 ? 272 : 144, BLK = FMT == MVF_Q80 ? 34 : 18, TSB = FMT == MVF_Q80 ? 2 : 4;
    constexpr int RB = TSB * SBB, CPR = RB / 16, NCH = 16 * CPR, NL = (NCH + 63) / 64;   // bytes / 16-byte chunks per staged row, chunks per tile, loads per lane
    static_assert(RB <= MMQ_ROW && RB % 16 == 0, "staged row");
    __shared__ __attribute__((aligned(16))) char stage_all[MMQ_NW][16 * MMQ_ROW];
    __shared__ float red[MMQ_NW - 1][NT * 4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 16;
    char * stage = stage_all[wave];
    const char * xa[NT]; const char * xo[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = nt * 16 + r < T ? nt * 16 + r : T - 1;
        xa[nt] = (const char *) (xq + (int64_t) col * nb);
#pragma unroll
        for (int q = 0; q < 4; q++) { const int co = nt * 16 + 4 * g + q < T ? nt * 16 + 4 * g + q : T - 1; xo[nt][q] = (const char *) (xq + (int64_t) co * nb); }
    }
    float acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[nt][q] = 0.f;
    const int ntiles = (nb + TSB - 1) / TSB;
    int grow[NL], goff[NL], loff[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int c0 = i * 64 + lane, c = c0 < NCH ? c0 : NCH - 1, rr = c / CPR, o = (c - rr * CPR) * 16;
        grow[i] = row0 + rr < M ? row0 + rr : M - 1;
        goff[i] = o;
        loff[i] = rr * MMQ_ROW + o;
    }
    auto load_tile = [&](int tile, u32x4 * dst) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            int64_t off = (int64_t) tile * RB + goff[i];
            if (off > row_bytes - 16) off = row_bytes - 16;   // ragged last tile: stay inside the row
            dst[i] = __builtin_nontemporal_load((const u32x4 *) (w + (int64_t) grow[i] * row_bytes + off));
        }
    };
    u32x4 rn[NL];
    if (wave < ntiles) load_tile(wave, rn);
    for (int tile = wave; tile < ntiles; tile += MMQ_NW) {
#pragma unroll
        for (int i = 0; i < NL; i++) if (i * 64 + lane < NCH) *(u32x4 *) (stage + loff[i]) = rn[i];
        if (tile + MMQ_NW < ntiles) load_tile(tile + MMQ_NW, rn);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nsb = nb - tile * TSB < TSB ? nb - tile * TSB : TSB;
        for (int sb = 0; sb < nsb; sb++) {
            const int64_t xoff = (int64_t) (tile * TSB + sb) * (int64_t) sizeof(xblk80b);
            const uint32_t * D = (const uint32_t *) (stage + r * MMQ_ROW + sb * SBB);   // this lane's weight row: eight blocks
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int off = BLK * j;
                const float dw = h2f((uint16_t) (D[off >> 2] >> ((off & 2) * 8)));
                long wq;
                if (FMT == MVF_Q80) {
                    const uint32_t w0 = dword_at2(D, off + 2 + 8 * g), w1 = dword_at2(D, off + 6 + 8 * g);
                    wq = (long) ((uint64_t) w0 | ((uint64_t) w1 << 32));
                } else {
                    const int sh = 4 * (g >> 1);
                    const uint32_t n0 = (dword_at2(D, off + 2 + 8 * (g & 1)) >> sh) & 0x0F0F0F0Fu, n1 = (dword_at2(D, off + 6 + 8 * (g & 1)) >> sh) & 0x0F0F0F0Fu;
                    const uint32_t w0 = ((n0 ^ 0x08080808u) + 0x78787878u) ^ 0x78787878u, w1 = ((n1 ^ 0x08080808u) + 0x78787878u) ^ 0x78787878u;
                    wq = (long) ((uint64_t) w0 | ((uint64_t) w1 << 32));
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const long xv = *(const long *) (xa[nt] + xoff + 32 * j + 8 * g);
                    const i32x4_t z = { 0, 0, 0, 0 };
                    const i32x4_t c = __builtin_amdgcn_mfma_i32_16x16x32_i8(xv, wq, z, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float dx = *(const float *) (xo[nt][q] + xoff + 256 + 4 * j);
                        if (FMT == MVF_Q80) acc[nt][q] += (float) c[q] * (dw * dx);
                        else acc[nt][q] += ((float) c[q] * dw) * dx;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // every lane is done with the staged tile before it is overwritten
    }
    if (wave > 0) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[wave - 1][nt * 4 + q][lane] = acc[nt][q];
    }
    __syncthreads();
    if (wave > 0) return;
    const int row = row0 + r;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v = acc[nt][q];
#pragma unroll
            for (int ww = 0; ww < MMQ_NW - 1; ww++) v += red[ww][nt * 4 + q][lane];
            const int col = nt * 16 + 4 * g + q;
            if (col >= T || row >= M) continue;
            if (residual) v = residual[(int64_t) col * r_cs + row] + v;
            y[(int64_t) col * y_cs + row] = v;
        }
}

static int env_int(const char * name, int def);
size_t k_mm_q4k_batched_ws_size(int64_t K, int64_t T) { return (size_t) (K / 256) * (size_t) T * sizeof(xblkb); }
bool k_mm_q4k_batched_supported(int wtype, int64_t K, int64_t M, int64_t T) {
    return (wtype == GGML_TYPE_Q4_K || wtype == GGML_TYPE_Q8_0 || wtype == GGML_TYPE_Q4_0) && K % 256 == 0 && T >= 2 && T <= 64 && M >= 16;
}
void k_mm_q4k_batched(hipStream_t s, int wtype, const char * w, int64_t row_bytes, int64_t K, int64_t M, int64_t T, const float * x, int64_t x_cs,
                      void * ws, float * y, int64_t y_cs, const float * residual, int64_t r_cs, int prologue, const float * alpha, float eps) {
    const int nb = (int) (K / 256);
    if (wtype != GGML_TYPE_Q4_K) {   // Q8_0 / Q4_0 weights: Q8_0 activation rows
        xblk80b * xq0 = (xblk80b *) ws;
        if (prologue == MV_RMSNORM) rms_quant_rows_q80_kernel<<<(int) T, 256, 0, s>>>(x, x_cs, alpha, eps, (int) K, nb, xq0);
        else if (prologue == MV_GATE_SILU) gate_quant_rows_q80_kernel<<<(int) (T * nb), 64, 0, s>>>(x, x_cs, (int) K, nb, xq0);
        else quant_rows_q80_kernel<<<(int) (T * nb), 64, 0, s>>>(x, x_cs, nb, xq0);
        const int grid = (int) ((M + 15) / 16), thr = 64 * MMQ_NW;
        for (int64_t c0 = 0; c0 < T; c0 += 32) {
            const int Tc = (int) (T - c0 < 32 ? T - c0 : 32);
            const xblk80b * xq = xq0 + c0 * nb;
            float * yc = y + c0 * y_cs;
            const float * rc = residual ? residual + c0 * r_cs : nullptr;
            if (wtype == GGML_TYPE_Q8_0) {
                if (Tc <= 16) mm_q80_mfma_kernel<1, MVF_Q80><<<grid, thr, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
                else mm_q80_mfma_kernel<2, MVF_Q80><<<grid, thr, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
            } else {
                if (Tc <= 16) mm_q80_mfma_kernel<1, MVF_Q40><<<grid, thr, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
                else mm_q80_mfma_kernel<2, MVF_Q40><<<grid, thr, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
            }
        }
        return;
    }
    if (prologue == MV_RMSNORM) rms_quant_rows_q8k_kernel<<<(int) T, 256, 0, s>>>(x, x_cs, alpha, eps, (int) K, nb, (xblkb *) ws);
    else if (prologue == MV_GATE_SILU) gate_quant_rows_q8k_kernel<<<(int) (T * nb), 64, 0, s>>>(x, x_cs, (int) K, nb, (xblkb *) ws);
    else quant_rows_q8k_kernel<<<(int) (T * nb), 64, 0, s>>>(x, x_cs, nb, (xblkb *) ws);
    const int grid = (int) ((M + 15) / 16);
    const int thr = 64 * MMQ_NW;
    for (int64_t c0 = 0; c0 < T; c0 += 32) {   // 32 columns per pass (two MFMA column tiles; four would leave one wave per SIMD)
        const int Tc = (int) (T - c0 < 32 ? T - c0 : 32);
        const xblkb * xq = (const xblkb *) ws + c0 * nb;
        float * yc = y + c0 * y_cs;
        const float * rc = residual ? residual + c0 * r_cs : nullptr;
        static const int rows_min = env_int("MI355X_MMQ_ROWS_MIN_M", 8192);   // >= 128 workgroups of 64 rows
        if (M >= rows_min) {
            const int grid64 = (int) ((M + 63) / 64);
            if (Tc <= 16) mm_q4k_mfma_rows_kernel<1><<<grid64, 256, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
            else mm_q4k_mfma_rows_kernel<2><<<grid64, 256, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
        } else if (Tc <= 16) mm_q4k_mfma_kernel<1><<<grid, thr, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
        else mm_q4k_mfma_kernel<2><<<grid, thr, 0, s>>>(w, row_bytes, nb, (int) M, Tc, xq, yc, y_cs, rc, r_cs);
    }
}

static mv_profile * g_mv_profile = nullptr;
void k_matvec_set_profile(mv_profile * p) { g_mv_profile = p; }

static int env_int(const char * name, int def) { const char * v = getenv(name); return v ? atoi(v) : def; }

void k_matvec(hipStream_t s, const mv_args & a) {
    if (a.wtype == GGML_TYPE_Q4_K || a.wtype == GGML_TYPE_Q8_0 || a.wtype == GGML_TYPE_Q4_0) {
        const int nb = (int) (a.K / 256);
        const int fmt = a.wtype == GGML_TYPE_Q4_K ? MVF_Q4K : a.wtype == GGML_TYPE_Q8_0 ? MVF_Q80 : MVF_Q40;
        const size_t tile_bytes = fmt == MVF_Q80 ? 64 * 272 : 64 * 144;
        // Workgroup shape. Large matrices (>= ~6 tiles per CU): ONE workgroup of 8 (optionally 12) waves per CU, so the activation
        // prologue (cost ~K, identical in every workgroup) runs once per CU and every wave keeps a 9 KB tile in flight.
        // Small matrices: 4-wave workgroups of >= 1 tile each, as many as there are tiles (latency-bound anyway).
        static const int force_nw = env_int("MI355X_MV_NW", 0), tpw_min = env_int("MI355X_MV_TPW", 4), grid_min = env_int("MI355X_MV_GRID_MIN", 128);   // bench sweep (tests/microbench/sweep_bench_mv.sh): Depth 1.27 -> 1.22 ms vs 256
        const int64_t tiles_total = (a.M * nb + 63) / 64;
        int nw = 4, rows;
        static const int big_tiles = env_int("MI355X_MV_BIG_TILES", 256 * 6), w12_tiles = env_int("MI355X_MV_12W_TILES", 1 << 30);   // bench sweep: 8 waves beat 12 on every moshika shape (Temporal 1.79 -> 1.73 ms)
        if (force_nw ? force_nw > 4 : tiles_total >= big_tiles) {
            nw = force_nw ? force_nw : (tiles_total >= w12_tiles ? 12 : 8);
            rows = (int) ((a.M + 255) / 256);
            while (rows * nb > 4096) rows = (rows + 1) / 2;
        } else {
            int tpw = tpw_min > nb / 4 ? tpw_min : nb / 4;
            rows = (tpw * 64 + nb - 1) / nb;
            if (rows * nb > 4096) rows = 4096 / nb;
            while (rows > 1 && (a.M + rows - 1) / rows < grid_min && (rows / 2) * nb >= 64) rows >>= 1;
        }
        if (a.prologue == MV_ATTN) {   // heads of 64 over all waves, 2 or 4 per wave; K <= 1024 keeps the gathered vector inside wave 0's staging area
            static const int attn_nw = env_int("MI355X_ATTN_PROLOGUE_NW", 8);
            const int H = a.attn->H;
            GGML_ASSERT(a.K == 1024 && H == 16);
            nw = attn_nw == 4 ? 4 : 8;
            static const int attn_grid_min = env_int("MI355X_ATTN_GRID_MIN", 128);
            int tpw = 4;
            rows = (tpw * 64 + nb - 1) / nb;
            while (rows > 1 && (a.M + rows - 1) / rows < attn_grid_min && (rows / 2) * nb >= 32) rows >>= 1;
        }
        static const int q80_nw = env_int("MI355X_Q80_NW", 8);   // A/B: Temporal 2.31 -> 2.21 ms at q8_0
        if (fmt == MVF_Q80 && nw > 4 && a.prologue != MV_ATTN) {   // 17 KB tiles: 8 waves only where xs + 8 tiles + partials fit 160 KB
            const size_t need8 = (size_t) nb * XBLK_BYTES + 8 * tile_bytes + (size_t) rows * nb * 4;
            if (q80_nw == 8 && need8 <= 158 * 1024) nw = 8;
            else {
                nw = 4;
                rows = (int) ((a.M + 511) / 512);
                while (rows * nb > 4096) rows = (rows + 1) / 2;
            }
        }
        if (a.pair_F > 0) {   // paired gate: h rows of each half per workgroup, whole tiles per half, 8 waves; the planner only asks when this fits
            GGML_ASSERT(a.M == 2 * a.pair_F && fmt != MVF_Q80);
            int h = (int) ((a.pair_F + 255) / 256);
            while ((h * nb) % 64 != 0) h++;
            GGML_ASSERT(a.pair_F % h == 0 && 2 * h * nb <= 4096);
            nw = 8; rows = 2 * h;
        }
        if (rows < 1) rows = 1;
        // small Q4_K matrices (the Depth transformer, the codebook heads): 8 lanes per super-block, no LDS tiles (WS = 1)
        static const int direct_on = env_int("MI355X_MV_DIRECT", 1), direct_passes = env_int("MI355X_MVD_PASSES", 3), direct_max_tiles = env_int("MI355X_MVD_MAX_TILES", 512);
        const bool direct = direct_on && a.pair_F == 0 && fmt == MVF_Q4K && (nw == 4 || (nw == 8 && a.prologue == MV_ATTN)) && tiles_total <= direct_max_tiles && nb <= nw * 8 * MVD_PMAX;
        if (direct) {
            int passes = direct_passes < 1 ? 1 : direct_passes > MVD_PMAX ? MVD_PMAX : direct_passes;
            if (nw == 8 && passes > 1) passes = 1;                                         // 8 waves: 64 super-blocks per pass already
            rows = passes * nw * 8 / nb;
            if (rows < 1) { rows = 1; }
            while (rows > 1 && (a.M + rows - 1) / rows < 96) rows = (rows + 1) / 2;   // keep >= ~100 workgroups in flight
            GGML_ASSERT(rows * nb <= MVD_PMAX * nw * 8);
        }
        const size_t stage_bytes = (!direct || a.prologue == MV_ATTN) ? (size_t) nw * tile_bytes : 0;
        const size_t smem = (size_t) nb * XBLK_BYTES + stage_bytes + (size_t) rows * nb * 4 + (a.prologue == MV_ATTN ? 4096 : 0);
        GGML_ASSERT(smem <= 160 * 1024);
        const int grid = a.pair_F > 0 ? (int) (a.pair_F / (rows / 2)) : (int) ((a.M + rows - 1) / rows);
        GGML_ASSERT(a.prologue != MV_RMSNORM || a.K <= nw * 1024);
        GGML_ASSERT(a.ncols == 1 && a.out_scale == nullptr && (a.prologue <= MV_GATE_SILU || a.prologue == MV_PREQ8K || a.prologue == MV_ATTN));
        void (*kern)(mv_args, int, attn_args) = nullptr;
#define MV_PICKF(NWV, F) (a.prologue == MV_RMSNORM ? matvec_q4k_kernel<MV_RMSNORM, NWV, F> : a.prologue == MV_GATE_SILU ? matvec_q4k_kernel<MV_GATE_SILU, NWV, F> \
                      : a.prologue == MV_ATTN ? matvec_q4k_kernel<MV_ATTN, NWV, F> \
                      : a.prologue == MV_PREQ8K ? matvec_q4k_kernel<MV_PREQ8K, NWV, F> : matvec_q4k_kernel<MV_PLAIN, NWV, F>)
#define MV_PICK(NWV) (fmt == MVF_Q4K ? MV_PICKF(NWV, MVF_Q4K) : fmt == MVF_Q40 ? MV_PICKF(NWV, MVF_Q40) : MV_PICKF(NWV, MVF_Q80))
#define MV_PICKD (a.prologue == MV_RMSNORM ? matvec_q4k_kernel<MV_RMSNORM, 4, MVF_Q4K, 1> : a.prologue == MV_GATE_SILU ? matvec_q4k_kernel<MV_GATE_SILU, 4, MVF_Q4K, 1> \
                  : a.prologue == MV_ATTN ? (nw == 8 ? matvec_q4k_kernel<MV_ATTN, 8, MVF_Q4K, 1> : matvec_q4k_kernel<MV_ATTN, 4, MVF_Q4K, 1>) \
                  : a.prologue == MV_PREQ8K ? matvec_q4k_kernel<MV_PREQ8K, 4, MVF_Q4K, 1> : matvec_q4k_kernel<MV_PLAIN, 4, MVF_Q4K, 1>)
        kern = direct ? MV_PICKD : nw == 12 ? MV_PICK(12) : nw == 8 ? MV_PICK(8) : MV_PICK(4);
        // the 8-wave RMS-norm launches at K <= 2048 x 2 without a residual (the Temporal linear_in in its paired form, the text head): two activation batches instead of
        // four, no residual pre-load (see the kernel's XB / RL parameters); same arithmetic
        static const int lean_on = env_int("MI355X_MV_LEAN", 1);
        if (lean_on && !direct && nw == 8 && fmt == MVF_Q4K && a.prologue == MV_RMSNORM && a.K <= 8 * 512 && !a.residual && !a.res_embed.table)
            kern = matvec_q4k_kernel<MV_RMSNORM, 8, MVF_Q4K, 0, 2, 0>;
        // the plain launches with a residual whose workgroups hold <= NW * 4 rows (the Temporal out_proj: 4 waves x 16 rows; linear_out: 8 waves x 16 rows): one residual load per thread
        if (lean_on && !direct && fmt == MVF_Q4K && a.prologue == MV_PLAIN && a.pair_F == 0 && a.residual && rows <= nw * 4 && (nw == 8 || nw == 4))
            kern = nw == 8 ? matvec_q4k_kernel<MV_PLAIN, 8, MVF_Q4K, 0, 4, 1> : matvec_q4k_kernel<MV_PLAIN, 4, MVF_Q4K, 0, 4, 1>;
        // ... and with 8 192 < K <= 12 288 at 8 waves (the Temporal linear_out, K = 11 264): SIX activation batches requested at entry, one round - with four, the last
        // 3 072 values were a second round whose loads went out only after the first round's blocks were quantised: a dependent L2 round trip inside every prologue
        static const int one_round = env_int("MI355X_MV_ONE_ROUND", 1);
        if (lean_on && one_round && !direct && fmt == MVF_Q4K && a.prologue == MV_PLAIN && a.pair_F == 0 && a.residual && rows <= nw * 4 && nw == 8 && a.K > 8 * 1024 && a.K <= 8 * 1536)
            kern = matvec_q4k_kernel<MV_PLAIN, 8, MVF_Q4K, 0, 6, 1>;
        if (smem > 64 * 1024) {   // > 64 KB of dynamic LDS needs a one-time opt-in per kernel
            static std::map<const void *, size_t> granted;
            size_t & g = granted[(const void *) kern];
            if (g < smem) { HIP_CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem)); g = smem; }
        }
        const int threads = nw * 64;
        attn_args at;
        memset(&at, 0, sizeof(at));
        if (a.prologue == MV_ATTN) at = *a.attn;   // host copy, passed by value
        if (g_mv_profile && g_mv_profile->used < g_mv_profile->capacity) {
            mv_profile::rec & r = g_mv_profile->recs[g_mv_profile->used++];
            r.bytes = a.M * a.row_bytes;
            r.variant = direct ? 1 : 0;
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, s, r.start, r.stop, 0, a, rows, at);
            return;
        }
        kern<<<grid, threads, smem, s>>>(a, rows, at);
        return;
    }
    GGML_ASSERT(a.ncols >= 1 && a.ncols <= MV_MAX_COLS);
    static const int no_reg = env_int("MI355X_MV_NOREG", 0);
    if (a.wtype == GGML_TYPE_F32 && !no_reg && (a.K == 512 || a.K == 1024 || a.K == 2048) && a.row_bytes % 16 == 0 && ((uintptr_t) a.w % 16) == 0) {
        // one row per wave until every CU has a workgroup, then two
        static const int two_min = env_int("MI355X_MVF_TWO_MIN", 4 * 2 * 256);
        const bool two = a.K <= 1024 && a.M >= two_min;
        const int grid = (int) ((a.M + (two ? 8 : 4) - 1) / (two ? 8 : 4));
        if (a.K == 512)       { if (two) matvec_f32_reg_kernel<2, 2><<<grid, 256, 0, s>>>(a); else matvec_f32_reg_kernel<2, 1><<<grid, 256, 0, s>>>(a); }
        else if (a.K == 1024) { if (two) matvec_f32_reg_kernel<4, 2><<<grid, 256, 0, s>>>(a); else matvec_f32_reg_kernel<4, 1><<<grid, 256, 0, s>>>(a); }
        else                  matvec_f32_reg_kernel<8, 1><<<grid, 256, 0, s>>>(a);
        return;
    }
    int rows_per_wave = a.K <= 1024 ? 4 : 2;
    while (rows_per_wave > 1 && (a.M + 4 * rows_per_wave - 1) / (4 * rows_per_wave) < 256) rows_per_wave >>= 1;
    const int grid = (int) ((a.M + 4 * rows_per_wave - 1) / (4 * rows_per_wave));
    const size_t smem = (size_t) a.K * 4 * (size_t) a.ncols;
    switch (a.wtype) {
        case GGML_TYPE_F32:  matvec_f_kernel<GGML_TYPE_F32><<<grid, 256, smem, s>>>(a, rows_per_wave); break;
        case GGML_TYPE_F16:  matvec_f_kernel<GGML_TYPE_F16><<<grid, 256, smem, s>>>(a, rows_per_wave); break;
        case GGML_TYPE_BF16: matvec_f_kernel<GGML_TYPE_BF16><<<grid, 256, smem, s>>>(a, rows_per_wave); break;
        default: GGML_ABORT("k_matvec: unsupported weight type");
    }
}

// ---------------------------------------------------------------------------------------------------
// single-token attention over the ring KV cache (one workgroup per head)
//
// Reproduces the node sequence of moshi_streaming_multihead_attention for T = 1
// (src/moshi/modules/transformer.h:543-576, rope.h:33-128, torch.h:225-237) with ggml's CPU numerics:
//   q, k interleaved pairs -> rotated, stored de-interleaved [re | im];
//   K, V rows written as BF16 at ring slot index[0];
//   scores = <bf16 K, bf16(q)> (double accumulate) ; softmax(score * scale + mask) ; p rounded to BF16;
//   out = sum_c bf16 V[c] * p[c] (double accumulate).
// Masked slots (mask = -inf) contribute exactly 0, so only un-masked slots are read from HBM; the
// result equals the reference's full-capacity soft_max (README.md:55-57).
// ---------------------------------------------------------------------------------------------------
#define ATTN_BODY_LOG
#include "hip_attn_body.h"

template <bool SPLIT, int NWA>
__global__ void __launch_bounds__(NWA * 64) __attribute__((amdgpu_waves_per_eu(2)))   // >= 2 workgroups per CU: the split grid (<= 512) is resident
attn_decode_kernel(attn_args a_in, attn_split_ws w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = SPLIT ? w.S : 1;
    attn_decode_body<SPLIT, NWA, AT_PLAIN>(a_in, w, smem, SPLIT ? (int) blockIdx.x / S : (int) blockIdx.x, SPLIT ? (int) blockIdx.x % S : 0, (int) blockIdx.y);
}

static bool attn_use_split(const attn_args & a) { return a.T == 1 && a.n_groups <= 1 && a.D == 128 && a.C >= ATTN_SPLIT_MIN_C; }
// waves per split workgroup: 4 (ranges of 128 / 256 slots: up to 12 workgroups per head at 3 000 slots = 384 on 256 CUs) or 8 (192 / 384 slots: 8 per head = one per CU)
static int attn_split_nw() { static const int v = env_int("MI355X_ATTN_SPLIT_NW", ATTN_SPLIT_NW_DEFAULT); return v == 8 ? 8 : 4; }
static int attn_split_slots() { static const int v = env_int("MI355X_ATTN_SLOTS", attn_split_nw() == 8 ? 192 : ATTN_SPLIT_SLOTS); return v < 64 ? 64 : v > 32 * attn_split_nw() ? 32 * attn_split_nw() : v; }
size_t k_attn_decode_ws_size(const attn_args & a) {
    if (!attn_use_split(a)) return 0;
    const size_t S = (size_t) (a.C + attn_split_slots() - 1) / attn_split_slots();
    return 256 + (((size_t) a.H * 4 + 255) & ~(size_t) 255) + (size_t) a.H * a.C * 8 + (size_t) a.H * S * a.D * 16;   // per-head sequence numbers | score granules | partial-output granules
}
static size_t attn_smem_bytes(const attn_args & a, bool split, bool wide) {
    const int Tg = a.n_groups > 1 ? ATTN_MAX_T : a.T;
    return (size_t) a.C * 4 + (size_t) Tg * a.D * 4 * 3 + (size_t) (split ? attn_split_nw() : wide ? ATTN_NW_WIDE : ATTN_NW_BASE) * 64 * 8 * 8 + 16 + (size_t) Tg * a.C * 4 + (size_t) env_int("MI355X_ATTN_LDS_PAD", 0);
}
// The split kernel's workgroups of one head wait for each other (scores, partial outputs): the planner only hands it a workspace - i.e. only splits - when
// the WHOLE grid (H x ceil(C / slots) workgroups) can be resident on the compute units the stream may use; otherwise a head stays one workgroup (correct at
// every length, slower at long context). HIP promises no dispatch order, so nothing weaker than full residency is assumed.
bool k_attn_split_resident(const attn_args & a, int usable_cus) {
    if (!attn_use_split(a)) return false;
    if (env_int("MI355X_ATTN_SPLIT_FORCE", 0)) return true;
    const int nw = attn_split_nw();
    const size_t smem = attn_smem_bytes(a, true, false);
    const void * fn = nw == 8 ? (const void *) attn_decode_kernel<true, ATTN_NW_WIDE> : (const void *) attn_decode_kernel<true, ATTN_NW_BASE>;
    if (smem > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem) != hipSuccess) return false;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nw * 64, smem) != hipSuccess) return false;
    const long long grid = (long long) a.H * ((a.C + attn_split_slots() - 1) / attn_split_slots());
    const bool ok = (long long) per_cu * usable_cus >= grid;
    if (env_int("MI355X_CHAIN_VERBOSE", 0)) fprintf(stderr, "split attention: grid %lld, %d per CU x %d CUs -> %s\n", grid, per_cu, usable_cus, ok ? "split" : "one workgroup per head");
    return ok;
}
// one head per workgroup of 8 waves, single token, ring of <= 64 slots of 64 (the tts-shaped Depth transformer): attn_ring64_body
__global__ void __launch_bounds__(512) attn_ring64_kernel(attn_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_ring64_body<AT_PLAIN>(a, smem, (int) blockIdx.x);
}
static bool attn_ring64_shape(const attn_args & a) {
    static const int on = env_int("MI355X_ATTN_RING64", 1);
    return on && a.D == 64 && a.T == 1 && a.n_groups <= 1 && !a.write_only && a.C > 8 && a.C <= 64 && a.out_ts >= 0;
}
// one (head, query row) per workgroup of 8 waves, T <= 2 new tokens, ring of 65 .. 256 slots of 64 (the codec transformers): attn_ring256_body
__global__ void __launch_bounds__(512) attn_ring256_kernel(attn_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_ring256_body<AT_PLAIN>(a, smem, (int) blockIdx.x, (int) blockIdx.y);
}
static bool attn_ring256_shape(const attn_args & a) {
    static const int on = env_int("MI355X_ATTN_RING256", 1);
    return on && a.D == 64 && a.T >= 1 && a.T <= 2 && a.n_groups <= 1 && !a.write_only && a.C > 64 && a.C <= 256 && a.out_ts >= 0;
}
void k_attn_decode(hipStream_t s, const attn_args & a, void * ws, unsigned * err) {
    if (attn_ring64_shape(a)) { attn_ring64_kernel<<<a.H, 512, ATTN_RING64_SMEM, s>>>(a); return; }
    if (attn_ring256_shape(a)) { attn_ring256_kernel<<<dim3((unsigned) a.H, (unsigned) a.T), 512, ATTN_RING256_SMEM, s>>>(a); return; }
    const int Tg = a.n_groups > 1 ? ATTN_MAX_T : a.T;   // rows per workgroup
    GGML_ASSERT(a.D % 8 == 0 && 64 % (a.D / 8) == 0 && a.D <= 512 && a.T >= 1 && Tg <= ATTN_MAX_T && Tg * a.D <= 2 * ATTN_NW_BASE * 64);
    GGML_ASSERT(a.n_groups <= 1 || a.n_groups == (a.T + 3) / 4);
    static const int wide_on = env_int("MI355X_ATTN_WIDE", 1);
    const bool wide = wide_on && a.D <= 64 && a.C >= 128;
    const bool split = ws && attn_use_split(a);
    const int split_nw = attn_split_nw();
    const size_t smem = attn_smem_bytes(a, split, wide);
    GGML_ASSERT(smem <= 160 * 1024);
    if (smem > 64 * 1024) {   // > 64 KB of dynamic LDS needs a one-time opt-in per kernel
        static size_t granted[4] = { 0, 0, 0, 0 };
        const int which = split ? (split_nw == 8 ? 3 : 1) : wide ? 2 : 0;
        if (granted[which] < smem) {
            HIP_CHECK(hipFuncSetAttribute(which == 3 ? (const void *) attn_decode_kernel<true, ATTN_NW_WIDE> : which == 1 ? (const void *) attn_decode_kernel<true, ATTN_NW_BASE> : which == 2 ? (const void *) attn_decode_kernel<false, ATTN_NW_WIDE> : (const void *) attn_decode_kernel<false, ATTN_NW_BASE>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
            granted[which] = smem;
        }
    }
    static const int single_max = env_int("MI355X_ATTN_SINGLE_MAX", ATTN_SINGLE_MAX), big_min = env_int("MI355X_ATTN_BIG_MIN", ATTN_SPLIT_BIG_MIN);
    attn_split_ws w = { nullptr, nullptr, nullptr, 1, err, ATTN_SPLIT_SLOTS, single_max, big_min };
    if (split) {
        const int S = (a.C + attn_split_slots() - 1) / attn_split_slots();
        w.slots = attn_split_slots();
        GGML_ASSERT(a.D == 128 && "split prefetch depth is sized for 16 slots per pass");
        char * p = (char *) ws;
        w.seq = (unsigned *) p; p += ((size_t) a.H * 4 + 255) & ~(size_t) 255;
        w.gscores = (unsigned long long *) p; p += (size_t) a.H * a.C * 8;
        w.gpart = (unsigned long long *) p;
        w.S = S;
        if (split_nw == 8) attn_decode_kernel<true, ATTN_NW_WIDE><<<a.H * S, ATTN_NW_WIDE * 64, smem, s>>>(a, w);
        else               attn_decode_kernel<true, ATTN_NW_BASE><<<a.H * S, ATTN_NW_BASE * 64, smem, s>>>(a, w);
        return;
    }
    static const int row_split_on = env_int("MI355X_ATTN_ROW_SPLIT", 1);
    attn_args b = a;
    b.row_split = row_split_on && a.n_groups <= 1 && a.T >= 2 && a.T <= ATTN_MAX_T && !a.write_only ? 1 : 0;
    const dim3 grid((unsigned) a.H, (unsigned) (a.n_groups > 1 ? a.n_groups : b.row_split ? a.T : 1));
    if (wide) attn_decode_kernel<false, ATTN_NW_WIDE><<<grid, ATTN_NW_WIDE * 64, smem, s>>>(b, w);
    else      attn_decode_kernel<false, ATTN_NW_BASE><<<grid, ATTN_NW_BASE * 64, smem, s>>>(b, w);
}

// ---------------------------------------------------------------------------------------------------
// merged launches of the Temporal layer: shared definitions (the kernel itself: inproj_attn_kernel below)
// ---------------------------------------------------------------------------------------------------
#define FOLD_NW 8
#define FOLD_GRID 256
#define FOLD_PMAX 4      // passes of 64 super-blocks a workgroup holds in registers: rows_wg * nb <= 256
struct fold_ws { unsigned long long * gbuf; unsigned long long * arrive; unsigned * err; };   // arrive: workgroups that have finished, over all launches so far
#if defined(MV_LOG)
#define FD_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_fold_log[fd_id][i] = t_; } } while (0)
__device__ unsigned long long g_fold_log[4096][8];
__device__ unsigned g_fold_launch;
extern "C" __attribute__((visibility("default"))) int mi355x_fold_log_read(unsigned long long * out, int max_records) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fold_launch), 4) != hipSuccess) return -1;
    const int m = (int) (n < 4096u ? n : 4096u) < max_records ? (int) (n < 4096u ? n : 4096u) : max_records;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fold_log), (size_t) m * 8 * 8) != hipSuccess) return -1;
    const unsigned z = 0; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_fold_launch), &z, 4);
    return (int) n;
}
#else
#define FD_STAMP(i) do {} while (0)
#endif

static int fold_split_S(const attn_args & at) { return FOLD_GRID / at.H; }
// may the attention stage split a head over its S parts? (the 8-wave split geometry: ranges of `slots` up to big_min live slots, twice that beyond)
static bool fold_use_split(const attn_args & at) {
    if (!attn_use_split(at)) return false;
    static const int big_min = env_int("MI355X_ATTN_BIG_MIN", ATTN_SPLIT_BIG_MIN);
    const int S = fold_split_S(at), slots = env_int("MI355X_FOLD_SLOTS", 192);
    return (int64_t) S * slots >= (at.C < big_min ? at.C : big_min) && (int64_t) 2 * S * slots >= at.C;
}
// ---------------------------------------------------------------------------------------------------
// in_proj + attention as ONE launch (the Temporal layer, transformer.h:449-576: norm1 -> in_proj -> RoPE -> ring write -> SDPA)
//
// As its own launch the attention pays a kernel boundary (2.3 - 2.6 us) and one cold memory round trip for q / k / v before its first useful instruction,
// 7.3 us in all for almost no bytes at short context. Here it is the TAIL of the in_proj launch: the mat-vec's 256 workgroups are dealt out head-major -
// workgroup b is part b / H of head b % H and owns rows [16 j, 16 j + 16) of that head's q, k AND v (three row segments of whole tiles; 48 rows = the
// 12 tiles the plain mapping gives it too) - and run matvec_q4k_kernel<RMSNORM, 8 waves>'s arithmetic on them to the bit (same Q8_K blocks, same tile
// dots, same 16-lane row sums). Every row sum is written to the in_proj node's storage AND published as an 8-byte {tag, value} granule; the head's
// workgroups then enter attn_decode_body with the granules as their q / k / v source (AT_GQKV): the ring rows, the mask and the RoPE table are requested
// first and land while the granules are polled, so the hand-off costs one round trip that the separate launch spent on its cold loads anyway. Split
// geometry, hand-offs and arithmetic are the attention kernel's own (S = 256 / H parts per head, 8 waves). Workgroups without attention work leave at once.
// tag = launches completed so far + 1, derived by every workgroup from ONE counter of finished workgroups (see the kernel's first lines).
// The head's parts wait for each other: the grid must be resident (k_inproj_attn_supported).
// ---------------------------------------------------------------------------------------------------
// NJ: activation batches of NW * 256 values a thread loads (K <= NJ * NW * 256: 2 at K = 4096, 1 at K = 2048). Until round 6 every thread issued four x and four
// alpha loads whatever K was - at K = 4096 half of them clamped duplicates whose results were thrown away: 4 of a wave's 17 vector-memory instructions, ~26 cycles
// of issue each in front of the weight tile, in every one of the frame's 32 launches.
template <bool SPLIT, int NJ>
__global__ void __launch_bounds__(FOLD_NW * 64) inproj_attn_kernel(mv_args a, attn_args at, attn_split_ws w, fold_ws f, attn_gqkv gq, int seg_rows, int64_t seg_stride) {
    constexpr int NW = FOLD_NW, NLOAD = 9, SB = 144, TILE = 64 * SB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double sh_red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, H = at.H, S = (int) gridDim.x / H;
    const int h = b % H, part_i = b / H;
    const int blk = h * S + part_i;                       // head-major: rows [blk * seg_rows, +seg_rows) of each of the three segments
#if defined(MV_LOG)
    __shared__ unsigned fd_id_s;
    if (b == 0 && tid == 0) fd_id_s = atomicAdd(&g_fold_launch, 1u) & 4095u;
    __syncthreads();
    const unsigned fd_id = b == 0 ? fd_id_s : 0u;
#endif
    FD_STAMP(0);
    // The launch's tag: launches completed so far + 1 = (workgroups that have finished, over all launches) / grid + 1. Every workgroup adds itself to
    // that count as its LAST act (a non-returning add: nothing waits for it), so whatever a workgroup of launch n reads - it has not added itself yet -
    // lies in [n G, (n + 1) G): every workgroup of a launch derives the same tag, late starters included, and no word is ever bumped by anybody.
    const unsigned tag = (unsigned) (__hip_atomic_load(f.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (unsigned long long) FOLD_GRID) + 1u;   // (the grid is FOLD_GRID workgroups, always)
    const int nb = (int) (a.K / 256), K = (int) a.K;
    const int rows = 3 * seg_rows, nblk = rows * nb, ntiles = nblk >> 6, tiles_seg = (seg_rows * nb) >> 6;   // (whole tiles per segment: host-checked)
    xblk * xs = (xblk *) smem;
    char * stage = smem + nb * XBLK_BYTES + wave * TILE;
    float * part = (float *) (smem + nb * XBLK_BYTES + NW * TILE);
    auto tile_src = [&](int t) {   // first 16-byte chunk of tile t
        const int sg = t / tiles_seg;
        return (const u32x4 *) (a.w + ((int64_t) sg * seg_stride + (int64_t) blk * seg_rows) * a.row_bytes) + (int64_t) (t - sg * tiles_seg) * (NLOAD * 64);
    };
    // ---- mat-vec, phase 1: activation loads, then the first weight tile (matvec_q4k_kernel's order)
    float4 xv[NJ], aux[NJ];
    bool ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int e0 = j * (NW * 256) + tid * 4;
        ok[j] = e0 < K;
        const int e = ok[j] ? e0 : K - 4;
        xv[j] = *(const float4 *) (a.x + e);
        aux[j] = *(const float4 *) (a.alpha + e);
    }
    __builtin_amdgcn_sched_barrier(0);
#if defined(MV_HEAD_EXP) && MV_HEAD_EXP == 1
    __builtin_amdgcn_s_barrier();
#elif defined(MV_HEAD_EXP) && MV_HEAD_EXP == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    u32x4 r[NLOAD];
    int t = wave;
    {
        const bool has_tile = t < ntiles;
        const u32x4 * src = tile_src(has_tile ? t : 0);
#pragma unroll
        for (int i = 0; i < NLOAD; i++) r[i] = __builtin_nontemporal_load(src + (has_tile ? i * 64 + lane : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: alpha * rms_norm(x) -> Q8_K blocks (K <= 4096: one batch, the whole vector in registers)
    {
        float v[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float z = ok[j] ? 1.f : 0.f;
            v[j][0] = xv[j].x * z; v[j][1] = xv[j].y * z; v[j][2] = xv[j].z * z; v[j][3] = xv[j].w * z;
        }
        double acc = 0;
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc += (double) (v[j][k] * v[j][k]);
        acc = wave_allsum_f64(acc);
        if (lane == 0) sh_red[wave] = acc;
        __syncthreads();
        double tot = 0;
#pragma unroll
        for (int w_ = 0; w_ < NW; w_++) tot += sh_red[w_];
        const float mean = (float) (tot / (double) K);
        const float scale = 1.0f / sqrtf(mean + a.eps);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float al[4] = { aux[j].x, aux[j].y, aux[j].z, aux[j].w };
#pragma unroll
            for (int k = 0; k < 4; k++) v[j][k] = al[k] * (v[j][k] * scale);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (!ok[j]) continue;
            const int bq = j * NW + wave;
            if (a.x_out != nullptr && b == 0) *(float4 *) (a.x_out + j * (NW * 256) + tid * 4) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
            quantize_block_q8k(xs + bq, v[j], lane);
        }
    }
    __syncthreads();
    FD_STAMP(1);
    // ---- phase 3: tiles registers -> LDS image -> one super-block per lane (next tile requested first)
    for (; t < ntiles; t += NW) {
#pragma unroll
        for (int i = 0; i < NLOAD; i++) ((u32x4 *) stage)[i * 64 + lane] = r[i];
        {
            const int tn = t + NW < ntiles ? t + NW : ntiles - 1;
            const u32x4 * src = tile_src(tn);
#pragma unroll
            for (int i = 0; i < NLOAD; i++) r[i] = __builtin_nontemporal_load(src + i * 64 + lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int bi = t * 64 + lane;
        const xblk * xb = xs + (bi % nb);
        part[bi] = q4k_q8k_block_dot((const block_q4_K *) (stage + lane * SB), xb->q, xb->bsums, xb->d);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    FD_STAMP(2);
    // ---- phase 4: fixed-order row sums; each row goes to the in_proj node's storage and, tagged, to the granule table the attention stage polls
    for (int rr = tid >> 4; rr < rows; rr += NW * 4) {
        float sum = 0.f;
        for (int j = tid & 15; j < nb; j += 16) sum += part[rr * nb + j];
        sum = row16_allsum_f32(sum);
        if ((tid & 15) == 0) {
            const int sg = rr / seg_rows;
            const int64_t row = (int64_t) sg * seg_stride + (int64_t) blk * seg_rows + (rr - sg * seg_rows);
            a.y[row] = sum;
            __hip_atomic_store(f.gbuf + row, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();   // the mat-vec's LDS is the attention stage's from here on
    FD_STAMP(3);
    // ---- the attention of head h, part part_i, its new rows taken from the granules
    attn_decode_body<SPLIT, NW, AT_GQKV>(at, w, smem, h, part_i, 0, tag, gq);
    FD_STAMP(4);
    // this workgroup is done: count it in (the tag was read at entry and is in use since - the asm pins that order for the compiler)
    if (tid == 0) { asm volatile("" :: "v"(tag) : "memory"); (void) __hip_atomic_fetch_add(f.arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    FD_STAMP(5);
}

typedef void (*fold_kernel_t)(mv_args, attn_args, attn_split_ws, fold_ws, attn_gqkv, int, int64_t);
static fold_kernel_t fold_kernel_fn(bool split, int K) {
    const int nj = (K + FOLD_NW * 256 - 1) / (FOLD_NW * 256);
    GGML_ASSERT(nj >= 1 && nj <= 2);   // K <= 4096 (k_inproj_attn_supported)
    return split ? (nj == 1 ? inproj_attn_kernel<true, 1> : inproj_attn_kernel<true, 2>) : (nj == 1 ? inproj_attn_kernel<false, 1> : inproj_attn_kernel<false, 2>);
}
static const void * fold_kernel(bool split, int K) { return (const void *) fold_kernel_fn(split, K); }
static size_t inproj_attn_smem(const mv_args & a, const attn_args & at, int seg_rows) {
    const size_t attn = (size_t) at.C * 4 + (size_t) at.T * at.D * 4 * 3 + (size_t) FOLD_NW * 64 * 8 * 8 + 16 + (size_t) at.T * at.C * 4;
    const size_t mv = (size_t) (a.K / 256) * XBLK_BYTES + (size_t) FOLD_NW * 64 * 144 + (size_t) 3 * seg_rows * (a.K / 256) * 4;
    return ((attn > mv ? attn : mv) + 15) & ~(size_t) 15;
}
// `a`: the in_proj mat-vec (alpha * rms_norm(x) prologue, Q4_K, no epilogue) whose output holds `at`'s q | k | v as three segments of H x D rows
bool k_inproj_attn_supported(const mv_args & a, const attn_args & at, int usable_cus) {
    static const int on = env_int("MI355X_INPROJ_ATTN", 1);
    if (!on) return false;
    if (a.wtype != GGML_TYPE_Q4_K || a.ncols != 1 || a.prologue != MV_RMSNORM || a.pair_F || a.out_scale || a.out_act || a.ticket || a.res_embed.table || a.residual ||
        a.argmax_out[0] || a.argmax_out[1]) return false;
    if (a.K % 256 != 0 || a.K > 4096 || a.K < 256 || a.row_bytes != (a.K / 256) * 144 || ((uintptr_t) a.w & 15)) return false;
    if (at.C <= 32 || at.T != 1 || at.n_groups > 1 || at.H < 1 || FOLD_GRID % at.H != 0 || at.D != 128) return false;
    const int64_t HD = (int64_t) at.H * at.D;
    if (a.M != 3 * HD || HD % FOLD_GRID != 0) return false;
    const int seg_rows = (int) (HD / FOLD_GRID), nb = (int) (a.K / 256);
    if ((seg_rows * nb) % 64 != 0 || 3 * seg_rows * nb > 4096 || (at.D % seg_rows) != 0 || (seg_rows & 1)) return false;   // whole tiles per segment; a part holds whole RoPE pairs of ONE head
    // q | k | v are the three H x D segments of y, head-major, one token
    if (at.q != a.y || at.k != a.y + HD || at.v != a.y + 2 * HD || at.q_hs != at.D || at.k_hs != at.D || at.v_hs != at.D) return false;
    // a ring the stand-alone kernel would split over workgroups (C >= ATTN_SPLIT_MIN_C) but the merged launch cannot (fold_use_split: more than 2 x S x slots,
    // or an MI355X_FOLD_SLOTS / head-count combination that fails its test) would run here as ONE workgroup per head scanning the whole ring while the other
    // 224 leave - correct, and a silent long-context slowdown: such rings keep the separate split attention launch
    if (attn_use_split(at) && !fold_use_split(at)) return false;
    const size_t smem = inproj_attn_smem(a, at, seg_rows);
    if (smem > 160 * 1024) return false;
    const void * fn = fold_kernel(fold_use_split(at), (int) a.K);
    if (smem > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem) != hipSuccess) return false;
    if (env_int("MI355X_FOLD_FORCE", 0)) return true;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, FOLD_NW * 64, smem) != hipSuccess) return false;
    return (long long) per_cu * usable_cus >= FOLD_GRID;   // the parts of a head wait for each other: the whole grid must be resident
}
size_t k_inproj_attn_ws_size(const mv_args & a, const attn_args & at) {
    size_t n = 256 + (size_t) a.M * 8;
    if (fold_use_split(at)) n += (((size_t) at.H * 4 + 255) & ~(size_t) 255) + (size_t) at.H * at.C * 8 + (size_t) at.H * fold_split_S(at) * at.D * 16;
    return n;
}
// ws: k_inproj_attn_ws_size bytes, zeroed once by the caller (launch counter, arrival counter, granule tags, per-head sequence numbers)
void k_inproj_attn(hipStream_t s, const mv_args & a, const attn_args & at, void * ws, unsigned * err) {
    static const int single_max = env_int("MI355X_ATTN_SINGLE_MAX", ATTN_SINGLE_MAX), big_min = env_int("MI355X_ATTN_BIG_MIN", ATTN_SPLIT_BIG_MIN);
    char * p = (char *) ws;
    fold_ws f = { (unsigned long long *) (p + 256), (unsigned long long *) p, err };
    p += 256 + (size_t) a.M * 8;
    const bool split = fold_use_split(at);
    attn_split_ws w = { nullptr, nullptr, nullptr, 1, err, ATTN_SPLIT_SLOTS, single_max, big_min };
    if (split) {
        w.slots = env_int("MI355X_FOLD_SLOTS", 192);
        w.S = fold_split_S(at);
        w.seq = (unsigned *) p; p += ((size_t) at.H * 4 + 255) & ~(size_t) 255;
        w.gscores = (unsigned long long *) p; p += (size_t) at.H * at.C * 8;
        w.gpart = (unsigned long long *) p;
    }
    const int64_t HD = (int64_t) at.H * at.D;
    const int seg_rows = (int) (HD / FOLD_GRID);
    const attn_gqkv gq = { f.gbuf, 0, HD, 2 * HD, err };   // granule g holds row g of y; a.q / a.k / a.v start at rows 0 / HD / 2 HD
    const size_t smem = inproj_attn_smem(a, at, seg_rows);
    const fold_kernel_t fold_fn = fold_kernel_fn(split, (int) a.K);
    if (g_mv_profile && g_mv_profile->used < g_mv_profile->capacity) {   // profile mode: events attached to the dispatch, like k_matvec's
        mv_profile::rec & r = g_mv_profile->recs[g_mv_profile->used++];
        r.bytes = a.M * a.row_bytes;
        r.variant = 2;
        hipExtLaunchKernelGGL(fold_fn, dim3(FOLD_GRID), dim3(FOLD_NW * 64), smem, s, r.start, r.stop, 0, a, at, w, f, gq, seg_rows, HD);
        return;
    }
    fold_fn<<<FOLD_GRID, FOLD_NW * 64, smem, s>>>(a, at, w, f, gq, seg_rows, HD);
}

// ---------------------------------------------------------------------------------------------------
// cross-attention over the cached condition (tts): the node sequence mul_mat(K, q) -> soft_max_ext(scale, no mask) ->
// mul_mat(cont(transpose(V)), p) -> cont(permute) with the per-op kernels' arithmetic (float products summed in double, expf of
// scale * s - max, probabilities scaled by float(1 / sum)), without materialising the transposed V every frame
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cross_attn_kernel(xattn_args a) {
    extern __shared__ float xsm[];          // [D] q | [Tc] scores -> probabilities
    __shared__ float sh_f[4];
    __shared__ double sh_d[4];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, Tc = a.Tc;
    float * qs = xsm, * sc = xsm + D;
    for (int d = tid; d < D; d += 256) qs[d] = a.q[(int64_t) h * D + d];
    __syncthreads();
    const char * kh = a.k + (int64_t) h * a.k_nb2, * vh = a.v + (int64_t) h * a.v_nb2;
    // scores: a wave takes four condition rows at a time so that their loads travel together (one round trip per four rows, not per row)
    for (int t0 = wave * 4; t0 < Tc; t0 += 16) {
        double acc[4] = { 0, 0, 0, 0 };
        for (int d = lane; d < D; d += 64) {
            float kv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) kv[u] = *(const float *) (kh + (int64_t) (t0 + u < Tc ? t0 + u : Tc - 1) * a.k_nb1 + (int64_t) d * 4);
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] += (double) (kv[u] * qs[d]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double r = wave_allsum_f64(acc[u]);
            if (lane == 0 && t0 + u < Tc) sc[t0 + u] = (float) r;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = tid; t < Tc; t += 256) mx = fmaxf(mx, sc[t] * a.scale);
    mx = wave_allmax_f32(mx);
    if (lane == 0) sh_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh_f[0], sh_f[1]), fmaxf(sh_f[2], sh_f[3]));
    double sum = 0;
    for (int t = tid; t < Tc; t += 256) { const float e = expf(sc[t] * a.scale - mx); sc[t] = e; sum += (double) e; }
    sum = wave_allsum_f64(sum);
    if (lane == 0) sh_d[wave] = sum;
    __syncthreads();
    const float inv = (float) (1.0 / (sh_d[0] + sh_d[1] + sh_d[2] + sh_d[3]));
    for (int t = tid; t < Tc; t += 256) sc[t] *= inv;
    __syncthreads();
    // out[d] = sum_t p_t V[d, t] in t order; the Tc rows are split over the 256 / D thread groups (partials through LDS), eight loads in flight
    {
        const int G = 256 / D > 0 ? 256 / D : 1, grp = tid / D, d = tid - grp * D;
        double acc = 0;
        if (grp < G) {
            const int per = (Tc + G - 1) / G, tb = grp * per, te = tb + per < Tc ? tb + per : Tc;
            for (int t = tb; t < te; t += 8) {
                float vv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) vv[u] = *(const float *) (vh + (int64_t) (t + u < te ? t + u : te - 1) * a.v_nb1 + (int64_t) d * 4);
#pragma unroll
                for (int u = 0; u < 8; u++) if (t + u < te) acc += (double) (vv[u] * sc[t + u]);
            }
        }
        double * part = (double *) (xsm + ((D + Tc + 1) & ~1));   // [G][D]
        if (grp < G) part[grp * D + d] = acc;
        __syncthreads();
        if (tid < D) {
            double tot = 0;
            for (int gi = 0; gi < G; gi++) tot += part[gi * D + tid];
            a.out[(int64_t) h * D + tid] = (float) tot;
        }
    }
}
void k_cross_attn(hipStream_t s, const xattn_args & a) {
    GGML_ASSERT(a.D <= 256);
    const size_t smem = (size_t) ((a.D + a.Tc + 1) & ~1) * 4 + (size_t) (256 / a.D > 0 ? 256 / a.D : 1) * a.D * 8;
    GGML_ASSERT(smem <= 64 * 1024);
    cross_attn_kernel<<<a.H, 256, smem, s>>>(a);
}

// ---------------------------------------------------------------------------------------------------
// embedding sum: out = (((e_0 + e_1) + e_2) + ...), e_i = dequant(table_i[row idx_i]) * scale_i
// (src/moshi/models/lm.h:555-584, lm_utils.h:157-170); same left-to-right float order as the graph
// ---------------------------------------------------------------------------------------------------
// TYPE: the ggml type every table shares (-1: mixed). Three dependent round trips in all - the 17 row indices and scales (uniform addresses: scalar loads),
// every term's row element, the store - which takes loads that sit in NO branch: dequant_elem's per-type switch (and, inside it, the low / high nibble
// choice of a Q4_0 byte) made hipcc wait for each term's loads before it issued the next (24 x s_waitcnt vmcnt(0) in a row: 21 us at the head of every
// Temporal graph, profiles/r03_bench_kernel_trace_summary.txt). With the type a template parameter the element of every term is ONE scale load + ONE
// quant load whose addresses are plain arithmetic, all requested before the first is looked at. 64-thread workgroups: 64 of them for a 4096-wide row.
template <int TYPE, int NMAX>
__global__ void __launch_bounds__(64) embed_sum_kernel(embed_sum_args a) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.K) return;
    int64_t r[NMAX];
    float sc[NMAX], v[NMAX];
#pragma unroll
    for (int t = 0; t < NMAX; t++) {
        const embed_src & e = a.src[t < a.n ? t : 0];
        r[t] = *e.index;
        sc[t] = e.scale ? *e.scale : 1.f;
    }
    if (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q8_0) {
        constexpr int BB = TYPE == GGML_TYPE_Q4_0 ? 18 : 34;
        const int64_t boff = (i >> 5) * BB;
        const int j = (int) (i & 31);
        const int64_t qoff = boff + 2 + (TYPE == GGML_TYPE_Q4_0 ? (j & 15) : j);
        uint16_t dh[NMAX]; uint8_t qb[NMAX];
#pragma unroll
        for (int t = 0; t < NMAX; t++) {
            const embed_src & e = a.src[t < a.n ? t : 0];
            const int64_t row = (r[t] < 0 || r[t] >= e.n_rows) ? 0 : r[t];
            const char * base = e.table + row * e.row_bytes;
            dh[t] = *(const uint16_t *) (base + boff);
            qb[t] = *(const uint8_t *) (base + qoff);
        }
        __builtin_amdgcn_sched_barrier(0);   // every load above is requested before any is consumed
#pragma unroll
        for (int t = 0; t < NMAX; t++) {
            if (TYPE == GGML_TYPE_Q4_0) { const int q = j < 16 ? (qb[t] & 0x0F) : (qb[t] >> 4); v[t] = (q - 8) * h2f(dh[t]); }
            else v[t] = (int8_t) qb[t] * h2f(dh[t]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < NMAX; t++) {
            const embed_src & e = a.src[t < a.n ? t : 0];
            if (r[t] < 0 || r[t] >= e.n_rows) r[t] = 0;
            v[t] = dequant_elem(e.table + r[t] * e.row_bytes, TYPE >= 0 ? TYPE : e.type, i);
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < NMAX; t++) {
        if (t < a.n) {
            const float x = a.src[t].scale ? v[t] * sc[t] : v[t];
            acc = t == 0 ? x : acc + x;
        }
    }
    a.out[i] = acc;
}
void k_embed_sum(hipStream_t s, const embed_sum_args & a) {
    int type = a.n > 0 ? a.src[0].type : -1;
    for (int t = 1; t < a.n; t++) if (a.src[t].type != type) type = -1;
    const int grid = (int) ((a.K + 63) / 64);
    // (every loop over the terms is unrolled to the instantiation's bound: sums of up to 24 terms - moshika 17, PersonaPlex 17 - keep the 24-term kernel)
#define EMBED_LAUNCH(T) do { if (a.n <= 8) embed_sum_kernel<T, 8><<<grid, 64, 0, s>>>(a); else if (a.n <= 24) embed_sum_kernel<T, 24><<<grid, 64, 0, s>>>(a); else embed_sum_kernel<T, EMBED_SUM_MAX><<<grid, 64, 0, s>>>(a); } while (0)   // (8: the RVQ decode sums - 7 + 1 rows - ran 16 dead iterations per stage in the 24-term instance)
    switch (type) {
        case GGML_TYPE_Q4_0: EMBED_LAUNCH(GGML_TYPE_Q4_0); break;
        case GGML_TYPE_Q8_0: EMBED_LAUNCH(GGML_TYPE_Q8_0); break;
        case GGML_TYPE_F32:  EMBED_LAUNCH(GGML_TYPE_F32); break;
        case GGML_TYPE_BF16: EMBED_LAUNCH(GGML_TYPE_BF16); break;
        case GGML_TYPE_F16:  EMBED_LAUNCH(GGML_TYPE_F16); break;
        default:             EMBED_LAUNCH(-1); break;
    }
#undef EMBED_LAUNCH
}

// ---------------------------------------------------------------------------------------------------
// residual-VQ encode level: argmax_c 1 / (||e_c - x||^2 + 1), then x <- x - e_best
// ---------------------------------------------------------------------------------------------------
// The reference materialises (e - x) for all 2048 centroids, squares, sum_rows (double accumulate), adds 1, takes the
// reciprocal and argmaxes (ggml's CPU argmax: the last maximum wins). Here every wave scores VQ_CPW centroids (row loads requested up front), the
// workgroup keeps its best candidate, and the last workgroup to arrive (device-scope counter) merges the candidates, writes
// the code and updates the residual - one launch per level instead of fifteen.
#define VQ_CPW 4
__global__ void __launch_bounds__(256) vq_level_kernel(vq_level_args a) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D;   // 256: one float4 per lane
    const int c0 = (blockIdx.x * 4 + wave) * VQ_CPW;
    f32x4 e[VQ_CPW];
#pragma unroll
    for (int u = 0; u < VQ_CPW; u++) {
        const int c = c0 + u < a.NC ? c0 + u : a.NC - 1;
        e[u] = *(const f32x4 *) (a.emb + (int64_t) c * a.emb_row_bytes + lane * 16);
    }
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = *(const float *) (a.resid + (int64_t) (lane * 4 + j) * a.resid_stride);
    const float addc = a.add_c[0];
    float best = -INFINITY; int bi = -1;
#pragma unroll
    for (int u = 0; u < VQ_CPW; u++) {
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { const float d = e[u][j] - x[j]; acc += (double) (d * d); }
        acc = wave_allsum_f64(acc);
        const int c = c0 + u;
        if (c < a.NC) {
            const float v = a.num[c] / ((float) acc + addc);
            if (v >= best) { best = v; bi = c; }   // centroids ascend: '>=' keeps the last maximum (ggml_vec_argmax_f32)
        }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) if (sv[w] > best || (sv[w] == best && si[w] > bi)) { best = sv[w]; bi = si[w]; }
        xchg_agent_wait(a.cand_val + blockIdx.x, best); xchg_agent_wait(a.cand_idx + blockIdx.x, bi);   // complete before the counter moves
        const unsigned ticket = atomicAdd(a.counter, 1u);
        s_last = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // merge: candidates are few (<= 256), one per thread
    best = -INFINITY; bi = -1;
    if (tid < (int) gridDim.x) { best = ld_agent(a.cand_val + tid); bi = ld_agent(a.cand_idx + tid); }
    __shared__ float mv[256];
    __shared__ int mi[256];
    mv[tid] = best; mi[tid] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            const float v = mv[tid + st]; const int j = mi[tid + st];
            if (v > mv[tid] || (v == mv[tid] && j > mi[tid])) { mv[tid] = v; mi[tid] = j; }
        }
        __syncthreads();
    }
    const int code = mi[0] < 0 ? 0 : mi[0];
    if (tid == 0) { *a.idx_i = code; *a.idx_f = (float) code; *a.counter = 0u; }
    if (a.resid_out) {
        for (int j = tid; j < D; j += 256) {
            const float q = *(const float *) (a.emb + (int64_t) code * a.emb_row_bytes + j * 4);
            a.resid_out[j] = *(const float *) (a.resid + (int64_t) j * a.resid_stride) - q;
        }
    }
}
void k_vq_level(hipStream_t s, const vq_level_args & a) {
    GGML_ASSERT(a.D == 256);
    const int grid = (a.NC + 4 * VQ_CPW - 1) / (4 * VQ_CPW);
    GGML_ASSERT(grid <= 256);
    vq_level_kernel<<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------------------------------------------
// fused sampler (moshi_sample_token, temp > 0; src/moshi/utils/sampling.h:4-64)
//   p = soft_max(logits * (1 / temp))  [ggml: expf(x - max) summed in double, scaled by float(1 / sum)]
//   top-k of p, descending, ties by lower index (the argsort convention of this library)
//   q_j = p_(j) / noise_j  (Exp(1) noise drawn on the host per compute, src/context.h:456-480);  token = index of the LAST maximum of q
// One workgroup; every thread keeps n / 1024 probabilities in registers. The k-th largest value is found by a 4-pass radix select on the
// float bits (p >= 0, so the bit patterns order like the values), the k survivors are rank-sorted in LDS.
// ---------------------------------------------------------------------------------------------------
#if defined(SMP_LOG)   // diagnostic build (tests/microbench/sampler_bench.hip): thread 0 stamps s_memrealtime (10 ns ticks) at the kernel's stages
__device__ unsigned long long g_smp_log[16];
#define SMP_STAMP(i) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_smp_log[i] = t_; } } while (0)
#else
#define SMP_STAMP(i) do {} while (0)
#endif
// NPT: probabilities per thread (every loop over them is unrolled: the 2 048-logit audio heads get their own instantiation with 2 instead of 32)
#define SMP_BIN_MAX 64   // the select stops as soon as the bin that holds the k-th largest value has at most this many members
template <int SMP_THREADS, int SMP_NPT>
__global__ void __launch_bounds__(SMP_THREADS) sample_topk_kernel(sample_args a) {
    constexpr int NW = SMP_THREADS / 64;
    __shared__ float shf[NW];
    __shared__ double shd[NW];
    __shared__ int shi[NW];
    __shared__ __attribute__((aligned(16))) unsigned whist[2][NW][256];   // one private histogram per wave, two sets (the other one is cleared while one is scanned)
    __shared__ unsigned s_bin, s_remaining, s_ncand, s_eq, s_base;
    // a candidate = (bits of p) << 32 | ~index: one unsigned 64-bit compare orders by value descending, then index ascending (p >= 0)
    __shared__ __attribute__((aligned(16))) unsigned long long cand[SAMPLE_MAX_K + SMP_BIN_MAX + 40];
    __shared__ int rank_acc[SAMPLE_MAX_K + SMP_BIN_MAX];
    __shared__ float sort_p[SAMPLE_MAX_K];
    __shared__ int sort_i[SAMPLE_MAX_K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = a.n, k = a.k;
    float p[SMP_NPT];
    SMP_STAMP(0);
    const float my_noise = a.noise[tid < k ? tid : 0];   // (k <= SAMPLE_MAX_K <= threads: rank j's noise sits in thread j from the start, not behind the sort)
    // ---- soft_max
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMP_NPT; j++) {
        const int i = tid + j * SMP_THREADS;
        const float lv = a.logits[i < n ? i : n - 1];   // (no load behind a per-lane branch: the text head's 32 loads per thread were 32 serial round trips)
        p[j] = i < n ? lv * a.scale : -INFINITY;
        mx = fmaxf(mx, p[j]);
    }
    for (int i = tid; i < NW * 256; i += SMP_THREADS) (&whist[0][0][0])[i] = 0u;   // (the first pass's histograms, behind the loads)
    SMP_STAMP(6);
    mx = wave_allmax_f32(mx);
    SMP_STAMP(7);
    if (lane == 0) shf[wave] = mx;
    if (tid == 0) { s_ncand = 0u; }
    __syncthreads();
    SMP_STAMP(8);
    mx = shf[0];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = fmaxf(mx, shf[w]);
    double sum = 0;
#pragma unroll
    for (int j = 0; j < SMP_NPT; j++) {
        if (j * SMP_THREADS >= n) { p[j] = 0.f; continue; }
        const int i = tid + j * SMP_THREADS;
        const float e = i < n ? expf(p[j] - mx) : 0.f;
        p[j] = e;
        sum += (double) e;
    }
    SMP_STAMP(9);
    sum = wave_allsum_f64(sum);
    if (lane == 0) shd[wave] = sum;
    __syncthreads();
    SMP_STAMP(10);
    sum = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) sum += shd[w];
    const float inv = (float) (1.0 / sum);
#pragma unroll
    for (int j = 0; j < SMP_NPT; j++) p[j] *= inv;
    SMP_STAMP(1);
    int total_c = k;   // candidates that reach the rank sort
    {
    // ---- k-th largest by radix select over the bit patterns (p >= 0: the patterns order like the values), most significant byte first. After a pass
    // everything whose leading bytes exceed the selected prefix is in the top k for sure, the members of the selected bin are undecided. The select
    // stops as soon as that bin has <= SMP_BIN_MAX members: they all go to the rank sort with the sure ones, which orders them (value descending,
    // index ascending - the same rule the remaining passes and the tie handling below apply) and keeps ranks < k. Two passes are the rule.
    unsigned prefix = 0u, mask = 0u, remaining = (unsigned) k;
    bool early = false;
    int buf = 0;
    for (int shift = 24; shift >= 0; shift -= 8, buf ^= 1) {
        // one private histogram per wave (same-bin lanes of a wave-instruction are serialised by the LDS unit itself, a few cycles each; a shared
        // histogram costs ~18 us per pass in same-address atomics, a software match loop ~25 us on the text vocabulary's many exponent bins)
#pragma unroll
        for (int j = 0; j < SMP_NPT; j++) {
            if (j * SMP_THREADS >= n) break;
            const int i = tid + j * SMP_THREADS;
            const unsigned key = __float_as_uint(p[j]);
            if (i < n && (key & mask) == prefix) atomicAdd(&whist[buf][wave][(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (wave == 0) {
            // lane l owns bins 255 - 4 l .. 252 - 4 l (descending) and adds the waves' counts itself; inclusive prefix over lanes, the first lane
            // reaching `remaining` resolves its bin
            unsigned h[4] = { 0u, 0u, 0u, 0u }, tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) {
                const u32x4 v = *(const u32x4 *) &whist[buf][w2][252 - 4 * lane];
                h[0] += v.w; h[1] += v.z; h[2] += v.y; h[3] += v.x;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) tot += h[t];
            unsigned inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
            const unsigned long long hit = __ballot(inc >= remaining);
            const int first = __ffsll((long long) hit) - 1;
            if (lane == first) {
                unsigned before = inc - tot;
                int t = 0;
                while (t < 3 && before + h[t] < remaining) { before += h[t]; t++; }
                s_bin = 255u - (unsigned) (4 * lane + t);
                s_remaining = remaining - before;
                s_eq = h[t];
            }
        } else {
            for (int i = tid - 64; i < NW * 256; i += SMP_THREADS - 64) (&whist[buf ^ 1][0][0])[i] = 0u;   // the next pass's histograms
        }
        __syncthreads();
        prefix |= s_bin << shift;
        mask |= 255u << shift;
        remaining = s_remaining;
        if (shift > 0 && s_eq <= (unsigned) SMP_BIN_MAX) { early = true; break; }
    }
    const unsigned T = prefix, need_eq = remaining, have_eq = s_eq;   // (all four passes:) key of the k-th largest; how many of its ties belong to the top k
    SMP_STAMP(2);
    if (early) total_c = k - (int) need_eq + (int) have_eq;          // the sure ones + the whole undecided bin
    // ---- collect: everything above the prefix, then (early: the whole bin; else) the `need_eq` lowest-indexed ties
#pragma unroll
    for (int j = 0; j < SMP_NPT; j++) {
        if (j * SMP_THREADS >= n) break;
        const int i = tid + j * SMP_THREADS;
        const unsigned key = __float_as_uint(p[j]) & mask;
        const bool take = i < n && (key > T || (key == T && (early || have_eq == need_eq)));
        const unsigned long long tb = __ballot(take);   // one LDS atomic per wave-instruction, not per survivor
        if (tb) {
            unsigned base = 0;
            if (lane == __ffsll((long long) tb) - 1) base = atomicAdd(&s_ncand, (unsigned) __popcll(tb));
            base = (unsigned) __shfl((int) base, __ffsll((long long) tb) - 1, 64);
            const unsigned c = base + (unsigned) __popcll(tb & ((1ull << lane) - 1ull));
            if (take && c < SAMPLE_MAX_K + SMP_BIN_MAX) cand[c] = ((unsigned long long) __float_as_uint(p[j]) << 32) | (unsigned long long) (~(unsigned) i);
        }
    }
    if (!early && have_eq != need_eq) {   // more ties than places (rare: exact float ties at the cut): take them in index order, j-major / thread-minor
        if (tid == 0) s_base = 0u;
        __syncthreads();
        for (int j = 0; j < SMP_NPT && j * SMP_THREADS < n; j++) {
            const int i = tid + j * SMP_THREADS;
            const bool f = i < n && __float_as_uint(p[j]) == T;
            const unsigned long long b = __ballot(f);
            if (lane == 0) shi[wave] = __popcll(b);
            __syncthreads();
            unsigned off = s_base;
            for (int w = 0; w < wave; w++) off += (unsigned) shi[w];
            off += (unsigned) __popcll(b & ((1ull << lane) - 1ull));
            if (f && off < need_eq) { const unsigned c = atomicAdd(&s_ncand, 1u); if (c < SAMPLE_MAX_K) cand[c] = ((unsigned long long) __float_as_uint(p[j]) << 32) | (unsigned long long) (~(unsigned) i); }
            __syncthreads();
            if (tid == 0) { unsigned tot = 0; for (int w = 0; w < NW; w++) tot += (unsigned) shi[w]; s_base += tot; }
            __syncthreads();
        }
    }
    if (tid < total_c) rank_acc[tid] = 0;
    if (tid < 40) cand[total_c + tid] = 0ull;   // sentinels (below every candidate) up to the end of the last quarter the sort below reads, eight at a time
    __syncthreads();
    SMP_STAMP(3);
    // ---- rank sort of the candidates: value descending, index ascending; four threads per candidate, each counting a quarter of the list
    // (a wave = 64 candidates x ONE quarter of the list: every lane reads the same eight keys at a time - LDS broadcasts, no bank conflicts; the four
    // quarters' counts of a candidate meet in rank_acc)
    const int quarter = ((total_c + 31) >> 5) << 3;   // multiple of 8, 4 * quarter in [total_c, total_c + 32): the overshoot reads sentinels
    for (int c0 = 0; c0 < total_c; c0 += SMP_THREADS / 4) {
        const int part = wave & 3, c = c0 + (wave >> 2) * 64 + lane;
        if (c0 + (wave >> 2) * 64 >= total_c) break;   // (wave-uniform)
        const bool live = c < total_c;
        const unsigned long long kc = live ? cand[c] : ~0ull;
        int rank = 0;
        for (int d0 = part * quarter; d0 < (part + 1) * quarter; d0 += 8) {
            unsigned long long kd[8];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { const u32x4 t = *(const u32x4 *) (cand + d0 + u); kd[u] = ((unsigned long long) t.y << 32) | t.x; kd[u + 1] = ((unsigned long long) t.w << 32) | t.z; }
#pragma unroll
            for (int u = 0; u < 8; u++) rank += kd[u] > kc ? 1 : 0;
        }
        if (live) atomicAdd(&rank_acc[c], rank);
    }
    __syncthreads();
    for (int c = tid; c < total_c; c += SMP_THREADS) {
        const int rank = rank_acc[c];
        const unsigned long long kc = cand[c];
        if (rank < k) { sort_p[rank] = __uint_as_float((unsigned) (kc >> 32)); sort_i[rank] = (int) ~(unsigned) kc; }
    }
    __syncthreads();
    SMP_STAMP(4);
    }
    // ---- q = p / noise, LAST maximum (ggml_vec_argmax_f32): q >= 0, so (bits of q) << 32 | j orders by value, then by the later position - one max
    // over the k <= 256 threads that hold a rank
    __shared__ unsigned long long s_best[4];
    if (wave < 4) {
        unsigned long long key = tid < k ? ((unsigned long long) __float_as_uint(sort_p[tid] / my_noise) << 32) | (unsigned long long) tid : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ok = ((unsigned long long) (unsigned) __shfl_xor((int) (key >> 32), o, 64) << 32) | (unsigned) __shfl_xor((int) key, o, 64);
            key = ok > key ? ok : key;
        }
        if (lane == 0) s_best[wave] = key;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long key = s_best[0];
        for (int w = 1; w < 4; w++) key = s_best[w] > key ? s_best[w] : key;
        const int tok = sort_i[(int) (unsigned) key];
        *a.out = tok;
        if (a.out2) *a.out2 = tok;
    }
    SMP_STAMP(5);
}
void k_sample_topk(hipStream_t s, const sample_args & a) {
    GGML_ASSERT(a.n >= 1 && a.n <= SAMPLE_MAX_N && a.k >= 1 && a.k <= SAMPLE_MAX_K && a.k <= a.n);
    if (a.n <= 2 * 1024) sample_topk_kernel<1024, 2><<<1, 1024, 0, s>>>(a);
    else sample_topk_kernel<1024, 32><<<1, 1024, 0, s>>>(a);   // (256 threads for the 2 048-logit audio heads measured 13 us slower per call: fewer lanes for the rank sort)
}

__global__ void gather_scalars_kernel(gather_args a) {
    const int i = threadIdx.x;
    if (i >= a.n) return;
    const float v = *a.src[i];
    if (a.dst_type == GGML_TYPE_I32) ((int32_t *) a.dst)[i] = (int32_t) v;
    else ((float *) a.dst)[i] = v;
}
void k_gather_scalars(hipStream_t s, const gather_args & a) {
    GGML_ASSERT(a.n <= GATHER_MAX && (a.dst_type == GGML_TYPE_I32 || a.dst_type == GGML_TYPE_F32));
    gather_scalars_kernel<<<1, 64, 0, s>>>(a);
}
