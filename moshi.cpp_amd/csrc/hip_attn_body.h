// hip_attn_body.h — the body of the streaming self-attention kernels (attn_decode_body) as a device function shared by three callers: attn_decode_kernel (one
// launch per attention block) and inproj_attn_kernel (the Temporal attention as the tail of its in_proj launch), both in hip_kernels_fused.hip, and - round 5 -
// the Q8_0 Depth step program of hip_chain.hip, whose head-owner workgroups run a long-ring attention as a phase of the persistent launch. Included by both
// translation units; MV_LOG stamps only exist in hip_kernels_fused.hip (ATTN_BODY_LOG).
#pragma once
#include "hip_device.h"

// waves per workgroup (NWA): 4 for D = 128 (Temporal / Depth: A/B at fills 0 / 100 / 600 / 2 800, 8 waves cost +50 ... +770 us per frame), 8 for the codec
// transformers' 250-slot rings of D = 64 (8 x NPRE x 8 = 256 slots, the whole ring, are in flight from kernel entry: Mimi -30 us each way)
#define ATTN_NW_BASE 4
#define ATTN_NW_WIDE 8
#ifndef ATTN_SPLIT_NW_DEFAULT
#define ATTN_SPLIT_NW_DEFAULT 4
#endif
#define ATTN_MAX_T 4
#define ATTN_NPRE 4      // ring-slot passes whose K and V rows are requested at kernel entry (4 passes x 4 waves x SPW slots)

// Latency structure: every global load that does not depend on a computed value - the new q/k/v rows, the RoPE table, the mask,
// and the K and V ring rows of the first ATTN_NPRE passes - is requested at kernel entry, ahead of the first wait, so a short
// context costs about one memory round trip; longer contexts stream the remaining rows in the pass loops. Ring rows are valid
// memory for every slot < C, so the speculative rows are simply discarded where the mask says -inf.
//
// SPLIT (long rings, T = 1): workgroup (h, s) owns ring slots [s * ATTN_SPLIT_SLOTS, +ATTN_SPLIT_SLOTS). Up to ATTN_SINGLE_MAX live
// slots the head's first workgroup does everything alone (P = 1, exactly the single-workgroup path; the others leave after a
// mask scan). Beyond that the first P = ceil(n_end / ATTN_SPLIT_SLOTS) workgroups take part: scores of the own slots -> global;
// arrive + bounded wait on the head's counter; every workgroup then redoes the cheap full soft_max from the published scores, so
// the probabilities are bit-identical to the single-workgroup path; P x V partials -> global; the last workgroup to arrive adds
// them in slot order. The wait cannot deadlock: workgroups are dispatched in blockIdx order and a head's workgroups are
// contiguous, so the lowest unfinished head always has all of its workgroups resident (the spin is bounded regardless).
// (round 3) hand-offs are data-tagged: a score / a half of a partial-output double travels as one 8-byte {tag, value} granule written by one agent-scope
// store and polled with agent-scope loads (cdna_hip_programming.md Guideline 16, form R2) - no returning exchanges, no arrival counter, no second hop
// to learn that the counter has moved. tag = the head's launch sequence number + 1, bumped by the workgroup that merges the head.
struct attn_split_ws { unsigned long long * gscores; unsigned long long * gpart; unsigned * seq; int S; unsigned * err; int slots; int single_max; int big_min; };

#if defined(MV_LOG) && defined(ATTN_BODY_LOG)
#define AT_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { if ((i) == 0) at_log_id = atomicAdd(&g_mv_launch, 1u) & 8191u; unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mv_log[at_log_id][i] = t_; \
    if ((i) == 0 || (i) == 7) { unsigned long long r_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory"); g_mv_log[at_log_id][16 + ((i) == 7)] = r_; } \
    if ((i) == 0) { g_mv_log[at_log_id][18] = (unsigned long long) a_in.D | ((unsigned long long) a_in.C << 32); g_mv_log[at_log_id][19] = 0xA7ull | ((unsigned long long) gridDim.x << 32); } } } while (0)
#else
#define AT_STAMP(i) do {} while (0)
#endif
#ifndef ATTN_V_EARLY
#define ATTN_V_EARLY 0   // A/B (profiles/r03_ab_attn_v_request_point.txt): the later V rows requested with the later K rows (1: fill 2 800 290 frames/s - they compete with the K rows the scores wait for) or behind the scores (0: 319)
#endif
// The body of the attention kernels, shared by attn_decode_kernel (one launch per attention block) and inproj_attn_kernel (the Temporal layer's attention
// as the tail of its in_proj launch, below). h / s_idx: the head and the part of the head this workgroup takes (S parts per head); group_y: blockIdx.y of
// the batched-prefill / row-split launches. Returns when this workgroup has no (more) part in the attention.
// MODE: AT_PLAIN; AT_GQKV - the new token's q / k / v rows do not come from memory but from 8-byte {gtag, value} granules (Guideline 16 R2) at
// gq.in[gq.qoff / koff / voff + the element's offset from a.q / a.k / a.v], published by the workgroups of the SAME launch; every thread polls the
// granules of its own elements.
#define AT_PLAIN 0
#define AT_GQKV  2
#define AT_GOUT  4   // the output row also leaves as {tag, value} granules at go.out[h * D + j] (T = 1, unsplit): the next phase of a persistent launch polls it
struct attn_gqkv { const unsigned long long * in; int64_t qoff, koff, voff; unsigned * err; };
struct attn_gout { unsigned long long * out; unsigned tag; int64_t ts;      // granule (t, h, j) at out[t * ts + h * D + j]
#if defined(CH_LOG)
                   unsigned long long * log;                                  // diagnostic build: stage stamps of attn_ring256_body (10 ns ticks) at log[11 ..]
#endif
};
#if defined(CH_LOG)
#define R256_STAMP(i) do { if (GOUT && go.log && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); go.log[i] = t_; } } while (0)
#else
#define R256_STAMP(i) do {} while (0)
#endif
template <bool SPLIT, int NWA, int MODE>
__device__ __forceinline__ void attn_decode_body(const attn_args & a_in, const attn_split_ws & w, char * smem, const int h, const int s_idx, const int group_y,
                                                 const unsigned gtag = 0u, const attn_gqkv gq = attn_gqkv(), const attn_gout go = attn_gout()) {
    constexpr bool GQKV = (MODE & AT_GQKV) != 0, GOUT = (MODE & AT_GOUT) != 0;
    constexpr int ATTN_NW = NWA, ATTN_THREADS = NWA * 64;
#if defined(MV_LOG)
    unsigned at_log_id = 0;
#endif
    AT_STAMP(0);
    attn_args a = a_in;
    if (!SPLIT && a.n_groups > 1) {   // rows 4 g .. 4 g + 3 of a longer block
        const int t0 = 4 * group_y;
        a.T = a.T - t0 < 4 ? a.T - t0 : 4;
        a.q += (int64_t) t0 * a.q_ts; a.k += (int64_t) t0 * a.k_ts; a.v += (int64_t) t0 * a.v_ts;
        if (a.rot) a.rot += (int64_t) t0 * a.D;
        a.mask += (int64_t) t0 * a.C;
        a.index += t0;
        a.out += (int64_t) t0 * a.out_ts;
    }
    // passes whose ring rows are requested before the first wait (the rest streams in batches of the same size in the pass loops)
    // NB: rows per later batch. A split workgroup owns at most 2 x w.slots = 256 slots = 16 passes: after the NPRE it asked for at entry, ALL the others go out
    // in one batch (K before the scores; V right behind the scores, landing during the head-wide hand-off) - one memory round trip each instead of three.
    constexpr int NB = SPLIT ? (NWA == 8 ? 8 : 12) : ATTN_NPRE;   // (8 waves: 32 slots per pass, 2 x 192 slots = 12 passes)
    constexpr int NPRE = ATTN_NPRE;   // A/B at bench level: 4 beats 2 and 8 for the split kernel at empty, 600-slot and full context (registers vs round trips)
    const int D = SPLIT ? 128 : a.D, C = a.C, T = SPLIT ? 1 : a.T;   // (attn_use_split: one query row, 128-wide heads - known to the compiler)
    const int S = SPLIT ? w.S : 1;
    // the range a workgroup owns adapts to the live length: w.slots (small) up to w.big_min live slots, twice that beyond - short
    // ranges cut the per-workgroup round trips at a few hundred slots, long ones the number of participants at a few thousand.
    // The grid is sized for the small range; with the big one the upper half of a head's workgroups simply leaves after the scan.
    int SLOTS = SPLIT ? w.slots : ATTN_SPLIT_SLOTS;
    int c_base = s_idx * SLOTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float * sc   = (float *) smem;          // [C] scores -> exponentials of the current query row
    float * qf   = sc + C;                  // [T][D] bf16-rounded rotated q
    float * knew = qf + T * D;              // [T][D] bf16-rounded new k rows
    float * vnew = knew + T * D;            // [T][D] bf16-rounded new v rows
    double * red = (double *) (smem + (((size_t) (C + 3 * T * D) * 4 + 15) & ~(size_t) 15));   // [4 waves][SPW slot groups][D] partial outputs
    float * msk = (float *) (red + ATTN_NW * (64 / (D / 8)) * D);   // [T][C] the mask rows, staged by the live-range scan (a global load per score pass otherwise)
    __shared__ float sh_f[ATTN_NW];
    __shared__ double sh_d[ATTN_NW];
    __shared__ int sh_i[ATTN_NW];
    __shared__ int sh_slot[ATTN_MAX_T];

    char * kc = a.kcache + (int64_t) h * a.k_nb2, * vc = a.vcache + (int64_t) h * a.v_nb2;
    const unsigned tag = SPLIT ? w.seq[h] + 1u : 0u;   // this launch's hand-off tag for the head (never 0; written back by the head's merger at the very end)
    const int LPS = D / 8;              // lanes per slot (16 for D=128, 8 for D=64): 8 dims (16 B) per lane
    const int SPW = 64 / LPS;           // slots per wave-instruction
    const int sub = lane / LPS, dl = (lane % LPS) * 8;
    const int half = D / 2;

    // last un-masked slot at or beyond `from`, over all query rows (the live range ends there: everything later is -inf and
    // contributes exactly 0). Rows are scanned four slots per load when they are 16-byte aligned.
    auto scan_last_live = [&](int from) {
        int last = -1;
        if ((C & 3) == 0 && (from & 3) == 0 && (((uintptr_t) a.mask) & 15) == 0) {
            const int row4 = C / 4, from4 = from / 4, per_row = row4 - from4, n4 = T * per_row;
            for (int e0 = tid; e0 < n4; e0 += 3 * ATTN_THREADS) {
                float4 m4[3];
                int cs[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int e = e0 + u * ATTN_THREADS, ee = e < n4 ? e : n4 - 1;
                    const int t = ee / per_row, i4 = from4 + (ee - t * per_row);
                    cs[u] = i4 * 4;
                    m4[u] = ((const float4 *) a.mask)[t * row4 + i4];
                    if (e < n4) ((float4 *) msk)[t * row4 + i4] = m4[u];
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    if (e0 + u * ATTN_THREADS < n4) {
                        const int hi = m4[u].w > -INFINITY ? 3 : m4[u].z > -INFINITY ? 2 : m4[u].y > -INFINITY ? 1 : m4[u].x > -INFINITY ? 0 : -1;
                        if (hi >= 0) last = max(last, cs[u] + hi);
                    }
                }
            }
        } else {
            const int per_row = C - from;
            for (int e = tid; e < T * per_row; e += ATTN_THREADS) {
                const int t = e / per_row, c = from + (e - t * per_row);
                const float mv = a.mask[t * C + c];
                msk[t * C + c] = mv;
                if (mv > -INFINITY) last = max(last, c);
            }
        }
        return last;
    };
    auto block_max_i32 = [&](int v) {
        v = wave_allmax_i32(v);
        if (lane == 0) sh_i[wave] = v;
        __syncthreads();
        int r_ = sh_i[0];
#pragma unroll
        for (int w_ = 1; w_ < ATTN_NW; w_++) r_ = max(r_, sh_i[w_]);
        return r_;
    };

    // A split workgroup other than the head's first only has work when something at or beyond its first slot is live; that
    // same scan yields n_end. It finds out BEFORE touching the ring (at short context 11 of 12 workgroups leave here).
    int n_end = 0;
    if (MODE != AT_PLAIN && !SPLIT && s_idx > 0) return;   // (a short ring: the head's first workgroup does everything, the other parts have no attention work)
    if (SPLIT && s_idx > 0) {
        n_end = block_max_i32(scan_last_live(c_base)) + 1;      // exact whenever it exceeds c_base, which is all that matters below
        if (n_end > w.big_min) { SLOTS *= 2; c_base = s_idx * SLOTS; }
        if (n_end <= w.single_max || n_end <= c_base) return;
        __syncthreads();
    }

    // ---- entry loads -----------------------------------------------------------------------------------------------
    int slot_t[ATTN_MAX_T];
#pragma unroll
    for (int t = 0; t < ATTN_MAX_T; t++) slot_t[t] = a.index[t < T ? t : 0];
    // new rows: thread e < T*D handles element (t, j); T*D <= 512 -> at most 2 per thread
    float in_q0[2], in_q1[2], in_k0[2], in_k1[2], in_v[2], in_c[2], in_s[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int e = tid + u * ATTN_THREADS;
        const int ee = e < T * D ? e : 0;
        const int t = ee / D, j = ee - t * D;
        const float * q = a.q + (int64_t) t * a.q_ts + (int64_t) h * a.q_hs;
        const float * k = a.k + (int64_t) t * a.k_ts + (int64_t) h * a.k_hs;
        const float * v = a.v + (int64_t) t * a.v_ts + (int64_t) h * a.v_hs;
        if (!GQKV) in_v[u] = v[j];
        if (a.rot) {
            const int p = j < half ? j : j - half;
            in_c[u] = a.rot[t * D + p]; in_s[u] = a.rot[t * D + half + p];
            if (!GQKV) { in_q0[u] = q[2 * p]; in_q1[u] = q[2 * p + 1]; in_k0[u] = k[2 * p]; in_k1[u] = k[2 * p + 1]; }
        } else {
            in_c[u] = 1.f; in_s[u] = 0.f;
            if (!GQKV) { in_q0[u] = q[j]; in_q1[u] = 0.f; in_k0[u] = k[j]; in_k1[u] = 0.f; }
        }
    }
    // The mask row of the head's first workgroup is requested BEFORE the ring rows (and looked at behind them): vector loads return in order, so a scan
    // issued behind the 2 x NPRE ring-row requests only saw its L2-resident mask once those HBM rows had landed (2.9 us from kernel entry to "loads
    // issued" at 2 800 live slots, profiles/r03_frame_stamps_fill_2800.txt) - and the live length decides everything that follows.
    const bool scan_fast = (!SPLIT || s_idx == 0) && (C & 3) == 0 && (((uintptr_t) a.mask) & 15) == 0 && T * (C / 4) <= 3 * ATTN_THREADS;
    float4 m4pre[3];
    if (scan_fast) {
        const int row4 = C / 4, n4 = T * row4;
#pragma unroll
        for (int u = 0; u < 3; u++) { const int e = tid + u * ATTN_THREADS; m4pre[u] = ((const float4 *) a.mask)[e < n4 ? e : n4 - 1]; }
    }
    // ring rows of the first passes; valid memory for every slot < C, discarded where the mask says -inf
    uint4 kpre[NPRE], vpre[NPRE];
#pragma unroll
    for (int pi = 0; pi < NPRE; pi++) {
        const int c = c_base + wave * SPW + pi * ATTN_NW * SPW + sub;
        const int cc = c < C ? c : C - 1;
        kpre[pi] = *(const uint4 *) (kc + (int64_t) cc * a.k_nb1 + dl * 2);
        if (pi < ATTN_NPRE) vpre[pi] = *(const uint4 *) (vc + (int64_t) cc * a.v_nb1 + dl * 2);   // later V passes: after the scores (registers)
    }
    if (GQKV) {
        // the new rows, from the granules their producers publish (requested behind the ring rows: everything above is in flight while this polls)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = tid + u * ATTN_THREADS;
            in_q0[u] = in_q1[u] = in_k0[u] = in_k1[u] = in_v[u] = 0.f;
            if (e < T * D) {   // (wave-uniform but for the last wave: T * D is a multiple of 64 on this path)
                const int t = e / D, j = e - t * D, p = j < half ? j : j - half;
                const int64_t eq = (int64_t) t * a.q_ts + (int64_t) h * a.q_hs, ek = (int64_t) t * a.k_ts + (int64_t) h * a.k_hs, ev = (int64_t) t * a.v_ts + (int64_t) h * a.v_hs;
                const unsigned long long * gp[5] = { gq.in + gq.qoff + eq + (a.rot ? 2 * p : j), gq.in + gq.qoff + eq + (a.rot ? 2 * p + 1 : j),
                                                     gq.in + gq.koff + ek + (a.rot ? 2 * p : j), gq.in + gq.koff + ek + (a.rot ? 2 * p + 1 : j), gq.in + gq.voff + ev + j };
                unsigned long long g[5];
                int spins = 0;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < 5; i++) g[i] = __hip_atomic_load(gp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 5; i++) ok = ok && (unsigned) (g[i] >> 32) == gtag;
                    if (ok) break;
                    if (++spins >= (1 << 20)) { if (gq.err) *gq.err = 5u; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                in_q0[u] = __uint_as_float((unsigned) g[0]); in_q1[u] = a.rot ? __uint_as_float((unsigned) g[1]) : 0.f;
                in_k0[u] = __uint_as_float((unsigned) g[2]); in_k1[u] = a.rot ? __uint_as_float((unsigned) g[3]) : 0.f;
                in_v[u] = __uint_as_float((unsigned) g[4]);
            }
        }
    }
    int last_live = -1;
    if (scan_fast) {
        const int row4 = C / 4, n4 = T * row4;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int e = tid + u * ATTN_THREADS;
            if (e < n4) {
                ((float4 *) msk)[e] = m4pre[u];
                const int i4 = e % row4;
                const int hi = m4pre[u].w > -INFINITY ? 3 : m4pre[u].z > -INFINITY ? 2 : m4pre[u].y > -INFINITY ? 1 : m4pre[u].x > -INFINITY ? 0 : -1;
                if (hi >= 0) last_live = max(last_live, i4 * 4 + hi);
            }
        }
    } else if (!SPLIT || s_idx == 0) last_live = scan_last_live(0);
    __builtin_amdgcn_sched_barrier(0);
    AT_STAMP(1);

    // ---- 1. RoPE + cache write for all T new rows (the reference's set_rows precede the attention of every row) ----------------
    if (tid < T) sh_slot[tid] = a.index[tid];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int e = tid + u * ATTN_THREADS;
        if (e < T * D) {
            const int t = e / D, j = e - t * D;
            float qo, ko;
            if (a.rot) {
                if (j < half) { qo = in_q0[u] * in_c[u] - in_q1[u] * in_s[u]; ko = in_k0[u] * in_c[u] - in_k1[u] * in_s[u]; }
                else          { qo = in_q0[u] * in_s[u] + in_q1[u] * in_c[u]; ko = in_k0[u] * in_s[u] + in_k1[u] * in_c[u]; }
            } else { qo = in_q0[u]; ko = in_k0[u]; }
            const uint16_t kb = f2bf(ko), vb = f2bf(in_v[u]);
            qf[e] = bf2f(f2bf(qo));
            knew[e] = bf2f(kb);
            vnew[e] = bf2f(vb);
            const int slot = slot_t[t < ATTN_MAX_T ? t : 0];
            if (s_idx == 0 && (SPLIT || !a.row_split || group_y == 0) && slot >= 0 && slot < C) {
                ((uint16_t *) (kc + (int64_t) slot * a.k_nb1))[j] = kb;
                ((uint16_t *) (vc + (int64_t) slot * a.v_nb1))[j] = vb;
            }
        }
    }
    if (!SPLIT && a.write_only) return;
    if (!SPLIT || s_idx == 0) { n_end = block_max_i32(last_live) + 1; if (SPLIT && n_end > w.big_min) SLOTS *= 2; }
    else __syncthreads();
    AT_STAMP(2);
    const int P = SPLIT && n_end > w.single_max ? (n_end + SLOTS - 1) / SLOTS : 1;   // participating workgroups of this head
    const bool multi = SPLIT && P > 1;
    const int c_lo = multi ? c_base : 0, c_hi = multi ? min(n_end, c_base + SLOTS) : n_end;   // slots whose K / V rows this workgroup reads
    const int c_last = c_hi > 0 ? c_hi - 1 : 0;   // last row of the range (address clamp of the batched requests)
    // a prefetched row may be the slot that was just rewritten (last writer wins, like set_rows): take it from LDS instead
    auto pack_row = [&](const float * src) {   // (two 16-byte LDS reads: eight scalar ones, each waited for on its own, were ~0.45 us on the one wave that holds the new slot)
        float4 lo, hi;
        if ((C & 3) == 0) { lo = *(const float4 *) src; hi = *(const float4 *) (src + 4); }   // (knew / vnew start C floats into the buffer: 16-byte aligned when C % 4 == 0)
        else { lo = make_float4(src[0], src[1], src[2], src[3]); hi = make_float4(src[4], src[5], src[6], src[7]); }
        uint4 r;
        r.x = (uint32_t) f2bf(lo.x) | ((uint32_t) f2bf(lo.y) << 16); r.y = (uint32_t) f2bf(lo.z) | ((uint32_t) f2bf(lo.w) << 16);
        r.z = (uint32_t) f2bf(hi.x) | ((uint32_t) f2bf(hi.y) << 16); r.w = (uint32_t) f2bf(hi.z) | ((uint32_t) f2bf(hi.w) << 16);
        return r;   // knew / vnew hold bf16-representable floats: the round trip is exact
    };
    auto fresh_of = [&](int c) { int f = -1; for (int tt = 0; tt < T; tt++) if (slot_t[tt] == c) f = tt; return f; };
#pragma unroll
    for (int pi = 0; pi < NPRE; pi++) {
        const int f = fresh_of(c_base + wave * SPW + pi * ATTN_NW * SPW + sub);
        if (f >= 0) { kpre[pi] = pack_row(knew + f * D + dl); if (pi < ATTN_NPRE) vpre[pi] = pack_row(vnew + f * D + dl); }
    }
    AT_STAMP(8);

    const int t_first = (!SPLIT && a.row_split) ? group_y : 0, t_last = (!SPLIT && a.row_split) ? group_y + 1 : T;   // (row_split: this workgroup's query row only)
    for (int t = t_first; t < t_last; t++) {
        const float * mask = msk + t * C;
        // 2. scores
        float qv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) qv[i] = qf[t * D + dl + i];
        AT_STAMP(9);
        float lmax = -INFINITY;
        auto score_pass = [&](int c0, const uint4 kv) {
            const int c = c0 + sub;
            const float m = c < c_hi ? mask[c] : -INFINITY;
            const bool live = m > -INFINITY;
            const uint32_t kw[4] = { kv.x, kv.y, kv.z, kv.w };
            double acc = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc += (double) (bf2f((uint16_t) (kw[i] & 0xffff)) * qv[2 * i]);
                acc += (double) (bf2f((uint16_t) (kw[i] >> 16)) * qv[2 * i + 1]);
            }
            acc = group_allsum_f64(acc, LPS);
            if (c < c_hi && (lane % LPS) == 0) {
                const float sv = live ? (float) acc * a.scale + m : -INFINITY;
                sc[c] = sv;
                lmax = fmaxf(lmax, sv);
            }
        };
        // further slots: NB rows requested together, then consumed (one memory round trip per batch, not per pass). A split workgroup asks for its one
        // later batch BEFORE it scores the rows it requested at entry.
        uint4 kb[NB];
        auto request_k_batch = [&](int cb) {
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = cb + wave * SPW + pi * ATTN_NW * SPW, c = c0 + sub;
                // Unconditional, back to back (a request under a branch gets an s_waitcnt vmcnt(0) in front of it from the compiler: twelve serial round trips);
                // passes beyond the range ask for its last row again (a cache hit) and are not used.
                kb[pi] = *(const uint4 *) (kc + (int64_t) (c < c_last ? c : c_last) * a.k_nb1 + dl * 2);
                if (SPLIT) __builtin_amdgcn_sched_barrier(0);   // requests in pass order: the first pass's row must not be the last one asked for
            }
        };
        const int cb_first = c_lo + NPRE * ATTN_NW * SPW;
        // (only when the range reaches beyond the rows requested at entry - wave-uniform: at short context the sixteen clamped re-reads of the range's last
        // row would sit in the memory pipe in front of every later wait; MI355X_ATTN_LATE_ALWAYS=1 at build time restores the unconditional form)
#ifndef ATTN_LATE_ALWAYS
        const bool late_rows = SPLIT && cb_first < c_hi;
#else
        const bool late_rows = SPLIT;
#endif
        if (late_rows) request_k_batch(cb_first);
        // ... and its later V rows right behind them (registers are there: two workgroups of four waves per CU): they land during the scores and the hand-off
        uint4 vb0[SPLIT ? NB : 1];
        if (late_rows && ATTN_V_EARLY) {
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = c_lo + (NPRE + pi) * ATTN_NW * SPW + wave * SPW, c = c0 + sub;
                vb0[pi] = *(const uint4 *) (vc + (int64_t) (c < c_last ? c : c_last) * a.v_nb1 + dl * 2);   // (fresh row: patched where it is used)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int pi = 0; pi < NPRE; pi++) {
            const int c0 = c_lo + wave * SPW + pi * ATTN_NW * SPW;
            if (c0 < c_hi) score_pass(c0, kpre[pi]);
        }
        for (int cb = cb_first; cb < c_hi; cb += NB * ATTN_NW * SPW) {
            if (!SPLIT || cb != cb_first) request_k_batch(cb);
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = cb + wave * SPW + pi * ATTN_NW * SPW, f = fresh_of(c0 + sub);
                if (c0 < c_hi) {
                    if (f >= 0) kb[pi] = pack_row(knew + f * D + dl);
                    score_pass(c0, kb[pi]);
                }
            }
        }
        if (late_rows && !ATTN_V_EARLY) {
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = c_lo + (NPRE + pi) * ATTN_NW * SPW + wave * SPW, c = c0 + sub;
                vb0[pi] = *(const uint4 *) (vc + (int64_t) (c < c_last ? c : c_last) * a.v_nb1 + dl * 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        AT_STAMP(10);
#ifndef ATTN_QUICK
#define ATTN_QUICK 1
#endif
        // Short context, one workgroup per head, one query row (the regime every frame of the benchmark window runs in): every live slot is among the
        // rows requested at entry and there are at most 128 scores, so each WAVE can redo the soft-max statistics for itself from the scores in LDS -
        // maximum, exponentials, double sum: the same values in every wave, same inputs, same instructions - and take the probabilities of its own slots
        // from them. Two of the five workgroup barriers of the general path (block maximum, block sum) and its end-of-row barrier disappear; the
        // arithmetic is the general path's except for the association of the exponentials' double sum (lane l adds slots l and l + 64 first).
        if (ATTN_QUICK && !multi && T == 1 && n_end <= 128 && n_end <= NPRE * ATTN_NW * SPW) {
            __syncthreads();                                   // every score is in LDS
            AT_STAMP(3);
            const float s0 = lane < n_end ? sc[lane] : -INFINITY, s1 = lane + 64 < n_end ? sc[lane + 64] : -INFINITY;
            const float qmax = wave_allmax_f32(fmaxf(s0, s1));
            const float e0 = s0 > -INFINITY ? expf(s0 - qmax) : 0.f, e1 = s1 > -INFINITY ? expf(s1 - qmax) : 0.f;
            const double qsum = wave_allsum_f64((double) e0 + (double) e1);
            const float qinv = (float) (1.0 / qsum);
            AT_STAMP(4);
            double q8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q8[i] = 0;
#pragma unroll
            for (int pi = 0; pi < NPRE; pi++) {
                const int c0 = c_lo + wave * SPW + pi * ATTN_NW * SPW, c = c0 + sub;
                if (c0 < c_hi) {
                    const float sv = c < c_hi ? sc[c] : -INFINITY;
                    const float e = sv > -INFINITY ? expf(sv - qmax) : 0.f;
                    const float pq = c < c_hi ? bf2f(f2bf(e * qinv)) : 0.f;
                    const uint32_t vw[4] = { vpre[pi].x, vpre[pi].y, vpre[pi].z, vpre[pi].w };
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        q8[2 * i]     += (double) (bf2f((uint16_t) (vw[i] & 0xffff)) * pq);
                        q8[2 * i + 1] += (double) (bf2f((uint16_t) (vw[i] >> 16)) * pq);
                    }
                }
            }
            AT_STAMP(5);
#pragma unroll
            for (int i = 0; i < 8; i++) red[(wave * SPW + sub) * D + dl + i] = q8[i];
            __syncthreads();
            AT_STAMP(6);
            // fixed order, two stages: every wave adds its own SPW slot groups (into its first row), then the waves are added in index order - two chains
            // of SPW and ATTN_NW terms instead of one of ATTN_NW * SPW dependent LDS reads + double adds by two waves (round 5; attn_ring256_body: 1.3 -> 0.5 us)
            for (int j = lane; j < D; j += 64) {
                double tw = 0;
#pragma unroll
                for (int g = 0; g < 8; g++) if (g < SPW) tw += red[(wave * SPW + g) * D + j];
                red[(wave * SPW) * D + j] = tw;
            }
            __syncthreads();
            for (int j = tid; j < D; j += ATTN_THREADS) {
                double tot = 0;
#pragma unroll
                for (int w = 0; w < ATTN_NW; w++) tot += red[(w * SPW) * D + j];
                a.out[(int64_t) t * a.out_ts + (int64_t) h * D + j] = (float) tot;
                if (GOUT) __hip_atomic_store(go.out + (int64_t) t * go.ts + (int64_t) h * D + j, ((unsigned long long) go.tag << 32) | (unsigned long long) __float_as_uint((float) tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;                                          // (T == 1: the row loop ends here; nothing reuses the LDS)
        }
        lmax = wave_allmax_f32(lmax);
        if (lane == 0) sh_f[wave] = lmax;
        AT_STAMP(11);
        __syncthreads();
        AT_STAMP(3);
        float gmax = sh_f[0];
#pragma unroll
        for (int w_ = 1; w_ < ATTN_NW; w_++) gmax = fmaxf(gmax, sh_f[w_]);
        if (multi) {
            // publish this workgroup's scores (at most one per thread: SLOTS <= 256 = ATTN_THREADS) as tagged granules - plain agent-scope stores, nothing to
            // wait for - then pull everybody else's the same way: every thread polls ITS OWN granules until their tags are this launch's. The head-wide
            // maximum is the maximum over all pulled scores. (Before: returning exchanges + arrival counter + poll + pull = four dependent memory round
            // trips of ~2 us at long context, 14.3 us of a 33 us kernel at 2 800 live slots, profiles/r02_frame_stamps_fill_2800.txt.)
            {
                const int c = c_lo + tid;
                if (c < c_hi) __hip_atomic_store(w.gscores + (int64_t) h * C + c, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(sc[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            {
                constexpr int PULL = 3072 / ATTN_THREADS;                            // C <= 3 072: one batch
                float omax = -INFINITY;
                for (int c0 = tid; c0 < n_end; c0 += PULL * ATTN_THREADS) {
                    unsigned long long g[PULL];
                    int spins = 0;
                    for (;;) {
#pragma unroll
                        for (int u = 0; u < PULL; u++) {
                            const int c = c0 + u * ATTN_THREADS;
                            g[u] = __hip_atomic_load(w.gscores + (int64_t) h * C + (c < n_end ? c : n_end - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        bool ok = true;
#pragma unroll
                        for (int u = 0; u < PULL; u++) {
                            const int c = c0 + u * ATTN_THREADS;
                            if (c < n_end && (c < c_lo || c >= c_hi)) ok = ok && (unsigned) (g[u] >> 32) == tag;   // (own slots come from LDS)
                        }
                        if (ok) break;
                        if (++spins >= (1 << 20)) { if (w.err) *w.err = 1u; break; }   // host-visible: the backend aborts at the next read-back
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int u = 0; u < PULL; u++) {
                        const int c = c0 + u * ATTN_THREADS;
                        if (c < n_end && (c < c_lo || c >= c_hi)) { const float v = __uint_as_float((unsigned) g[u]); sc[c] = v; omax = fmaxf(omax, v); }
                    }
                }
                omax = wave_allmax_f32(omax);
                __syncthreads();                       // (sh_f was read above)
                if (lane == 0) sh_f[wave] = omax;
                __syncthreads();
#pragma unroll
                for (int w_ = 0; w_ < ATTN_NW; w_++) gmax = fmaxf(gmax, sh_f[w_]);
            }
        }

        // 3. soft_max: exp in float, sum in double, scale by (float)(1/sum), round to BF16 for the V product
        double lsum = 0;
        for (int c = tid; c < n_end; c += ATTN_THREADS) {
            const float sv = sc[c];
            const float e = sv > -INFINITY ? expf(sv - gmax) : 0.f;
            sc[c] = e;
            lsum += (double) e;
        }
        lsum = wave_allsum_f64(lsum);
        if (lane == 0) sh_d[wave] = lsum;
        __syncthreads();
        double lall = sh_d[0];
#pragma unroll
        for (int w_ = 1; w_ < ATTN_NW; w_++) lall += sh_d[w_];
        const float inv = (float) (1.0 / lall);   // p = bf16(e * inv), formed where it is used
        AT_STAMP(4);

        // 4. out[d] = sum_c V[d, c] * p[c]
        double o8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) o8[i] = 0;
        auto pv_pass = [&](int c0, const uint4 vv) {
            const int c = c0 + sub;
            const float p = c < c_hi ? bf2f(f2bf(sc[c] * inv)) : 0.f;   // masked slots: e = 0 -> p = 0 -> contributes exactly 0
            const uint32_t vw[4] = { vv.x, vv.y, vv.z, vv.w };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o8[2 * i]     += (double) (bf2f((uint16_t) (vw[i] & 0xffff)) * p);
                o8[2 * i + 1] += (double) (bf2f((uint16_t) (vw[i] >> 16)) * p);
            }
        };
#pragma unroll
        for (int pi = 0; pi < NPRE; pi++) {
            const int c0 = c_lo + wave * SPW + pi * ATTN_NW * SPW;
            if (c0 < c_hi) pv_pass(c0, vpre[pi]);
        }
        if (late_rows) {
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = c_lo + (NPRE + pi) * ATTN_NW * SPW + wave * SPW, f = fresh_of(c0 + sub);
                if (c0 < c_hi) {
                    if (f >= 0) vb0[pi] = pack_row(vnew + f * D + dl);
                    pv_pass(c0, vb0[pi]);
                }
            }
        }
        for (int cb = c_lo + (SPLIT ? NPRE + NB : NPRE) * ATTN_NW * SPW; cb < c_hi; cb += NB * ATTN_NW * SPW) {
            uint4 vb[NB];
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = cb + wave * SPW + pi * ATTN_NW * SPW, c = c0 + sub;
                vb[pi] = *(const uint4 *) (vc + (int64_t) (c < c_last ? c : c_last) * a.v_nb1 + dl * 2);
            }
#pragma unroll
            for (int pi = 0; pi < NB; pi++) {
                const int c0 = cb + wave * SPW + pi * ATTN_NW * SPW, f = fresh_of(c0 + sub);
                if (c0 < c_hi) {
                    if (f >= 0) vb[pi] = pack_row(vnew + f * D + dl);
                    pv_pass(c0, vb[pi]);
                }
            }
        }
        AT_STAMP(5);
#pragma unroll
        for (int i = 0; i < 8; i++) red[(wave * SPW + sub) * D + dl + i] = o8[i];
        __syncthreads();
        AT_STAMP(6);
        for (int j = tid; j < D; j += ATTN_THREADS) {
            double tot = 0;
#pragma unroll 8
            for (int g = 0; g < ATTN_NW * SPW; g++) tot += red[g * D + j];   // fixed order: wave-major, slot group minor
            if (multi) {
                unsigned long long * gp = w.gpart + (((int64_t) h * S + s_idx) * D + j) * 2;
                const unsigned long long bits = (unsigned long long) __double_as_longlong(tot);
                __hip_atomic_store(gp, ((unsigned long long) tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gp + 1, ((unsigned long long) tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                a.out[(int64_t) t * a.out_ts + (int64_t) h * D + j] = (float) tot;
                if (GOUT) __hip_atomic_store(go.out + (int64_t) t * go.ts + (int64_t) h * D + j, ((unsigned long long) go.tag << 32) | (unsigned long long) __float_as_uint((float) tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (multi && s_idx == 0) {
            // the head's first workgroup adds the P partial outputs in slot order (its own included: read back like the others), polling each granule pair
            // until it carries this launch's tag; then it bumps the head's sequence number - every workgroup has long read it, and a workgroup publishes its
            // partial only after it has finished reading the scores
            for (int j = tid; j < D; j += ATTN_THREADS) {
                double tot = 0;
                for (int q0 = 0; q0 < P; q0 += 8) {
                    unsigned long long lo[8], hi[8];
                    int spins = 0;
                    for (;;) {
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const unsigned long long * gp = w.gpart + (((int64_t) h * S + (q0 + u < P ? q0 + u : P - 1)) * D + j) * 2;
                            lo[u] = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            hi[u] = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        bool ok = true;
#pragma unroll
                        for (int u = 0; u < 8; u++) ok = ok && (unsigned) (lo[u] >> 32) == tag && (unsigned) (hi[u] >> 32) == tag;
                        if (ok) break;
                        if (++spins >= (1 << 20)) { if (w.err) *w.err = 1u; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) if (q0 + u < P) tot += __longlong_as_double((long long) ((lo[u] & 0xffffffffull) | (hi[u] << 32)));   // slot order
                }
                a.out[(int64_t) t * a.out_ts + (int64_t) h * D + j] = (float) tot;
            }
            __syncthreads();
            if (tid == 0) w.seq[h] = tag;   // (plain store: read by the next launch)
        }
        __syncthreads();
    }
    AT_STAMP(7);
}


// ---- single-token attention of ONE head over a SHORT ring (C <= 64 slots of D = 64) by one workgroup of 8 waves -------------------------------------------
// The tts-shaped Depth transformer's attention (16 heads x 64, a ring as long as its 32-step schedule, lm_default.h:86-90). attn_decode_body is built for
// rings of hundreds of slots - mask scan, prefetch passes, batched streaming - and spends ~7 us on 32 slots; this is the direct form: wave w owns slots
// 8 w .. 8 w + 7, lane = (slot, 8-dim chunk). ggml's CPU sequence for T = 1 (transformer.h:543-576, torch.h:225-237): q / k rotated (interleaved pairs in,
// [re | im] out) and rounded to BF16, K / V written to the ring as BF16; score_c = <K_c, q> as float products summed in double; soft_max(score * scale +
// mask) with the sum in double and p = e * float(1 / sum) rounded to BF16; out_d = sum over the slots IN SLOT ORDER of V_c[d] * p_c in double.
// MODE: AT_GQKV - the new token's q / k / v come from {gtag, value} granules (gq), polled by the 64 threads that stage them; AT_GOUT - the 64 outputs also
// leave as granules (go). smem: >= ATTN_RING64_SMEM bytes. Every thread of the 512 must call it (four workgroup barriers inside).
#define ATTN_RING64_SMEM ((3 * 64 + 64 * 64 + 16 + 16) * 4)
template <int MODE>
__device__ __forceinline__ void attn_ring64_body(const attn_args & a, char * smem, const int h, const unsigned gtag = 0u, const attn_gqkv gq = attn_gqkv(), const attn_gout go = attn_gout()) {
    constexpr bool GQKV = (MODE & AT_GQKV) != 0, GOUT = (MODE & AT_GOUT) != 0;
    constexpr int D = 64, half = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float * qf = (float *) smem, * knew = qf + 64, * vnew = knew + 64, * prod = vnew + 64;   // prod: [64 slots][64 dims]
    float * wmax = prod + 64 * 64; double * wsum = (double *) (wmax + 16);
    const int C = a.C;
    const int sub = lane >> 3, dl = (lane & 7) * 8, c = wave * 8 + sub, cc = c < C ? c : C - 1;
    // ---- entry loads: everything that does not depend on a computed value
    const int slot = a.index[0];
    const float m = a.mask[cc];
    const char * kc = a.kcache + (int64_t) h * a.k_nb2, * vc = a.vcache + (int64_t) h * a.v_nb2;
    const uint4 kq = *(const uint4 *) (kc + (int64_t) cc * a.k_nb1 + dl * 2);
    const uint4 vq = *(const uint4 *) (vc + (int64_t) cc * a.v_nb1 + dl * 2);
    if (wave == 0) {
        const int j = lane, p = j < half ? j : j - half;
        float rc = 1.f, rs = 0.f;
        if (a.rot) { rc = a.rot[p]; rs = a.rot[half + p]; }
        float q0, q1, k0, k1, vv;
        if (GQKV) {
            const int64_t eq = (int64_t) h * a.q_hs, ek = (int64_t) h * a.k_hs, ev = (int64_t) h * a.v_hs;
            const unsigned long long * gp5[5] = { gq.in + gq.qoff + eq + (a.rot ? 2 * p : j), gq.in + gq.qoff + eq + (a.rot ? 2 * p + 1 : j),
                                                  gq.in + gq.koff + ek + (a.rot ? 2 * p : j), gq.in + gq.koff + ek + (a.rot ? 2 * p + 1 : j), gq.in + gq.voff + ev + j };
            unsigned long long g[5];
            int spins = 0;
            for (;;) {
#pragma unroll
                for (int i = 0; i < 5; i++) g[i] = __hip_atomic_load(gp5[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 5; i++) ok = ok && (unsigned) (g[i] >> 32) == gtag;
                if (__all(ok)) break;
                if (++spins >= (1 << 20)) { if (gq.err) *gq.err = 5u; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            q0 = __uint_as_float((unsigned) g[0]); q1 = a.rot ? __uint_as_float((unsigned) g[1]) : 0.f;
            k0 = __uint_as_float((unsigned) g[2]); k1 = a.rot ? __uint_as_float((unsigned) g[3]) : 0.f;
            vv = __uint_as_float((unsigned) g[4]);
        } else {
            const float * q = a.q + (int64_t) h * a.q_hs, * k = a.k + (int64_t) h * a.k_hs, * v = a.v + (int64_t) h * a.v_hs;
            if (a.rot) { q0 = q[2 * p]; q1 = q[2 * p + 1]; k0 = k[2 * p]; k1 = k[2 * p + 1]; }
            else { q0 = q[j]; q1 = 0.f; k0 = k[j]; k1 = 0.f; }
            vv = v[j];
        }
        float qo, ko;
        if (a.rot) {
            if (j < half) { qo = q0 * rc - q1 * rs; ko = k0 * rc - k1 * rs; }
            else          { qo = q0 * rs + q1 * rc; ko = k0 * rs + k1 * rc; }
        } else { qo = q0; ko = k0; }
        const uint16_t kb = f2bf(ko), vb = f2bf(vv);
        qf[j] = bf2f(f2bf(qo)); knew[j] = bf2f(kb); vnew[j] = bf2f(vb);
        if (slot >= 0 && slot < C) {
            ((uint16_t *) (a.kcache + (int64_t) h * a.k_nb2 + (int64_t) slot * a.k_nb1))[j] = kb;
            ((uint16_t *) (a.vcache + (int64_t) h * a.v_nb2 + (int64_t) slot * a.v_nb1))[j] = vb;
        }
    }
    __syncthreads();
    // ---- scores
    const bool live = c < C && m > -INFINITY, fresh = slot == c;
    float kv[8], vv8[8];
    {
        const uint32_t kw[4] = { kq.x, kq.y, kq.z, kq.w }, vw[4] = { vq.x, vq.y, vq.z, vq.w };
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float kr8 = bf2f((uint16_t) ((i & 1) ? (kw[i >> 1] >> 16) : (kw[i >> 1] & 0xffff))), vr8 = bf2f((uint16_t) ((i & 1) ? (vw[i >> 1] >> 16) : (vw[i >> 1] & 0xffff)));
            const float kn = knew[dl + i], vn = vnew[dl + i];   // (read by every lane, selected afterwards: no LDS read behind a per-lane branch)
            kv[i] = fresh ? kn : kr8;
            vv8[i] = fresh ? vn : vr8;
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += (double) (kv[i] * qf[dl + i]);
    acc = group_allsum_f64(live ? acc : 0.0, 8);
    const float sv = live ? (float) acc * a.scale + m : -INFINITY;
    {
        float wm = sv;
        wm = fmaxf(wm, dpp_f32<DPP_ROW_MIRROR>(wm));   // (uniform inside a slot's 8 lanes: the in-group steps would return their input)
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 48));
        if (lane == 0) wmax[wave] = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    }
    __syncthreads();
    float gmax = wmax[0];
#pragma unroll
    for (int w = 1; w < 8; w++) gmax = fmaxf(gmax, wmax[w]);
    const float e = sv > -INFINITY ? expf(sv - gmax) : 0.f;
    {
        double ws = (lane & 7) == 0 ? (double) e : 0.0;   // one representative per slot
        ws += dpp_f64<DPP_QUAD_XOR1>(ws); ws += dpp_f64<DPP_QUAD_XOR2>(ws); ws += dpp_f64<DPP_HALF_MIRROR>(ws); ws += dpp_f64<DPP_ROW_MIRROR>(ws);
        const int lo = __double2loint(ws), hi = __double2hiint(ws);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0)), r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)), r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        if (lane == 0) wsum[wave] = (r0 + r1) + (r2 + r3);
    }
    __syncthreads();
    double lsum = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) lsum += wsum[w];
    const float inv = (float) (1.0 / lsum);
    const float pr = bf2f(f2bf(e * inv));
    {
        float pf[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pf[i] = pr != 0.f ? vv8[i] * pr : 0.f;
        float * dst = prod + c * 64 + dl;
        *(float4 *) dst = make_float4(pf[0], pf[1], pf[2], pf[3]);
        *(float4 *) (dst + 4) = make_float4(pf[4], pf[5], pf[6], pf[7]);
    }
    __syncthreads();
    if (tid < D) {
        double tot = 0;
        for (int c2 = 0; c2 < C; c2++) tot += (double) prod[c2 * 64 + tid];
        a.out[(int64_t) h * D + tid] = (float) tot;
        if (GOUT) __hip_atomic_store(go.out + (int64_t) h * D + tid, ((unsigned long long) go.tag << 32) | (unsigned long long) __float_as_uint((float) tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// ---- one query row of T <= 2 new tokens of ONE head over a ring of C <= 256 slots of D = 64, by one workgroup of 8 waves -----------------------------------
// The codec transformers' shape (8 heads x 64, ring of 250, two 25 Hz frames per call; compression.h:173, 293). attn_decode_body spends ~8.7 us here; this is
// the direct form of attn_ring64_body with four passes: wave w owns slots 32 w .. 32 w + 31, pass i its slots 8 i .. 8 i + 7, lane = (slot, 8-dim chunk); all
// ring rows are requested at entry. Workgroup (h, t) computes row t: it rotates and rounds ALL T new rows (row 1 attends to row 0's K / V, which is not in the
// ring yet - or is being written right now by workgroup (h, 0), the only one that writes), taking new rows from LDS wherever a slot is one of the new tokens'.
// Arithmetic as attn_ring64_body; the P x V sums run per (wave, slot group) over the passes in double, then over a wave's 8 groups and over the 8 waves in index order in double.
#define ATTN_RING256_SMEM ((64 + 2 * 64 + 2 * 64 + 16) * 4 + 8 * 8 + 64 * 64 * 8)
template <int MODE>
__device__ __forceinline__ void attn_ring256_body(const attn_args & a, char * smem, const int h, const int t, const unsigned gtag = 0u, const attn_gqkv gq = attn_gqkv(), const attn_gout go = attn_gout()) {
    constexpr bool GQKV = (MODE & AT_GQKV) != 0, GOUT = (MODE & AT_GOUT) != 0;
    constexpr int D = 64, half = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float * qf = (float *) smem, * knew = qf + 64, * vnew = knew + 128, * wmax = vnew + 128;
    double * wsum = (double *) (wmax + 16), * red = wsum + 8;                                  // red: [64 groups][64 dims]
    const int C = a.C, T = a.T;
    const int sub = lane >> 3, dl = (lane & 7) * 8;
    // ---- entry loads
    const int slot0 = a.index[0], slot1 = T > 1 ? a.index[1] : -1;
    const char * kc = a.kcache + (int64_t) h * a.k_nb2, * vc = a.vcache + (int64_t) h * a.v_nb2;
    uint4 kq[4], vq[4]; float m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = wave * 32 + i * 8 + sub, cc = c < C ? c : C - 1;
        m[i] = a.mask[(int64_t) t * C + cc];
        kq[i] = *(const uint4 *) (kc + (int64_t) cc * a.k_nb1 + dl * 2);
        vq[i] = *(const uint4 *) (vc + (int64_t) cc * a.v_nb1 + dl * 2);
    }
    if (wave < T) {   // wave tt stages new row tt (its q only where it is THE query row)
        const int tt = wave, j = lane, p = j < half ? j : j - half;
        float rc = 1.f, rs = 0.f;
        if (a.rot) { rc = a.rot[tt * D + p]; rs = a.rot[tt * D + half + p]; }
        float q0, q1, k0, k1, vv;
        if (GQKV) {
            const int64_t eq = (int64_t) tt * a.q_ts + (int64_t) h * a.q_hs, ek = (int64_t) tt * a.k_ts + (int64_t) h * a.k_hs, ev = (int64_t) tt * a.v_ts + (int64_t) h * a.v_hs;
            const unsigned long long * gp5[5] = { gq.in + gq.qoff + eq + (a.rot ? 2 * p : j), gq.in + gq.qoff + eq + (a.rot ? 2 * p + 1 : j),
                                                  gq.in + gq.koff + ek + (a.rot ? 2 * p : j), gq.in + gq.koff + ek + (a.rot ? 2 * p + 1 : j), gq.in + gq.voff + ev + j };
            unsigned long long g[5];
            int spins = 0;
            for (;;) {
#pragma unroll
                for (int i = 0; i < 5; i++) g[i] = __hip_atomic_load(gp5[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 5; i++) ok = ok && (unsigned) (g[i] >> 32) == gtag;
                if (__all(ok)) break;
                if (++spins >= (1 << 20)) { if (gq.err) *gq.err = 5u; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            q0 = __uint_as_float((unsigned) g[0]); q1 = a.rot ? __uint_as_float((unsigned) g[1]) : 0.f;
            k0 = __uint_as_float((unsigned) g[2]); k1 = a.rot ? __uint_as_float((unsigned) g[3]) : 0.f;
            vv = __uint_as_float((unsigned) g[4]);
        } else {
            const float * q = a.q + (int64_t) tt * a.q_ts + (int64_t) h * a.q_hs, * k = a.k + (int64_t) tt * a.k_ts + (int64_t) h * a.k_hs, * v = a.v + (int64_t) tt * a.v_ts + (int64_t) h * a.v_hs;
            if (a.rot) { q0 = q[2 * p]; q1 = q[2 * p + 1]; k0 = k[2 * p]; k1 = k[2 * p + 1]; }
            else { q0 = q[j]; q1 = 0.f; k0 = k[j]; k1 = 0.f; }
            vv = v[j];
        }
        float qo, ko;
        if (a.rot) {
            if (j < half) { qo = q0 * rc - q1 * rs; ko = k0 * rc - k1 * rs; }
            else          { qo = q0 * rs + q1 * rc; ko = k0 * rs + k1 * rc; }
        } else { qo = q0; ko = k0; }
        const uint16_t kb = f2bf(ko), vb = f2bf(vv);
        if (tt == t) qf[j] = bf2f(f2bf(qo));
        knew[tt * 64 + j] = bf2f(kb); vnew[tt * 64 + j] = bf2f(vb);
        const int slot = tt == 0 ? slot0 : slot1;
        if (t == 0 && slot >= 0 && slot < C) {
            ((uint16_t *) (a.kcache + (int64_t) h * a.k_nb2 + (int64_t) slot * a.k_nb1))[j] = kb;
            ((uint16_t *) (a.vcache + (int64_t) h * a.v_nb2 + (int64_t) slot * a.v_nb1))[j] = vb;
        }
        R256_STAMP(11);
    }
    __syncthreads();
    R256_STAMP(12);
    // ---- scores of the four passes
    float qv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) qv[i] = qf[dl + i];
    float sv[4], vv8[4][8];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = wave * 32 + i * 8 + sub;
        live[i] = c < C && m[i] > -INFINITY;
        const int f = c == slot0 ? 0 : (T > 1 && c == slot1) ? 1 : -1, fc = f > 0 ? 1 : 0;
        const uint32_t kw[4] = { kq[i].x, kq[i].y, kq[i].z, kq[i].w }, vw[4] = { vq[i].x, vq[i].y, vq[i].z, vq[i].w };
        // (the new rows are read from LDS by every lane and selected afterwards: behind the per-lane `f >= 0` each of the 16 reads was a branch with its
        //  own wait - 64 serial LDS round trips over the four passes)
        float kn8[8], vn8[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const float4 k4 = *(const float4 *) (knew + fc * 64 + dl + e), v4 = *(const float4 *) (vnew + fc * 64 + dl + e);
            kn8[e] = k4.x; kn8[e + 1] = k4.y; kn8[e + 2] = k4.z; kn8[e + 3] = k4.w; vn8[e] = v4.x; vn8[e + 1] = v4.y; vn8[e + 2] = v4.z; vn8[e + 3] = v4.w;
        }
        double acc = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float kr8 = bf2f((uint16_t) ((e & 1) ? (kw[e >> 1] >> 16) : (kw[e >> 1] & 0xffff))), vr8 = bf2f((uint16_t) ((e & 1) ? (vw[e >> 1] >> 16) : (vw[e >> 1] & 0xffff)));
            const float kk = f >= 0 ? kn8[e] : kr8;
            vv8[i][e] = f >= 0 ? vn8[e] : vr8;
            acc += (double) (kk * qv[e]);
        }
        acc = group_allsum_f64(live[i] ? acc : 0.0, 8);
        sv[i] = live[i] ? (float) acc * a.scale + m[i] : -INFINITY;
    }
    R256_STAMP(0);
    {
        float wm = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        wm = fmaxf(wm, dpp_f32<DPP_ROW_MIRROR>(wm));   // (uniform inside a slot's 8 lanes)
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 48));
        if (lane == 0) wmax[wave] = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    }
    __syncthreads();
    R256_STAMP(13);
    float gmax = wmax[0];
#pragma unroll
    for (int w = 1; w < 8; w++) gmax = fmaxf(gmax, wmax[w]);
    float e4[4];
    {
        double ws = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { e4[i] = sv[i] > -INFINITY ? expf(sv[i] - gmax) : 0.f; if ((lane & 7) == 0) ws += (double) e4[i]; }
        ws += dpp_f64<DPP_QUAD_XOR1>(ws); ws += dpp_f64<DPP_QUAD_XOR2>(ws); ws += dpp_f64<DPP_HALF_MIRROR>(ws); ws += dpp_f64<DPP_ROW_MIRROR>(ws);
        const int lo = __double2loint(ws), hi = __double2hiint(ws);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0)), r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)), r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        if (lane == 0) wsum[wave] = (r0 + r1) + (r2 + r3);
    }
    R256_STAMP(3);
    __syncthreads();
    R256_STAMP(4);
    double lsum = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) lsum += wsum[w];
    const float inv = (float) (1.0 / lsum);
    {
        double o8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o8[e] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float pr = bf2f(f2bf(e4[i] * inv));
#pragma unroll
            for (int e = 0; e < 8; e++) o8[e] += (double) (pr != 0.f ? vv8[i][e] * pr : 0.f);
        }
        R256_STAMP(6);
        double * dst = red + (wave * 8 + sub) * 64 + dl;
#pragma unroll
        for (int e = 0; e < 8; e++) dst[e] = o8[e];
    }
    R256_STAMP(7);
    __syncthreads();
    R256_STAMP(14);
    {   // each wave adds its own 8 slot groups (into its first row), then one wave adds the 8 waves: two chains of 8 instead of one of 64 (1.3 -> ~0.6 us)
        double tw = 0;
#pragma unroll
        for (int g = 0; g < 8; g++) tw += red[(wave * 8 + g) * 64 + lane];
        red[(wave * 8) * 64 + lane] = tw;
    }
    __syncthreads();
    if (tid < D) {
        double tot = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) tot += red[(w * 8) * 64 + tid];
        a.out[(int64_t) t * a.out_ts + (int64_t) h * D + tid] = (float) tot;
        if (GOUT) __hip_atomic_store(go.out + (int64_t) t * go.ts + (int64_t) h * D + tid, ((unsigned long long) go.tag << 32) | (unsigned long long) __float_as_uint((float) tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    R256_STAMP(15);
    __syncthreads();
}
