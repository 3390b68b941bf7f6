// hip_chain_nest80.h — part of hip_chain.hip (included there, same translation unit): the Q8_0 Depth transformer of the tts-shaped models (BASELINE.json
// configs[1]: `moshi-tts -q q8_0`; /root/reference/src/moshi/models/lm.h:446-553 with lm_utils.h:157-217 low-rank embeddings, the per-step weight schedule of
// lm_default.h:71-81 and a ring as long as the schedule, lm_default.h:86-90) as ONE persistent launch per frame - the step program of hip_chain_nest.h carried
// over to Q8_0 blocks, with the three things that shape brings:
//   * the attention runs over a ring of up to 32 slots: recomputing it in every workgroup (what the 8-slot ring of moshika affords) measured slower than its own
//     launch (profiles/r04_ab_tts_depth_attention_in_out_proj.txt). Here it is a PHASE: 16 head-owner workgroups (spread over the XCDs) take their head's q / k / v
//     from the in_proj phase's granules, run attn_ring64_body (hip_attn_body.h) - the very code of the stand-alone launch - and publish the head's 64 outputs as granules;
//     out_proj then is an ordinary phase that polls 1024 values. The other 240 workgroups skip the phase and wait in out_proj's poll;
//   * step k's input is depformer_in[schedule[k]] x transformer_out + low_rank_k(emb_k[token]): the products of the DISTINCT depformer_in matrices (9 for 32
//     steps) are hoisted to the head of the launch; the low-rank embedding - one 128-wide table row re-quantised to Q8_0 and sent through a 128 -> 1024 Q8_0
//     projection, lowrank_embed_kernel's arithmetic - is computed by every workgroup for itself behind the token hand-off (139 KB of weights, L2-resident after
//     the first workgroup) and folded into layer 0's in_proj input;
//   * Q8_0 weights: 8 lanes per 272-byte chunk of 256 weights, one 34-byte block per lane (the layout of matvec_chain_kernel's Q8_0 phases), Q8_0 activations,
//     the chunk's eight terms added in block order.
// Arithmetic: bit-identical to one launch per step of the plan (tests/test_chain_engine.py::test_tts_shaped_depth_program_*): same quantisers, same block
// dots and float sequences, same row sums, the attention by the same device function at the same wave count.
#pragma once

struct n80_at { char * kcache; char * vcache; const float * rot; const float * mask; const int32_t * index; float * out; };               // one attention (48 B)
struct n80_st {                                                                                                                          // one step
    float * din_y; const float * res_mem;   // the node din_k + last (its storage); last from memory (step 0: computed by launches in front of the program) or NULL
    const char * lr_table; long long lr_row_bytes, lr_n_rows; const int32_t * lr_index; const char * lr_w; long long lr_w_row_bytes; float * lr_out;
    int32_t * prev_out[2]; int32_t * argmax_out[2];
    int lr_type; int din_set; int emb_chain; int pad; long long pad2;
};
static_assert(sizeof(n80_at) == 48 && sizeof(n80_st) % 16 == 0, "tables are copied to LDS by 16-byte lanes");

struct nest80_params {
    chain_params P;
    const u32x4 * tables;      // device: nest_ph[n_steps * (4 L + 1)] | n80_at[n_steps * L] | n80_st[n_steps] | din weight pointers [16] (8 bytes each)
    int n_steps, n_layers, n_sets;
    u64 * din_buf;             // [n_sets][1024] granules
    const float * din_x;       // transformer_out
    attn_args at;              // the attention's shape (pointers replaced per phase)
    int q_off, k_off, v_off;   // of head 0's q / k / v inside the in_proj vector
    size_t attn_smem;          // LDS bytes attn_decode_body needs at this shape
    int delay[4];              // s_sleep units before the first poll of: in_proj / linear_in / linear_out / head [0], out_proj of non-owner workgroups [1], owners [2], -
};
#define N80_SETS_MAX 16
#define N80_STEPS_MAX 32

template <int KIN, int FF>
__global__ void __launch_bounds__(CH_THREADS) depth_nest80_kernel(nest80_params N) {
    constexpr int G = 256, grid = G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int L = N.n_layers, per_step = 5 * L + 1, mv_per_step = 4 * L + 1;
    const int n_ph = N.n_steps * mv_per_step, n_at = N.n_steps * L;
    static_assert(KIN % 256 == 0 && KIN <= 4096 && FF % 256 == 0 && FF <= 4096 && (FF / 256) * 2 * 4 <= 64 && FF % G == 0, "shapes of the Q8_0 step program");
    constexpr int NBIN = KIN / 256, NBF = FF / 256, FROWS = FF / G;

    xblk80 * xs = (xblk80 *) smem;                                   // [16]
    float * part = (float *) (xs + 16);                              // [1024]
    float * xres = part + 1024;                                      // [16]
    float * emb = xres + 16;                                         // [1024] the low-rank embedding of the step
    int8_t * lrq = (int8_t *) (emb + 1024);                          // [128] its table row, re-quantised
    float * lrd = (float *) (lrq + 128);                             // [4] ... and the blocks' scales (+ pad)
    chain_ctl * ctl = (chain_ctl *) (lrd + 4);
    char * at_smem = (char *) (ctl + 1);
    const nest_ph * t_ph = (const nest_ph *) (at_smem + ((N.attn_smem + 15) & ~(size_t) 15));
    const n80_at * t_at = (const n80_at *) (t_ph + n_ph);
    const n80_st * t_st = (const n80_st *) (t_at + n_at);
    const char * const * t_dw = (const char * const *) (t_st + N.n_steps);   // [N80_SETS_MAX]
    float * din_all = (float *) (t_dw + N80_SETS_MAX);               // [n_sets][1024]

    const chain_params & P = N.P;
    auto nbar = [&]() { lds_barrier(); };
    if (tid == 0) { ctl->failed = 0; ctl->token = 0; }
    {
        const int n16 = (n_ph * (int) sizeof(nest_ph) + n_at * (int) sizeof(n80_at) + N.n_steps * (int) sizeof(n80_st) + N80_SETS_MAX * 8) / 16;
        const GLOBAL_AS u32x4 * src = (const GLOBAL_AS u32x4 *) N.tables;
        for (int i = tid; i < n16; i += CH_THREADS) ((u32x4 *) t_ph)[i] = src[i];
    }
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    __syncthreads();

    auto ld_ph = [&](int q) {
        const u32x4 a = ((const u32x4 *) (t_ph + q))[0], b = ((const u32x4 *) (t_ph + q))[1];
        nest_ph r;
        unsigned w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) w[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };
    auto ld_w = [&](int q) -> const char * {
        const unsigned * s = (const unsigned *) (t_ph + q);
        unsigned w[2] = { (unsigned) __builtin_amdgcn_readfirstlane((int) s[0]), (unsigned) __builtin_amdgcn_readfirstlane((int) s[1]) };
        const char * r;
        __builtin_memcpy(&r, w, 8);
        return r;
    };
    auto ld_at = [&](int q) {
        n80_at r;
        unsigned w[12];
        const unsigned * s = (const unsigned *) (t_at + q);
#pragma unroll
        for (int i = 0; i < 12; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) s[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };
    auto ld_st = [&](int q) {
        n80_st r;
        constexpr int NW = (int) sizeof(n80_st) / 4;
        unsigned w[NW];
        const unsigned * s = (const unsigned *) (t_st + q);
#pragma unroll
        for (int i = 0; i < NW; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) s[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };

    const __amdgpu_buffer_rsrc_t gb = make_rsrc(P.gbuf, 2u * CH_XF_MAX * 8u);
    const __amdgpu_buffer_rsrc_t cb = make_rsrc(P.cand, 2u * 2u * (unsigned) grid * 8u);
    auto give_up = [&]() { if (lane == 0) { lds_store(&ctl->failed, 1u); *gp(P.err) = 2u; } };

    // ---- one phase's weights in registers: one pass = 64 chunks of 272 bytes, 8 lanes per chunk, lane j the 9 dwords from byte 34 j - (j odd ? 2 : 0)
    u32x4 wh, wq; uint32_t we;
    const int j8 = lane & 7, boff = 34 * j8 - ((j8 & 1) ? 2 : 0);
    auto request = [&](const char * w, int nb, int rows, int pairF) {   // rows: of this workgroup (paired: of each half)
        const long long row_bytes = (long long) nb * 272;
        const int nseg = rows * nb, nall = pairF ? 2 * nseg : nseg;
        const long long r0 = (long long) wg * rows;
        const int sb = wave * 8 + (lane >> 3);
        const int sbc = sb < nall ? sb : nall - 1;
        const GLOBAL_AS char * src = gp(w) + (sbc < nseg ? r0 * row_bytes + (long long) sbc * 272 : (r0 + pairF) * row_bytes + (long long) (sbc - nseg) * 272) + boff;
        wh = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) src);
        wq = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) (src + 16));
        we = __builtin_nontemporal_load((const GLOBAL_AS uint32_t *) (src + 32));
    };
    // lane j's block against the activation's block j, then the chunk's eight terms in block order from 0 (q80_q80_sb_dot's float sequence): in the group's lane 0
    auto chunk_dot = [&](const u32x4 & h, const u32x4 & q, uint32_t e, const xblk80 * xb) -> float {
        const bool odd = (j8 & 1) != 0;
        const uint32_t D[9] = { h.x, h.y, h.z, h.w, q.x, q.y, q.z, q.w, e };
        const float dw = h2f((uint16_t) (odd ? (D[0] >> 16) : (D[0] & 0xffff)));
        const u32x4 y0 = *(const u32x4 *) (xb->q + 32 * j8), y1 = *(const u32x4 *) (xb->q + 32 * j8 + 16);
        const uint32_t y[8] = { y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w };
        int sumi = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t qd = odd ? D[t + 1] : __builtin_amdgcn_alignbyte(D[t + 1], D[t], 2);
            sumi = dot4_i8((int) qd, (int) y[t], sumi);
        }
        const float term = (float) sumi * (dw * xb->d[j8]);
        float sumf = 0.f;
        sumf += term;
        sumf += dpp_f32<0x101>(term); sumf += dpp_f32<0x102>(term); sumf += dpp_f32<0x103>(term); sumf += dpp_f32<0x104>(term);
        sumf += dpp_f32<0x105>(term); sumf += dpp_f32<0x106>(term); sumf += dpp_f32<0x107>(term);
        return sumf;
    };
    auto dots = [&](int nb, int nall) {
        if (wave * 8 >= nall) return;   // (wave-uniform)
        const int sb = wave * 8 + (lane >> 3);
        const int sbc = sb < nall ? sb : nall - 1;
        const float r = chunk_dot(wh, wq, we, xs + (sbc % nb));
        if (j8 == 0 && sb < nall) part[sb] = r;
    };
    // the first phase's weights go out before anything else
    request(ld_w(0), 4, 12, 0);

    auto poll_blocks = [&](auto nbc, int p, float v[4], int delay) {   // block b = wave (NB <= 8): values 256 b + 4 lane .. + 3
        constexpr int NB = decltype(nbc)::value;
        static_assert(NB <= CH_NCW, "one block per wave");
        for (int i = 0; i < delay; i++) __builtin_amdgcn_s_sleep(1);
        const unsigned tag_in = tag_base | (unsigned) p;
        const unsigned in_base = (unsigned) ((p - 1) & 1) * (CH_XF_MAX * 8u);
        const bool has0 = wave < NB;
        u32x4 g0, g1;
        g0 = g1 = (u32x4) { 0u, tag_in, 0u, tag_in };
        auto issue = [&]() {
            if (has0) {
                const unsigned o0 = in_base + ((unsigned) wave * 256u + (unsigned) lane * 4u) * 8u;
                g0 = ld16_agent(gb, o0); g1 = ld16_agent(gb, o0 + 16u);
            }
        };
        issue();
        unsigned spins = 0;
        for (;;) {
            const bool ok = g0.y == tag_in && g0.w == tag_in && g1.y == tag_in && g1.w == tag_in;
            if (__all(ok)) break;
            if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
            __builtin_amdgcn_s_sleep(1);
            issue();
        }
        settle_vmcnt();
        v[0] = __uint_as_float(g0.x); v[1] = __uint_as_float(g0.z); v[2] = __uint_as_float(g1.x); v[3] = __uint_as_float(g1.z);
    };
    auto load_alpha = [&](const float * alpha) -> f32x4 {
        f32x4 al = (f32x4) { 1.f, 1.f, 1.f, 1.f };
        if (wave < 4) al = *(const GLOBAL_AS f32x4 *) (gp(alpha) + wave * 256 + lane * 4);
        return al;
    };
    // RMS norm over 1024 values (blocks on waves 0 .. 3, matvec_q4k_kernel's order) / none, then Q8_0 blocks into xs; ends behind the "blocks ready" barrier
    auto norm_quant = [&](auto nbc, auto rmsc, float v[4], const f32x4 al, float eps) {
        constexpr int NB = decltype(nbc)::value;
        constexpr bool RMS = decltype(rmsc)::value;
        if (RMS) {
            static_assert(!RMS || NB == 4, "the norms of the Depth transformer are 1024 wide");
            double acc = 0;
            if (wave < NB)
#pragma unroll
                for (int k = 0; k < 4; k++) acc += (double) (v[k] * v[k]);
            acc = wave_allsum_f64(acc);
            if (lane == 0) ctl->sumsq[wave] = acc;
            nbar();
            double sq[CH_NCW];
#pragma unroll
            for (int w = 0; w < CH_NCW; w += 2) { const double2 t = *(const double2 *) &ctl->sumsq[w]; sq[w] = t.x; sq[w + 1] = t.y; }
            asm volatile("" : "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]), "+v"(sq[4]), "+v"(sq[5]), "+v"(sq[6]), "+v"(sq[7]));
            double tot = 0;
#pragma unroll
            for (int w = 0; w < CH_NCW; w++) tot += sq[w];
            const float mean = (float) (tot * (1.0 / 1024.0));
            const float scale = 1.0f / sqrtf(mean + eps);
            const float a4[4] = { al.x, al.y, al.z, al.w };
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = a4[k] * (v[k] * scale);
        }
        if (wave < NB) quantize_block_q80(xs + wave, v, lane);
        nbar();
    };
    auto rowsum = [&](int nb, int rows, int p, float * y, bool res, bool save, bool pub, float & best, int & bi) {
        const unsigned tag_out = tag_base | (unsigned) (p + 1);
        const unsigned pub_base = (unsigned) (p & 1) * CH_XF_MAX;
        const long long row0 = (long long) wg * rows;
        for (int rr = tid >> 4; rr < rows; rr += CH_NCW * 4) {
            float sum = 0.f;
            const int j = tid & 15;
            if (j < nb) sum += part[rr * nb + j];
            sum = row16_allsum_f32(sum);
            if ((tid & 15) == 0) {
                const long long row = row0 + rr;
                if (res) sum = xres[rr] + sum;
                if (save) xres[rr] = sum;
                if (pub) st_granule(P.gbuf + pub_base + row, tag_out, __float_as_uint(sum));
                gp(y)[row] = sum;
                if (sum >= best) { best = sum; bi = (int) row; }
            }
        }
    };

    // =================================================================================================================================
    // hoist: din_set = depformer_in[set] x transformer_out for every DISTINCT depformer_in matrix of the schedule (4 rows x NBIN chunks per set and workgroup)
    {
        constexpr int CPS = 4 * NBIN;               // chunks of a set per workgroup
        constexpr int SPP = 64 / CPS;               // sets per pass of 64 chunks
        static_assert(64 % CPS == 0 && NBIN <= CH_NCW, "whole sets per pass; one activation block per wave");
        float v[4] = { 0.f, 0.f, 0.f, 0.f };
        if (wave < NBIN) { const f32x4 t = *(const GLOBAL_AS f32x4 *) (gp(N.din_x) + wave * 256 + lane * 4); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        norm_quant(std::integral_constant<int, NBIN>(), std::false_type(), v, (f32x4) { 1.f, 1.f, 1.f, 1.f }, 0.f);
        const __amdgpu_buffer_rsrc_t db = make_rsrc(N.din_buf, (unsigned) N.n_sets * 1024u * 8u);
        const unsigned tag_d = tag_base | 1u;
#pragma unroll 1
        for (int s0 = 0; s0 < N.n_sets; s0 += SPP) {
            const int c = wave * 8 + (lane >> 3);                     // chunk of the pass
            const int set = s0 + c / CPS, cs = c % CPS;               // its set, and its index inside the set's 4 rows
            const int setc = set < N.n_sets ? set : N.n_sets - 1;
            const char * w = t_dw[setc];
            const GLOBAL_AS char * src = gp(w) + (long long) wg * 4 * (NBIN * 272) + (long long) cs * 272 + boff;
            const u32x4 dh = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) src), dq = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) (src + 16));
            const uint32_t de = __builtin_nontemporal_load((const GLOBAL_AS uint32_t *) (src + 32));
            if (s0 > 0) nbar();                                       // part[] of the previous pass has been read
            const float r = chunk_dot(dh, dq, de, xs + (cs % NBIN));
            if (j8 == 0) part[c] = r;
            nbar();
            {   // SPP sets x 4 rows, NBIN partials each: one 16-lane group per row
                const int r16 = tid >> 4, j = tid & 15;
                if (r16 < SPP * 4) {
                    float sum = 0.f;
                    if (j < NBIN) sum += part[r16 * NBIN + j];
                    sum = row16_allsum_f32(sum);
                    const int st_ = s0 + r16 / 4;
                    if (j == 0 && st_ < N.n_sets) st_granule(N.din_buf + (size_t) st_ * 1024 + (size_t) wg * 4 + (r16 & 3), tag_d, __float_as_uint(sum));
                }
            }
        }
        const int npairs = N.n_sets * 512;
        for (int base = 0; base < npairs; base += 8 * CH_THREADS) {
            unsigned spins = 0;
            for (;;) {
                u32x4 g[8];
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int pi = base + it * CH_THREADS + tid;
                    g[it] = ld16_agent(db, (unsigned) (pi < npairs ? pi : npairs - 1) * 16u);
                }
                bool ok = true;
#pragma unroll
                for (int it = 0; it < 8; it++) ok = ok && g[it].y == tag_d && g[it].w == tag_d;
                if (__all(ok)) {
#pragma unroll
                    for (int it = 0; it < 8; it++) {
                        const int pi = base + it * CH_THREADS + tid;
                        if (pi < npairs) *(float2 *) (din_all + 2 * pi) = make_float2(__uint_as_float(g[it].x), __uint_as_float(g[it].z));
                    }
                    break;
                }
                if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            settle_vmcnt();
        }
        nbar();
    }

    // =================================================================================================================================
    int p = 0, q = 0;   // p: phase counter (tags, buffer parity; attention phases count); q: index into the mat-vec phase table
#pragma unroll 1
    for (int s = 0; s < N.n_steps; s++) {
#pragma unroll 1
        for (int l = 0; l < L; l++) {
            // ---------------------------------------------------------------- in_proj: rms_norm -> 1024 -> 3072
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(q);
                const f32x4 al = load_alpha(ph.alpha);
                float v[4] = { 0.f, 0.f, 0.f, 0.f };
                if (l > 0) poll_blocks(std::integral_constant<int, 4>(), p, v, N.delay[0]);
                else {
                    // layer 0: x = din_set + last, last = the low-rank embedding of the previous step's token (lm_utils.h:208-217) or a vector in memory (step 0)
                    const n80_st st = ld_st(s);
                    const float * dk = din_all + st.din_set * 1024;
                    if (st.lr_table) {
                        // the 2 x 136 weight bytes of this thread's two rows of the 128 -> 1024 projection go out first: they do not depend on the token
                        const GLOBAL_AS uint32_t * wr = (const GLOBAL_AS uint32_t *) (gp(st.lr_w) + (long long) (2 * tid) * st.lr_w_row_bytes);
                        uint32_t W0[34], W1[34];
#pragma unroll
                        for (int i = 0; i < 34; i++) { W0[i] = wr[i]; W1[i] = wr[34 + i]; }
                        int64_t r;
                        if (st.emb_chain) {
                            if (wave == 0) {
                                int token = 0;
                                if (!gather_token(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token)) give_up();
                                if (lane == 0) {
                                    ctl->token = token;
                                    if (wg == 0) { if (st.prev_out[0]) *gp(st.prev_out[0]) = token; if (st.prev_out[1]) *gp(st.prev_out[1]) = token; }
                                }
                            }
                            nbar();
                            r = (int64_t) __builtin_amdgcn_readfirstlane(ctl->token);
                        } else r = (int64_t) *gp(st.lr_index);
                        const bool bad = r < 0 || r >= st.lr_n_rows;   // (get_rows hands on a NaN row)
                        // the table row, re-quantised as convert_rows does: 4 blocks of 32, d = amax / 127 kept as F16, q = roundf(x / d) - waves 0 and 1
                        if (wave < 2) {
                            const GLOBAL_AS char * row = gp(st.lr_table) + (bad ? 0 : r) * st.lr_row_bytes;
                            const float xv = dequant_elem_g(row, st.lr_type, wave * 64 + lane);
                            float amax = fabsf(xv);
                            amax = fmaxf(amax, dpp_f32<DPP_QUAD_XOR1>(amax));
                            amax = fmaxf(amax, dpp_f32<DPP_QUAD_XOR2>(amax));
                            amax = fmaxf(amax, dpp_f32<DPP_HALF_MIRROR>(amax));
                            amax = fmaxf(amax, dpp_f32<DPP_ROW_MIRROR>(amax));
                            amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                            const float d = amax / 127.f;
                            const float id = d ? 1.0f / d : 0.0f;
                            lrq[wave * 64 + lane] = (int8_t) roundf(xv * id);
                            if ((lane & 31) == 0) lrd[wave * 2 + (lane >> 5)] = h2f(f2h(d));
                        }
                        nbar();
                        {
                            const uint32_t * xq = (const uint32_t *) lrq;
                            float o[2];
#pragma unroll
                            for (int rr = 0; rr < 2; rr++) {
                                const uint32_t * W = rr ? W1 : W0;
                                float sumf = 0.f;
#pragma unroll
                                for (int ib = 0; ib < 4; ib++) {
                                    // block ib of the row: bytes 34 ib .. 34 ib + 33 of the 136 (dword 8.5 ib)
                                    const int d0 = (34 * ib) >> 2;
                                    const bool odd = (ib & 1) != 0;
                                    const float dw = h2f((uint16_t) (odd ? (W[d0] >> 16) : (W[d0] & 0xffff)));
                                    int sumi = 0;
#pragma unroll
                                    for (int t = 0; t < 8; t++) {
                                        const uint32_t qd = odd ? W[d0 + 1 + t] : __builtin_amdgcn_alignbyte(W[d0 + 1 + t], W[d0 + t], 2);
                                        sumi = dot4_i8((int) qd, (int) xq[ib * 8 + t], sumi);
                                    }
                                    sumf += sumi * (dw * lrd[ib]);
                                }
                                o[rr] = bad ? NAN : sumf;
                            }
                            *(float2 *) (emb + 2 * tid) = make_float2(o[0], o[1]);
                            if (wg == 0) { gp(st.lr_out)[2 * tid] = o[0]; gp(st.lr_out)[2 * tid + 1] = o[1]; }
                        }
                        nbar();
                        if (wave < 4) {
                            const f32x4 d4 = *(const f32x4 *) (dk + wave * 256 + lane * 4), e4 = *(const f32x4 *) (emb + wave * 256 + lane * 4);
                            v[0] = e4.x + d4.x; v[1] = e4.y + d4.y; v[2] = e4.z + d4.z; v[3] = e4.w + d4.w;
                        }
                    } else if (wave < 4) {
                        const f32x4 d4 = *(const f32x4 *) (dk + wave * 256 + lane * 4), e4 = *(const GLOBAL_AS f32x4 *) (gp(st.res_mem) + wave * 256 + lane * 4);
                        v[0] = e4.x + d4.x; v[1] = e4.y + d4.y; v[2] = e4.z + d4.z; v[3] = e4.w + d4.w;
                    }
                    if (wave == (wg >> 6) && lane == (wg & 63)) {   // rows 4 wg .. 4 wg + 3: this workgroup's part of the residual stream and of the node's storage
#pragma unroll
                        for (int k = 0; k < 4; k++) { xres[k] = v[k]; gp(st.din_y)[wg * 4 + k] = v[k]; }
                    }
                }
                CH_STAMP(2);
                norm_quant(std::integral_constant<int, 4>(), std::true_type(), v, al, ph.eps);
                CH_STAMP(5);
                dots(4, 48);
                request(ld_w(q + 1), 4, 4, 0);                         // out_proj (the attention phase in between has no weights)
                nbar();
                CH_STAMP(6);
                float best = -INFINITY; int bi = -1;
                rowsum(4, 12, p, ph.y, false, false, true, best, bi);
                CH_STAMP(8);
                p = __builtin_amdgcn_readfirstlane(p + 1); q = __builtin_amdgcn_readfirstlane(q + 1);
            }
            // ---------------------------------------------------------------- attention: head h on workgroup 16 h + (h & 7), everybody else moves on
            {
                CH_STAMP(10);
                const int h = wg >> 4;
                if ((wg & 15) == (h & 7)) {
                    const n80_at ta = ld_at(s * L + l);
                    attn_args at = N.at;
                    at.kcache = ta.kcache; at.vcache = ta.vcache; at.rot = ta.rot; at.mask = ta.mask; at.index = ta.index; at.out = ta.out;
                    at.q = nullptr; at.k = nullptr; at.v = nullptr;   // (element offsets come through gq)
                    for (int i = 0; i < N.delay[2]; i++) __builtin_amdgcn_s_sleep(1);
                    const attn_gqkv gq = { P.gbuf + (size_t) ((p - 1) & 1) * CH_XF_MAX, (int64_t) N.q_off, (int64_t) N.k_off, (int64_t) N.v_off, P.err };
                    const attn_gout go = { P.gbuf + (size_t) (p & 1) * CH_XF_MAX, tag_base | (unsigned) (p + 1), 0 };
                    __syncthreads();
                    CH_STAMP(2);
                    attn_ring64_body<AT_GQKV | AT_GOUT>(at, at_smem, h, tag_base | (unsigned) p, gq, go);
                    __syncthreads();
                    CH_STAMP(8);
                }
                p = __builtin_amdgcn_readfirstlane(p + 1);
            }
            // ---------------------------------------------------------------- out_proj 1024 -> 1024 + residual
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(q);
                float v[4];
                const int h = wg >> 4;
                poll_blocks(std::integral_constant<int, 4>(), p, v, (wg & 15) == (h & 7) ? 0 : N.delay[1]);
                CH_STAMP(2);
                norm_quant(std::integral_constant<int, 4>(), std::false_type(), v, (f32x4) { 1.f, 1.f, 1.f, 1.f }, 0.f);
                CH_STAMP(5);
                dots(4, 16);
                request(ld_w(q + 1), 4, FROWS, FF);                    // linear_in, paired
                nbar();
                CH_STAMP(6);
                float best = -INFINITY; int bi = -1;
                rowsum(4, 4, p, ph.y, true, true, true, best, bi);
                CH_STAMP(8);
                p = __builtin_amdgcn_readfirstlane(p + 1); q = __builtin_amdgcn_readfirstlane(q + 1);
            }
            // ---------------------------------------------------------------- linear_in, paired: rms_norm -> 1024 -> 2 x FF -> silu(l) * r
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(q);
                const f32x4 al = load_alpha(ph.alpha);
                float v[4];
                poll_blocks(std::integral_constant<int, 4>(), p, v, N.delay[0]);
                CH_STAMP(2);
                norm_quant(std::integral_constant<int, 4>(), std::true_type(), v, al, ph.eps);
                CH_STAMP(5);
                dots(4, 2 * FROWS * 4);
                request(ld_w(q + 1), NBF, 4, 0);                       // linear_out
                nbar();
                CH_STAMP(6);
                {
                    const unsigned tag_out = tag_base | (unsigned) (p + 1);
                    const unsigned pub_base = (unsigned) (p & 1) * CH_XF_MAX;
                    const long long row0 = (long long) wg * FROWS;
                    for (int rr = tid >> 4; rr < FROWS; rr += CH_NCW * 4) {
                        float sl = 0.f, sr = 0.f;
                        const int j = tid & 15;
                        if (j < 4) { sl += part[rr * 4 + j]; sr += part[(FROWS + rr) * 4 + j]; }
                        sl = row16_allsum_f32(sl); sr = row16_allsum_f32(sr);
                        if ((tid & 15) == 0) {
                            const float g = (sl / (1.0f + expf(-sl))) * sr;
                            st_granule(P.gbuf + pub_base + row0 + rr, tag_out, __float_as_uint(g));
                            gp(ph.y)[row0 + rr] = sl; gp(ph.y)[FF + row0 + rr] = sr;
                        }
                    }
                }
                CH_STAMP(8);
                p = __builtin_amdgcn_readfirstlane(p + 1); q = __builtin_amdgcn_readfirstlane(q + 1);
            }
            // ---------------------------------------------------------------- linear_out FF -> 1024 + residual
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(q);
                float v[4];
                poll_blocks(std::integral_constant<int, NBF>(), p, v, N.delay[0]);
                CH_STAMP(2);
                norm_quant(std::integral_constant<int, NBF>(), std::false_type(), v, (f32x4) { 1.f, 1.f, 1.f, 1.f }, 0.f);
                CH_STAMP(5);
                dots(NBF, 4 * NBF);
                if (q + 1 < n_ph) request(ld_w(q + 1), 4, l + 1 < L ? 12 : 8, 0);   // the next layer's in_proj, or this step's linears[k]
                nbar();
                CH_STAMP(6);
                float best = -INFINITY; int bi = -1;
                rowsum(NBF, 4, p, ph.y, true, true, true, best, bi);
                CH_STAMP(8);
                p = __builtin_amdgcn_readfirstlane(p + 1); q = __builtin_amdgcn_readfirstlane(q + 1);
            }
        }
        // -------------------------------------------------------------------- linears[k]: 1024 -> 2048 -> arg-max candidate
        {
            CH_STAMP(10);
            const nest_ph ph = ld_ph(q);
            float v[4];
            poll_blocks(std::integral_constant<int, 4>(), p, v, N.delay[0]);
            CH_STAMP(2);
            norm_quant(std::integral_constant<int, 4>(), std::false_type(), v, (f32x4) { 1.f, 1.f, 1.f, 1.f }, 0.f);
            CH_STAMP(5);
            dots(4, 32);
            if (s + 1 < N.n_steps) request(ld_w(q + 1), 4, 12, 0);
            nbar();
            float best = -INFINITY; int bi = -1;
            rowsum(4, 8, p, ph.y, false, false, false, best, bi);
            // (the 8 rows' values sit in the first lanes of 8 sixteen-lane groups: they meet through LDS - a wave-wide merge costs twelve ds_bpermute round trips)
            if ((tid & 15) == 0 && (tid >> 4) < 8) { ctl->am_v[tid >> 4] = best; ctl->am_i[tid >> 4] = bi; }
            nbar();
            if (tid == 0) {
                for (int w = 1; w < 8; w++) am_merge(best, bi, ctl->am_v[w], ctl->am_i[w]);
                u64 * c = P.cand + (size_t) (p & 1) * 2 * grid + 2 * wg;
                st_granule(c, tag_base | (unsigned) (p + 1), __float_as_uint(best));
                st_granule(c + 1, tag_base | (unsigned) (p + 1), (unsigned) bi);
            }
            CH_STAMP(8);
            p = __builtin_amdgcn_readfirstlane(p + 1); q = __builtin_amdgcn_readfirstlane(q + 1);
        }
    }
    if (wg == 0 && wave == 0) {
        const n80_st st = ld_st(N.n_steps - 1);
        int token = 0;
        if (gather_token(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token)) {
            if (lane == 0) { if (st.argmax_out[0]) *gp(st.argmax_out[0]) = token; if (st.argmax_out[1]) *gp(st.argmax_out[1]) = token; }
        } else give_up();
    }
    if (wg == 0 && tid == 0) *gp(P.launch_seq) = launch + 1u;
    (void) per_step;
}
