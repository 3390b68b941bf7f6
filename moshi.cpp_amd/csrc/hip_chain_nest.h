// hip_chain_nest.h — part of hip_chain.hip (included there, same translation unit): the chained Depth transformer of moshika / PersonaPlex
// (/root/reference/src/moshi/models/lm.h:446-553, weights per step lm_default.h:136-146) as a COMPILE-TIME step program.
//
// matvec_chain_kernel reads a 400-byte descriptor per phase and switches over seven phase bodies inside one loop. Two things cost there
// (profiles/r04_chain_stamps.txt, the disassembly): every body leaves the next phase's prefetched weights in registers of its own choice, so the
// loop's back edge moves them - behind an s_waitcnt vmcnt(0): a phase cannot issue its hand-off poll before ITS weights have landed (0.63 us) -
// and the descriptor fields are fetched from LDS one v_readfirstlane at a time at the head of every phase (0.22 us). Here the program is the
// loop nest itself
//     hoist: depformer_in[k] x transformer_out for ALL steps k (they do not depend on the chain, lm.h:505-512)
//     step k:  x = din_k + emb_k(token)   ->  layers x { in_proj, attention + out_proj, linear_in (paired gate), linear_out }  ->  linears[k] (no norm) -> arg-max
// with every shape a constant and one phase following another in straight-line code: the weights stay where the request put them, what a phase
// reads from a table is three pointers (compact records parked in LDS once per launch), and the poll of phase p + 1 goes out right behind the
// publication of phase p. The 8 (16) depformer_in products leave the chain: one wide phase at the head of the launch computes them for all steps,
// all workgroups exchange them once, and "+ emb_k(token)" becomes the input stage of layer 0's in_proj - a step is 4 L + 1 hand-offs instead of
// 4 L + 2, and the 4096-wide quantisation of transformer_out happens once per launch instead of once per step.
// Arithmetic: the generic chain kernel's (= the unchained kernels') to the bit - same Q8_K rounding, same per-super-block float expression, same
// 16-lane row sums, same attention, same last-maximum arg-max; tests/test_chain_engine.py compares the three plans bit for bit.
#pragma once

struct nest_ph { const char * w; const float * alpha; float * y; float eps; int pad; };                                  // one mat-vec phase
struct nest_at { char * kcache; char * vcache; const float * rot; const float * mask; const int32_t * index; int64_t pad; };   // one attention
struct nest_st {                                                                                                         // one step
    const char * din_w; float * din_y;      // depformer_in[k] and the storage of the node din_k + emb_k
    embed_src emb; int32_t * prev_out[2];   // the embedding row added to din_k; emb_chain: its index is the previous step's arg-max, stored to prev_out
    int32_t * argmax_out[2];                // where the LAST step's token goes (the others are their successors' prev_out)
    int emb_chain; int pad[3];
    const float * noise; float smp_scale; int smp_k;   // head_argmax = 2: this step's sampler (Exp(1) noise F32[k] uploaded per compute, 1 / temp, top-k)
};
static_assert(sizeof(nest_ph) == 32 && sizeof(nest_at) == 48 && sizeof(nest_st) % 16 == 0, "nest tables are copied to LDS by 16-byte lanes");

struct nest_params {
    chain_params P;                         // hand-off buffers, launch counter, error word (phases / n_phases unused)
    const u32x4 * tables;                   // device: nest_ph[n_steps * (4 L + 1)] | nest_at[n_steps * L] | nest_st[n_steps]
    int n_steps, n_layers;
    int head_argmax;                        // 0: linears[k] ends the run as plain logits (a sampler launch follows: temp > 0, one step per run); 1: greedy arg-max;
                                            // 2: the top-k sampler of sampling.h:4-64 as the tail of every linears[k] phase (nest_st::noise / smp_scale / smp_k)
    u64 * din_buf;                          // [n_steps][1024] granules: the hoisted depformer_in products
    const float * din_x;                    // transformer_out (lm.h:434), x of every depformer_in
    // the attention's shape, the same in every layer and step (checked when the plan is made)
    int q_off, k_off, v_off; int q_hs, k_hs, v_hs; int k_nb1, k_nb2, v_nb1, v_nb2; int C; float scale;
    // s_sleep units (64 cycles) a phase kind waits between its predecessor's publication and its first poll: [0] in_proj, [1] out_proj, [2] linear_in,
    // [3] linear_out, [4] linears[k]. A poll that samples before the slowest producer's stores are visible comes back empty and costs a whole fabric round
    // trip under the load of 256 workgroups all doing the same.
    int delay[5];
};

#define NEST_XF      3072                   // floats: the in_proj vector an attention phase gathers
#define NEST_PART    1024                   // super-block partial sums (the hoisted phase: 64 per step)
#define NEST_STEPS_MAX 16

// the shapes (K, M, pair) of the six mat-vec kinds at grid G: super-blocks of a workgroup and register passes of 64
template <class SH, int G> struct nest_dim {
    static constexpr int NB = SH::K / 256;
    static constexpr int ROWS = (SH::PAIR ? SH::PAIR : SH::M) / G;     // rows of a workgroup (paired: of each half)
    static constexpr int NSEG = ROWS * NB, NALL = SH::PAIR ? 2 * NSEG : NSEG;
    static constexpr int PASSES = (NALL + CH_NCW * 8 - 1) / (CH_NCW * 8);
    static_assert((SH::PAIR ? SH::PAIR : SH::M) % G == 0, "rows divide over the grid");
};

// Sampling mode: every workgroup's candidate of `tag` is a 64-bit key - (bits of q) << 32 | rank << 11 | token index, q = p / noise[rank] >= 0, 0 = none - whose
// maximum is the sampler's choice (the LAST maximum of q over the ranks, sampling.h:15 / ggml_vec_argmax_f32). One wave reads and merges them like gather_token.
__device__ __forceinline__ bool gather_sampled(__amdgpu_buffer_rsrc_t cb, unsigned base_bytes, int grid, unsigned tag, int lane, chain_ctl * ctl, int & token) {
    unsigned spins = 0;
    for (;;) {
        u32x4 c[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int g = i * 64 + lane;
            c[i] = ld16_agent(cb, base_bytes + (unsigned) (g < grid ? g : grid - 1) * 16u);
        }
        u64 key = 0;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int g = i * 64 + lane;
            ok = ok && c[i].y == tag && c[i].w == tag;
            const u64 kc = ((u64) c[i].x << 32) | (u64) c[i].z;
            if (g < grid && kc > key) key = kc;
        }
        if (__all(ok)) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const u64 ok2 = ((u64) (unsigned) __shfl_xor((int) (key >> 32), o, 64) << 32) | (u64) (unsigned) __shfl_xor((int) (unsigned) key, o, 64);
                key = ok2 > key ? ok2 : key;
            }
            token = (int) ((unsigned) key & 0x7ffu);
            return true;
        }
        if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { settle_vmcnt(); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Attention of a wave's TWO heads (h0, h0 + 1; 64 wide) for the one new token over the ring of <= 8 slots, both heads in ONE pass of the wave:
// lane = (head of the pair: lane >> 5, slot: (lane >> 2) & 7, 16-dim chunk: lane & 3). chain_attn_wave gives a head the whole wave (slot x 8-dim chunk) and
// runs the pair's two chains side by side - twice the soft-max chain, 16 per-element selects per head between ring row and new row - and the phase was bound by
// VALU issue (~450 instructions per wave, two waves per SIMD: profiles/r05_chain_stamps.txt, 2.1 us on each of the 48 attention phases). Here the new
// row's q / k / v come straight out of the hand-off poll's registers (lane l polled values 2 l, 2 l + 1 of the wave's 128 q, k and v values: no workgroup
// staging, no barrier in front), the new K / V rows are staged as packed BF16 (ring row vs. new row is a select on 8 + 8 dwords), and max / exp / sum /
// reciprocal run once for the pair.
// Arithmetic and ORDER are chain_attn_wave's (= attn_small_wave's, = the unchained kernel's), value for value:
//   * score: that function's lane sums 8 products sequentially in double, then butterflies over its 8 chunk lanes ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
//     a lane here holds chunks 2 c', 2 c' + 1: two sequential 8-term sums, their sum (that butterfly's first step), then the two remaining steps over 4 lanes;
//   * maximum: exact in any order; sum of the exponentials ((e0 + e1) + (e2 + e3)) + ((e4 + e5) + (e6 + e7)): half-mirror, row-mirror, the half's two rows;
//   * P x V: products rounded to float, the 8 slots added in slot order in double.
__device__ __forceinline__ void nest_attn_pair(float2 gq, float2 gk, float2 gv, const u32x4 kq[2], const u32x4 vq[2], int C, int slot_new, float m, float scale,
                                               bool write_cache, __amdgpu_buffer_rsrc_t kr, __amdgpu_buffer_rsrc_t vr, int k_nb1, int k_nb2, int v_nb1, int v_nb2,
                                               int h0, int lane, float * wbuf, float * xa, int log_wg = -1, int log_p = 0) {
    CH_ASTAMP(11);
    const int ah = lane >> 5, slot = (lane >> 2) & 7, chunk = lane & 3;
    // ---- the new row: BF16 roundings as the ring stores them (q is rounded too: ggml multiplies BF16 rows by a BF16 copy of q)
    float * wq = wbuf;                                   // [2][64] floats
    unsigned * wk = (unsigned *) (wbuf + 128), * wv = wk + 64;     // [2][32] packed BF16 pairs each
    float * prod = wbuf + 256;                           // [2][8][64] floats
    const unsigned kpack = (unsigned) f2bf(gk.x) | ((unsigned) f2bf(gk.y) << 16), vpack = (unsigned) f2bf(gv.x) | ((unsigned) f2bf(gv.y) << 16);
    *(float2 *) (wq + 2 * lane) = make_float2(bf2f(f2bf(gq.x)), bf2f(f2bf(gq.y)));
    wk[lane] = kpack; wv[lane] = vpack;
    if (write_cache && slot_new >= 0 && slot_new < C) {
        const int h = h0 + ah, d2 = (lane & 31) * 4;
        __builtin_amdgcn_raw_buffer_store_b32(kpack, kr, h * k_nb2 + slot_new * k_nb1 + d2, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(vpack, vr, h * v_nb2 + slot_new * v_nb1 + d2, 0, 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    CH_ASTAMP(12);
    // ---- every LDS read in front
    float qf[16];
    u32x4 kn[2], vn[2];
#pragma unroll
    for (int u = 0; u < 4; u++) { const float4 t = *(const float4 *) (wq + ah * 64 + chunk * 16 + 4 * u); qf[4 * u] = t.x; qf[4 * u + 1] = t.y; qf[4 * u + 2] = t.z; qf[4 * u + 3] = t.w; }
#pragma unroll
    for (int u = 0; u < 2; u++) { kn[u] = *(const u32x4 *) (wk + ah * 32 + chunk * 8 + 4 * u); vn[u] = *(const u32x4 *) (wv + ah * 32 + chunk * 8 + 4 * u); }
    const bool live = slot < C && m > -INFINITY;
    const bool fresh = slot == slot_new;
    unsigned kd[8], vd[8];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const unsigned a4[4] = { kq[u].x, kq[u].y, kq[u].z, kq[u].w }, b4[4] = { kn[u].x, kn[u].y, kn[u].z, kn[u].w };
        const unsigned c4[4] = { vq[u].x, vq[u].y, vq[u].z, vq[u].w }, d4[4] = { vn[u].x, vn[u].y, vn[u].z, vn[u].w };
#pragma unroll
        for (int i = 0; i < 4; i++) { kd[4 * u + i] = fresh ? b4[i] : a4[i]; vd[4 * u + i] = fresh ? d4[i] : c4[i]; }
    }
    // ---- scores
    double part[2];
#pragma unroll
    for (int hx = 0; hx < 2; hx++) {
        double a2 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int e = hx * 8 + i;
            const float kf = __uint_as_float((e & 1) ? (kd[e >> 1] & 0xffff0000u) : (kd[e >> 1] << 16));
            a2 += (double) (kf * qf[e]);
        }
        part[hx] = live ? a2 : 0.0;
    }
    double acc = part[0] + part[1];
    acc += dpp_f64<DPP_QUAD_XOR1>(acc);
    acc += dpp_f64<DPP_QUAD_XOR2>(acc);
    const float sv = live ? (float) acc * scale + m : -INFINITY;
    // ---- soft-max over the head's 8 slots (sv is uniform inside a slot's 4 lanes)
    float gm = fmaxf(sv, dpp_f32<DPP_HALF_MIRROR>(sv));
    gm = fmaxf(gm, dpp_f32<DPP_ROW_MIRROR>(gm));
    {
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), 48));
        gm = ah ? fmaxf(r2, r3) : fmaxf(r0, r1);
    }
    const float e = sv > -INFINITY ? expf(sv - gm) : 0.f;
    double ls = (double) e;
    ls += dpp_f64<DPP_HALF_MIRROR>(ls);
    ls += dpp_f64<DPP_ROW_MIRROR>(ls);
    {
        const int lo = __double2loint(ls), hi = __double2hiint(ls);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
        const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        ls = ah ? r2 + r3 : r0 + r1;
    }
    const float inv = (float) (1.0 / ls);
    const float pr = bf2f(f2bf(e * inv));
    // ---- P x V products (a slot whose probability is zero contributes exact zeros whatever its row holds)
    {
        // (pr != 0 ? v * pr : 0, with the select on the 8 packed words: a zeroed row gives 0 * pr = +0 for pr = +0, the products otherwise)
        float pf[16];
#pragma unroll
        for (int i = 0; i < 8; i++) vd[i] = pr != 0.f ? vd[i] : 0u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float vf = __uint_as_float((i & 1) ? (vd[i >> 1] & 0xffff0000u) : (vd[i >> 1] << 16));
            pf[i] = vf * pr;
        }
        float * dst = prod + ah * 512 + slot * 64 + chunk * 16;
#pragma unroll
        for (int u = 0; u < 4; u++) *(float4 *) (dst + 4 * u) = make_float4(pf[4 * u], pf[4 * u + 1], pf[4 * u + 2], pf[4 * u + 3]);
    }
    CH_ASTAMP(13);
    CH_ASTAMP(14);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {   // lane l: dims 2 (l & 31), + 1 of head l >> 5, the 8 slots in slot order
        float2 t[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; c2++) t[c2] = *(const float2 *) (prod + ah * 512 + c2 * 64 + 2 * (lane & 31));
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int c2 = 0; c2 < 8; c2++) { t0 += (double) t[c2].x; t1 += (double) t[c2].y; }
        *(float2 *) (xa + 2 * lane) = make_float2((float) t0, (float) t1);
    }
    CH_ASTAMP(15);
}

// SMP: the sampling-mode instance (head_argmax = 2). A template argument, not a run-time test: with the sampler's tail as a run-time branch of the one kernel the
// greedy path measured 5 - 8 us per frame slower (more live registers across the head phase; same-box A/B profiles/r06_ab_lean_activation_loads.txt).
template <int G, bool SMP = false>
__global__ void __launch_bounds__(CH_THREADS) depth_nest_kernel(nest_params N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    constexpr int grid = G;
    const int L = N.n_layers, per_step = 4 * L + 1;
    const int n_ph = N.n_steps * per_step, n_at = N.n_steps * L;

    xblk * xs = (xblk *) smem;
    float * xf = (float *) (xs + 16);
    float * part = xf + NEST_XF;
    float * xres = part + NEST_PART;
    float * xa = xres + 16;
    float * attw = xa + 1024;
    chain_ctl * ctl = (chain_ctl *) (attw + CH_NCW * CH_ATTW);
    const nest_ph * t_ph = (const nest_ph *) (ctl + 1);
    const nest_at * t_at = (const nest_at *) (t_ph + n_ph);
    const nest_st * t_st = (const nest_st *) (t_at + n_at);
    float * din_all = (float *) (t_st + N.n_steps);                      // [n_steps][1024]

    const chain_params & P = N.P;
    // Workgroup rendezvous (LDS traffic only is waited for). There is NO early exit anywhere below: a wave that gives up a bounded wait raises ctl->failed (and
    // the host-visible error word), every later poll of the workgroup then leaves at its first look, and the program runs to its end on whatever it has. An exit
    // out of the loop nest would be cheap at run time but not at compile time: the compiler funnels all exits of a loop through one guard block that the normal
    // path shares, and its wait-count bookkeeping then assumes at the head of a phase whatever was in flight at ANY exit - the phase opens with s_waitcnt vmcnt(0).
    auto nbar = [&]() { lds_barrier(); };
    if (tid == 0) { ctl->failed = 0; ctl->token = 0; }
    {
        const int n16 = (n_ph * (int) sizeof(nest_ph) + n_at * (int) sizeof(nest_at) + N.n_steps * (int) sizeof(nest_st)) / 16;
        const GLOBAL_AS u32x4 * src = (const GLOBAL_AS u32x4 *) N.tables;
        for (int i = tid; i < n16; i += CH_THREADS) ((u32x4 *) t_ph)[i] = src[i];
    }
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    __syncthreads();

    // ---- table reads: one record -> scalar registers (LDS broadcast reads, v_readfirstlane)
    auto ld_ph = [&](int q) {
        const u32x4 a = ((const u32x4 *) (t_ph + q))[0], b = ((const u32x4 *) (t_ph + q))[1];
        nest_ph r;
        unsigned w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) w[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };
    auto ld_w = [&](int q) -> const char * {   // only the weight pointer (the next phase's, for the request)
        const unsigned * s = (const unsigned *) (t_ph + q);
        unsigned w[2] = { (unsigned) __builtin_amdgcn_readfirstlane((int) s[0]), (unsigned) __builtin_amdgcn_readfirstlane((int) s[1]) };
        const char * r;
        __builtin_memcpy(&r, w, 8);
        return r;
    };
    auto ld_at = [&](int q) {
        const u32x4 a = ((const u32x4 *) (t_at + q))[0], b = ((const u32x4 *) (t_at + q))[1], c = ((const u32x4 *) (t_at + q))[2];
        nest_at r;
        unsigned w[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
#pragma unroll
        for (int i = 0; i < 12; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) w[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };
    auto ld_st = [&](int q) {
        nest_st r;
        constexpr int NW = (int) sizeof(nest_st) / 4;
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

    // ---- the weights of ONE phase in registers: pass ps = super-blocks 64 ps .. 64 ps + 63 of the workgroup's rows, 8 lanes per super-block
    // (16-byte header in every lane of the group + its own 16-byte nibble chunk), as matvec_chain_kernel holds them
    u32x4 wh[2], wq[2];
    auto request = [&](auto shape_tag, const char * w, int rows_rt) {   // rows_rt: rows of this workgroup when two shapes share the request (K = 1024, unpaired)
        using SH = decltype(shape_tag);
        using D = nest_dim<SH, G>;
        constexpr long long row_bytes = (long long) D::NB * 144;
        const int rows = rows_rt > 0 ? rows_rt : D::ROWS;
        const int nseg = rows * D::NB, nall = SH::PAIR ? 2 * nseg : nseg;
        const long long r0 = (long long) wg * rows;
        const GLOBAL_AS u32x4 * w0 = (const GLOBAL_AS u32x4 *) (gp(w) + r0 * row_bytes);
        const GLOBAL_AS u32x4 * w1 = (const GLOBAL_AS u32x4 *) (gp(w) + (r0 + SH::PAIR) * row_bytes);
#pragma unroll
        for (int ps = 0; ps < D::PASSES; ps++) {
            const int sb = ps * (CH_NCW * 8) + wave * 8 + (lane >> 3);
            const int sbc = sb < nall ? sb : nall - 1;
            const GLOBAL_AS u32x4 * src = (!SH::PAIR || sbc < nseg) ? w0 + sbc * 9 : w1 + (sbc - nseg) * 9;
            wh[ps] = __builtin_nontemporal_load(src);
            wq[ps] = __builtin_nontemporal_load(src + 1 + (lane & 7));
        }
    };
    // the first phase's weights go out before anything else
    request(shape_inproj(), ld_w(0), 0);

    // one super-block against its Q8_K block (the WS = 1 arithmetic of matvec_q4k_kernel): result in lane j8 == 0 of the group
    auto sb_dot = [&](const u32x4 & h, const u32x4 & q, const xblk * xb, float & out) {
        const int j8 = lane & 7, g32 = j8 >> 1, hf = j8 & 1;
        const uint32_t hw[4] = { h.x, h.y, h.z, h.w };
        uint32_t sc[2], mn[2];
        q4k_unpack_scales_w(hw[1], hw[2], hw[3], sc, mn);
        const u32x4 ylo = *(const u32x4 *) (xb->q + 64 * g32 + 16 * hf), yhi = *(const u32x4 *) (xb->q + 64 * g32 + 32 + 16 * hf);
        const uint32_t qw[4] = { q.x, q.y, q.z, q.w }, yl[4] = { ylo.x, ylo.y, ylo.z, ylo.w }, yh[4] = { yhi.x, yhi.y, yhi.z, yhi.w };
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
        const float d = h2f((uint16_t) (hw[0] & 0xffff)) * xb->d, dmin = h2f((uint16_t) (hw[0] >> 16)) * xb->d;
        out = d * (float) isum - dmin * (float) msum;
    };
    // the dots of a phase out of wh / wq into part[]
    auto dots = [&](auto shape_tag, int rows_rt) {
        using SH = decltype(shape_tag);
        using D = nest_dim<SH, G>;
        const int rows = rows_rt > 0 ? rows_rt : D::ROWS;
        const int nall = (SH::PAIR ? 2 : 1) * rows * D::NB;
#pragma unroll
        for (int ps = 0; ps < D::PASSES; ps++) {
            if (ps * (CH_NCW * 8) + wave * 8 >= nall) break;   // (wave-uniform)
            const int sb = ps * (CH_NCW * 8) + wave * 8 + (lane >> 3);
            const int sbc = sb < nall ? sb : nall - 1;
            float r;
            sb_dot(wh[ps], wq[ps], xs + (sbc % D::NB), r);
            if ((lane & 7) == 0 && sb < nall) part[sb] = r;
        }
    };

    // ---- the values of a hand-off straight into registers: block b = 256 values = lane l's granules 4 l .. 4 l + 3 = two 16-byte agent-scope loads;
    // waves take blocks w and w + 8. Only block-owning waves poll (every poll is a fabric read).
    auto poll_blocks = [&](auto nbc, int p, float v[2][4], int delay) {
        constexpr int NB = decltype(nbc)::value;
        for (int i = 0; i < delay; i++) __builtin_amdgcn_s_sleep(1);
        const unsigned tag_in = tag_base | (unsigned) p;
        const unsigned in_base = (unsigned) ((p - 1) & 1) * (CH_XF_MAX * 8u);
        const bool has0 = wave < NB, has1 = wave + CH_NCW < NB;
        u32x4 gq[2][2];
        gq[0][0] = gq[0][1] = gq[1][0] = gq[1][1] = (u32x4) { 0u, tag_in, 0u, tag_in };
        auto issue = [&]() {
            if (has0) {
                const unsigned o0 = in_base + ((unsigned) wave * 256u + (unsigned) lane * 4u) * 8u;
                gq[0][0] = ld16_agent(gb, o0); gq[0][1] = ld16_agent(gb, o0 + 16u);
                if (has1) {
                    const unsigned o1 = o0 + CH_NCW * 256u * 8u;
                    gq[1][0] = ld16_agent(gb, o1); gq[1][1] = ld16_agent(gb, o1 + 16u);
                }
            }
        };
        issue();
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 2; r++) ok = ok && gq[r][0].y == tag_in && gq[r][0].w == tag_in && gq[r][1].y == tag_in && gq[r][1].w == tag_in;
            if (__all(ok)) break;
            if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
            __builtin_amdgcn_s_sleep(1);
            issue();
        }
        settle_vmcnt();
        // (read unconditionally - a wave without the block holds the zeros it was initialised with: a use under the same condition as the load leaves, as far
        // as the compiler's wait-count bookkeeping can tell, a path on which the load is never waited for, and the NEXT phase then starts behind vmcnt(0))
#pragma unroll
        for (int r = 0; r < 2; r++) {
            v[r][0] = __uint_as_float(gq[r][0].x); v[r][1] = __uint_as_float(gq[r][0].z);
            v[r][2] = __uint_as_float(gq[r][1].x); v[r][3] = __uint_as_float(gq[r][1].z);
        }
    };
    // norm weights of the blocks a wave owns (requested before the poll: they do not depend on the chain)
    auto load_alpha = [&](auto nbc, const float * alpha, f32x4 al[2]) {
        constexpr int NB = decltype(nbc)::value;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int b = wave + r * CH_NCW;
            al[r] = (f32x4) { 1.f, 1.f, 1.f, 1.f };
            if (b < NB) al[r] = *(const GLOBAL_AS f32x4 *) (gp(alpha) + b * 256 + lane * 4);
        }
    };
    // RMS norm (matvec_q4k_kernel's order: per-thread squares in double, wave butterfly, waves added in index order) + Q8_K blocks into xs.
    // Ends behind the "blocks ready" barrier.
    auto norm_quant = [&](auto nbc, auto rmsc, float v[2][4], const f32x4 al[2], float eps) {
        constexpr int NB = decltype(nbc)::value;
        constexpr bool RMS = decltype(rmsc)::value;
        if (RMS) {
            double acc = 0;
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (wave + r * CH_NCW < NB)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc += (double) (v[r][k] * v[r][k]);
            acc = wave_allsum_f64(acc);
            if (lane == 0) ctl->sumsq[wave] = acc;
            nbar();
            double sq[CH_NCW];   // (all eight read before the first add: one LDS round trip instead of four)
#pragma unroll
            for (int w = 0; w < CH_NCW; w += 2) { const double2 t = *(const double2 *) &ctl->sumsq[w]; sq[w] = t.x; sq[w + 1] = t.y; }
            asm volatile("" : "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]), "+v"(sq[4]), "+v"(sq[5]), "+v"(sq[6]), "+v"(sq[7]));
            double tot = 0;
#pragma unroll
            for (int w = 0; w < CH_NCW; w++) tot += sq[w];
            constexpr int K = NB * 256;
            static_assert(!RMS || (K & (K - 1)) == 0, "the mean is a product with 1 / K");
            const float mean = (float) (tot * (1.0 / (double) K));
            const float scale = 1.0f / sqrtf(mean + eps);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (r * CH_NCW >= NB) continue;   // (compile-time; a wave without the block scales its zeros by the ones al[] was initialised with - see poll_blocks)
                const float a4[4] = { al[r].x, al[r].y, al[r].z, al[r].w };
#pragma unroll
                for (int k = 0; k < 4; k++) v[r][k] = a4[k] * (v[r][k] * scale);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int b = wave + r * CH_NCW;
            if (b < NB) quantize_block_q8k(xs + b, v[r], lane);
        }
        nbar();
    };
    // fixed-order row sums of an unpaired phase + epilogue + publication; rows_rt as in request()
    auto rowsum = [&](auto shape_tag, int rows_rt, int p, float * y, float & best, int & bi, bool pub_rt = false) {   // pub_rt: publish although the shape does not (sampled logits)
        using SH = decltype(shape_tag);
        using D = nest_dim<SH, G>;
        const int rows = rows_rt > 0 ? rows_rt : D::ROWS;
        const unsigned tag_out = tag_base | (unsigned) (p + 1);
        const unsigned pub_base = (unsigned) (p & 1) * CH_XF_MAX;
        const long long row0 = (long long) wg * rows;
        for (int rr = tid >> 4; rr < rows; rr += CH_NCW * 4) {
            float sum = 0.f;
#pragma unroll
            for (int j0 = 0; j0 < D::NB; j0 += 16) { const int j = j0 + (tid & 15); if (j < D::NB) sum += part[rr * D::NB + j]; }
            sum = row16_allsum_f32(sum);
            if ((tid & 15) == 0) {
                const long long row = row0 + rr;
                if (SH::RES == 1) sum = xres[rr] + sum;
                if (SH::SAVE) xres[rr] = sum;
                if (SH::PUB || pub_rt) st_granule(P.gbuf + pub_base + row, tag_out, __float_as_uint(sum));
                gp(y)[row] = sum;
                if (sum >= best) { best = sum; bi = (int) row; }   // rows ascend per thread: '>=' keeps the last maximum
            }
        }
    };

    // =================================================================================================================================
    // hoist: din_k = depformer_in[k] x transformer_out for every step k (shape_din's dots: K = 4096, 4 rows of each matrix per workgroup = one
    // register pass per step), published once, gathered by every workgroup into din_all
    {
        const int p = 511; (void) p;   // (diagnostic build: the hoist logs into the last record)
        CH_STAMP(10);
        using DD = nest_dim<shape_din, G>;
        static_assert(DD::NALL == 64 && DD::NB == 16, "the hoisted phase: one register pass of 64 super-blocks per step");
        float v[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const f32x4 t = *(const GLOBAL_AS f32x4 *) (gp(N.din_x) + (wave + r * CH_NCW) * 256 + lane * 4);
            v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
        }
        const __amdgpu_buffer_rsrc_t db = make_rsrc(N.din_buf, (unsigned) N.n_steps * 1024u * 8u);
        const unsigned tag_d = tag_base | 1u;
        constexpr long long row_bytes = 16 * 144;
#pragma unroll 1
        for (int k0 = 0; k0 < N.n_steps; k0 += 8) {
            u32x4 dh[8], dq[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = k0 + i < N.n_steps ? k0 + i : N.n_steps - 1;
                const char * w = *(const char * const *) &t_st[k].din_w;   // (LDS; uniform)
                const GLOBAL_AS u32x4 * src = (const GLOBAL_AS u32x4 *) (gp(w) + (long long) wg * DD::ROWS * row_bytes) + (wave * 8 + (lane >> 3)) * 9;
                dh[i] = __builtin_nontemporal_load(src);
                dq[i] = __builtin_nontemporal_load(src + 1 + (lane & 7));
            }
            if (k0 == 0) {
                const f32x4 one[2] = { (f32x4) { 1.f, 1.f, 1.f, 1.f }, (f32x4) { 1.f, 1.f, 1.f, 1.f } };
                norm_quant(std::integral_constant<int, 16>(), std::false_type(), v, one, 0.f);
            } else nbar();   // part[] of the previous round has been read
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (k0 + i >= N.n_steps) break;
                const int sb = wave * 8 + (lane >> 3);
                float r;
                sb_dot(dh[i], dq[i], xs + (sb % 16), r);
                if ((lane & 7) == 0) part[i * 64 + sb] = r;
            }
            nbar();
            {   // 4 rows x (up to) 8 steps = 32 row sums of 16 partials: one per 16-lane row of the workgroup
                const int r32 = tid >> 4, i = r32 >> 2, rr = r32 & 3;
                float sum = part[i * 64 + rr * 16 + (tid & 15)];
                sum = row16_allsum_f32(sum);
                if ((tid & 15) == 0 && k0 + i < N.n_steps) st_granule(N.din_buf + (size_t) (k0 + i) * 1024 + (size_t) wg * 4 + rr, tag_d, __float_as_uint(sum));
            }
        }
        CH_STAMP(0);
        // everybody's products of every step -> din_all
        const int npairs = N.n_steps * 512;
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
        }
        CH_STAMP(9);
        nbar();
    }

    // =================================================================================================================================
    int p = 0;   // phase counter: phase p reads what carries tag p and publishes with tag p + 1, the two hand-off buffers alternate by parity
#pragma unroll 1
    for (int s = 0; s < N.n_steps; s++) {
#pragma unroll 1
        for (int l = 0; l < L; l++) {
            // ---------------------------------------------------------------- in_proj: rms_norm -> 1024 -> 3072
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(p);
                f32x4 al[2];
                load_alpha(std::integral_constant<int, 4>(), ph.alpha, al);
                CH_STAMP(0);
                float v[2][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
                if (l > 0) {
                    CH_STAMP(1);
                    poll_blocks(std::integral_constant<int, 4>(), p, v, N.delay[0]);
                    CH_STAMP(2);
                } else {
                    // layer 0: x = din_k + emb_k(token) (lm.h:512-526), the token being the previous step's arg-max (or an index in memory: step 0)
                    const nest_st st = ld_st(s);
                    int64_t r;
                    if (st.emb_chain) {
                        if (wave == 0) {
                            int token = 0;
                            const bool got = SMP ? gather_sampled(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token)
                                                                : gather_token(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token);
                            if (!got) give_up();
                            if (lane == 0) {
                                ctl->token = token;
                                if (wg == 0) { if (st.prev_out[0]) *gp(st.prev_out[0]) = token; if (st.prev_out[1]) *gp(st.prev_out[1]) = token; }
                            }
                        }
                        CH_STAMP(1);
                        nbar();
                        r = (int64_t) __builtin_amdgcn_readfirstlane(ctl->token);
                    } else { CH_STAMP(1); r = (int64_t) *gp(st.emb.index); }
                    CH_STAMP(2);
                    if (r < 0 || r >= st.emb.n_rows) r = 0;
                    const GLOBAL_AS char * emb_row = gp(st.emb.table) + r * st.emb.row_bytes;
                    float emb_scale = 1.f;
                    if (st.emb.scale) emb_scale = *gp(st.emb.scale);
                    const float * dk = din_all + s * 1024;
                    if (wave < 4) {
                        const f32x4 d4 = *(const f32x4 *) (dk + wave * 256 + lane * 4);
                        const float d[4] = { d4.x, d4.y, d4.z, d4.w };
                        float e[4];
                        dequant4_g(emb_row, st.emb.type, wave * 256 + lane * 4, e);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (st.emb.scale) e[k] = e[k] * emb_scale;
                            v[0][k] = d[k] + e[k];
                        }
                        // rows 4 wg .. 4 wg + 3 are this workgroup's part of the residual stream (and of the node's own storage): exactly one lane's four values
                        if (wave == (wg >> 6) && lane == (wg & 63)) {
#pragma unroll
                            for (int k = 0; k < 4; k++) { xres[k] = v[0][k]; gp(st.din_y)[wg * 4 + k] = v[0][k]; }
                        }
                    }
                }
                CH_STAMP(3);
                norm_quant(std::integral_constant<int, 4>(), std::true_type(), v, al, ph.eps);
                CH_STAMP(5);
                dots(shape_inproj(), 0);
                request(shape_outproj(), ld_w(p + 1), 0);
                CH_STAMP(6);
                nbar();
                CH_STAMP(7);
                float best = -INFINITY; int bi = -1;
                rowsum(shape_inproj(), 0, p, ph.y, best, bi);
                CH_STAMP(8);
                CH_STAMP(9);
                p = __builtin_amdgcn_readfirstlane(p + 1);
            }
            // ---------------------------------------------------------------- attention over the ring of <= 8 + out_proj 1024 -> 1024 + residual
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(p);
                const nest_at ta = ld_at(s * L + l);
                CH_STAMP(0);
                // what does not depend on the chain goes out before the wait: the ring rows of the wave's two heads as earlier steps of this frame (and earlier
                // frames, where the ring holds them) left them - 16 dims of one (head, slot) per lane, see nest_attn_pair -, the ring slot, the mask row
                const int a_h = lane >> 5, a_slot = (lane >> 2) & 7, a_cc = a_slot < N.C ? a_slot : N.C - 1;
                const __amdgpu_buffer_rsrc_t kr = make_rsrc(ta.kcache, (unsigned) (2 * CH_NCW * N.k_nb2));
                const __amdgpu_buffer_rsrc_t vr = make_rsrc(ta.vcache, (unsigned) (2 * CH_NCW * N.v_nb2));
                u32x4 kq[2], vq[2];
                {
                    const unsigned ko = (unsigned) ((wave * 2 + a_h) * N.k_nb2 + a_cc * N.k_nb1 + (lane & 3) * 32);
                    const unsigned vo = (unsigned) ((wave * 2 + a_h) * N.v_nb2 + a_cc * N.v_nb1 + (lane & 3) * 32);
                    kq[0] = ld16_agent(kr, ko); kq[1] = ld16_agent(kr, ko + 16u);
                    vq[0] = ld16_agent(vr, vo); vq[1] = ld16_agent(vr, vo + 16u);
                }
                const int at_slot = gp(ta.index)[0];
                const float at_m = gp(ta.mask)[a_cc];
                CH_STAMP(1);
                for (int i = 0; i < N.delay[1]; i++) __builtin_amdgcn_s_sleep(1);
                // the in_proj rows of the wave's own two heads, straight into registers: pair lane of the wave's 128 q values, of its 128 k and of its 128 v values
                float2 gq, gk, gv;
                {
                    const unsigned tag_in = tag_base | (unsigned) p;
                    const unsigned base = (unsigned) ((p - 1) & 1) * (CH_XF_MAX * 8u) + (unsigned) (wave * 64 + lane) * 16u;
                    u32x4 g[3];
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int it = 0; it < 3; it++) g[it] = ld16_agent(gb, base + (unsigned) it * (512u * 16u));
                        bool ok = true;
#pragma unroll
                        for (int it = 0; it < 3; it++) ok = ok && g[it].y == tag_in && g[it].w == tag_in;
                        if (__all(ok)) break;
                        if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    settle_vmcnt();
                    gq = make_float2(__uint_as_float(g[0].x), __uint_as_float(g[0].z));
                    gk = make_float2(__uint_as_float(g[1].x), __uint_as_float(g[1].z));
                    gv = make_float2(__uint_as_float(g[2].x), __uint_as_float(g[2].z));
                }
                CH_STAMP(2);
                nest_attn_pair(gq, gk, gv, kq, vq, N.C, at_slot, at_m, N.scale, wg == 0, kr, vr, N.k_nb1, N.k_nb2, N.v_nb1, N.v_nb2, wave * 2, lane,
                               attw + wave * CH_ATTW, xa + wave * 128
#if defined(CH_LOG)
                               , wg == 0 ? 0 : wg == grid - 1 ? 1 : -1, p
#endif
                               );
                nbar();
                CH_STAMP(3);
                float v[2][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
                if (wave < 4) { const f32x4 t = *(const f32x4 *) (xa + wave * 256 + lane * 4); v[0][0] = t.x; v[0][1] = t.y; v[0][2] = t.z; v[0][3] = t.w; }
                const f32x4 one[2] = { (f32x4) { 1.f, 1.f, 1.f, 1.f }, (f32x4) { 1.f, 1.f, 1.f, 1.f } };
                norm_quant(std::integral_constant<int, 4>(), std::false_type(), v, one, 0.f);
                CH_STAMP(5);
                dots(shape_outproj(), 0);
                // workgroup 0 wrote the new K / V ring rows with agent-scope stores: drained here, in front of every later publication (Guideline 16 R1)
                if (wg == 0) wait_vmcnt<0>();
                request(shape_linin(), ld_w(p + 1), 0);
                CH_STAMP(6);
                nbar();
                CH_STAMP(7);
                float best = -INFINITY; int bi = -1;
                rowsum(shape_outproj(), 0, p, ph.y, best, bi);
                CH_STAMP(8);
                CH_STAMP(9);
                p = __builtin_amdgcn_readfirstlane(p + 1);
            }
            // ---------------------------------------------------------------- linear_in, paired: rms_norm -> 1024 -> 2 x 2816 -> silu(l) * r
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(p);
                f32x4 al[2];
                load_alpha(std::integral_constant<int, 4>(), ph.alpha, al);
                CH_STAMP(0);
                float v[2][4];
                CH_STAMP(1);
                poll_blocks(std::integral_constant<int, 4>(), p, v, N.delay[2]);
                CH_STAMP(2);
                CH_STAMP(3);
                norm_quant(std::integral_constant<int, 4>(), std::true_type(), v, al, ph.eps);
                CH_STAMP(5);
                dots(shape_linin(), 0);
                request(shape_linout(), ld_w(p + 1), 0);
                CH_STAMP(6);
                nbar();
                CH_STAMP(7);
                {
                    using D = nest_dim<shape_linin, G>;
                    const unsigned tag_out = tag_base | (unsigned) (p + 1);
                    const unsigned pub_base = (unsigned) (p & 1) * CH_XF_MAX;
                    const long long row0 = (long long) wg * D::ROWS;
                    for (int rr = tid >> 4; rr < D::ROWS; rr += CH_NCW * 4) {
                        float sl = 0.f, sr = 0.f;
                        const int j = tid & 15;
                        if (j < D::NB) { sl += part[rr * D::NB + j]; sr += part[(D::ROWS + rr) * D::NB + j]; }
                        sl = row16_allsum_f32(sl); sr = row16_allsum_f32(sr);
                        if ((tid & 15) == 0) {
                            const float g = (sl / (1.0f + expf(-sl))) * sr;
                            st_granule(P.gbuf + pub_base + row0 + rr, tag_out, __float_as_uint(g));
                            gp(ph.y)[row0 + rr] = sl; gp(ph.y)[shape_linin::PAIR + row0 + rr] = sr;
                        }
                    }
                }
                CH_STAMP(8);
                CH_STAMP(9);
                p = __builtin_amdgcn_readfirstlane(p + 1);
            }
            // ---------------------------------------------------------------- linear_out 2816 -> 1024 + residual
            {
                CH_STAMP(10);
                const nest_ph ph = ld_ph(p);
                CH_STAMP(0);
                float v[2][4];
                CH_STAMP(1);
                poll_blocks(std::integral_constant<int, 11>(), p, v, N.delay[3]);
                CH_STAMP(2);
                CH_STAMP(3);
                const f32x4 one[2] = { (f32x4) { 1.f, 1.f, 1.f, 1.f }, (f32x4) { 1.f, 1.f, 1.f, 1.f } };
                norm_quant(std::integral_constant<int, 11>(), std::false_type(), v, one, 0.f);
                CH_STAMP(5);
                dots(shape_linout(), 0);
                // what follows: the next layer's in_proj (12 rows per workgroup) or this step's linears[k] (8 rows): both 1024 wide, one request
                if (p + 1 < n_ph) request(shape_inproj(), ld_w(p + 1), l + 1 < L ? 0 : nest_dim<shape_head, G>::ROWS);
                CH_STAMP(6);
                nbar();
                CH_STAMP(7);
                float best = -INFINITY; int bi = -1;
                rowsum(shape_linout(), 0, p, ph.y, best, bi);
                CH_STAMP(8);
                CH_STAMP(9);
                p = __builtin_amdgcn_readfirstlane(p + 1);
            }
        }
        // -------------------------------------------------------------------- linears[k]: 1024 -> 2048 -> arg-max candidate
        {
            CH_STAMP(10);
            const nest_ph ph = ld_ph(p);
            // sampling mode: this step's noise vector goes out now (rank j's noise in thread j; uploaded by the host in front of the graph, src/context.h:465-480)
            float smp_noise = 1.f, smp_scale = 1.f; int smp_k = 1;
            if (SMP) {
                const nest_st sst = ld_st(s);
                smp_scale = sst.smp_scale; smp_k = sst.smp_k;
                smp_noise = gp(sst.noise)[tid < smp_k ? tid : 0];
            }
            CH_STAMP(0);
            float v[2][4];
            CH_STAMP(1);
            poll_blocks(std::integral_constant<int, 4>(), p, v, N.delay[4]);
            CH_STAMP(2);
            CH_STAMP(3);
            const f32x4 one[2] = { (f32x4) { 1.f, 1.f, 1.f, 1.f }, (f32x4) { 1.f, 1.f, 1.f, 1.f } };
            norm_quant(std::integral_constant<int, 4>(), std::false_type(), v, one, 0.f);
            CH_STAMP(5);
            constexpr int HR = nest_dim<shape_head, G>::ROWS;
            dots(shape_inproj(), HR);
            if (s + 1 < N.n_steps) request(shape_inproj(), ld_w(p + 1), 0);
            CH_STAMP(6);
            nbar();
            CH_STAMP(7);
            float best = -INFINITY; int bi = -1;
            rowsum(shape_head(), 0, p, ph.y, best, bi, SMP);
            CH_STAMP(8);
            if (SMP) {
                // ---- the top-k sampler (moshi_sample_token, sampling.h:4-64) as the tail of the phase, spread over ALL workgroups: sample_topk_kernel's values
                // (soft-max in its summation order, ranks by value descending / index ascending, q = p / noise[rank], last maximum) without its sort. Every
                // workgroup gathers the 2 048 logits (published by rowsum above), computes the soft-max statistics itself, RANKS ITS OWN 8 ROWS by counting the
                // probabilities ahead of each, and publishes its best (q, rank, index) candidate; the merge of the 256 candidates is the greedy path's.
                float * own_l = xf; float * nz = xf + 16; float * shf = xf + 272; int * cnt = (int *) (xf + 288); double * shd = (double *) (xf + 352);
                if ((tid & 15) == 0 && (tid >> 4) < 8) own_l[tid >> 4] = best;
                if (tid < 256) nz[tid] = smp_noise;
                CH_STAMP(11);
                for (int i = 0; i < N.delay[4]; i++) __builtin_amdgcn_s_sleep(1);
                float lg[4];
                {   // thread T holds logits T, T + 512, T + 1024, T + 1536: sample_topk_kernel<1024, 2>'s thread t holds t and t + 1024 - T and T + 512 of them
                    const unsigned tag_l = tag_base | (unsigned) (p + 1);
                    const unsigned lbase = (unsigned) (p & 1) * (CH_XF_MAX * 8u) + (unsigned) tid * 8u;
                    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                    u32x2_ g[4];
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int r = 0; r < 4; r++) g[r] = __builtin_amdgcn_raw_buffer_load_b64(gb, (int) (lbase + (unsigned) r * (512u * 8u)), 0, 16);
                        bool ok = true;
#pragma unroll
                        for (int r = 0; r < 4; r++) ok = ok && g[r].y == tag_l;
                        if (__all(ok)) break;
                        if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    settle_vmcnt();
#pragma unroll
                    for (int r = 0; r < 4; r++) lg[r] = __uint_as_float(g[r].x);
                }
                CH_STAMP(12);
                float px[4];
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; r++) { px[r] = lg[r] * smp_scale; mx = fmaxf(mx, px[r]); }
                mx = wave_allmax_f32(mx);
                if (lane == 0) shf[wave] = mx;
                nbar();
                mx = shf[0];
#pragma unroll
                for (int w = 1; w < CH_NCW; w++) mx = fmaxf(mx, shf[w]);
#pragma unroll
                for (int r = 0; r < 4; r++) px[r] = expf(px[r] - mx);
                // that kernel's double sum: a thread adds its two terms, a wave butterflies, the 16 waves are added in index order
                double sa = 0, sb = 0;
                sa += (double) px[0]; sa += (double) px[2];
                sb += (double) px[1]; sb += (double) px[3];
                sa = wave_allsum_f64(sa); sb = wave_allsum_f64(sb);
                if (lane == 0) { shd[wave] = sa; shd[CH_NCW + wave] = sb; }
                nbar();
                double tot = 0;
#pragma unroll
                for (int w = 0; w < 2 * CH_NCW; w++) tot += shd[w];
                const float inv = (float) (1.0 / tot);
                CH_STAMP(13);
#pragma unroll
                for (int r = 0; r < 4; r++) px[r] *= inv;
                // the workgroup's own 8 rows (rows 8 wg .. 8 wg + 7), probability of row j in lane j of every wave
                const float po = expf(own_l[lane & 7] * smp_scale - mx) * inv;
                // "ahead of row j" = larger probability, or the same one at a lower index: ONE unsigned compare of (bits of p) << 32 | ~index (p >= 0: the bit
                // patterns order like the values) - sample_topk_kernel's candidate key. (Written as p > pj || (p == pj && i < ij) the compiler branches per term
                // and spills the 32 ballots it keeps alive.)
                u64 kx[4];
#pragma unroll
                for (int r = 0; r < 4; r++) kx[r] = ((u64) __float_as_uint(px[r]) << 32) | (u64) ~(unsigned) (tid + r * 512);
                int cn[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const unsigned pjb = (unsigned) __builtin_amdgcn_readlane(__float_as_int(po), j);
                    const u64 kj = ((u64) pjb << 32) | (u64) ~(unsigned) (wg * 8 + j);
                    int n_ahead = 0;
#pragma unroll
                    for (int r = 0; r < 4; r++) n_ahead += __popcll(__ballot(kx[r] > kj));
                    asm volatile("" : "+s"(n_ahead));   // (counted HERE: left alone the compiler sinks all 32 pop-counts into the lane-0 store below and spills the masks)
                    cn[j] = n_ahead;
                }
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < 8; j++) cnt[j * CH_NCW + wave] = cn[j];
                }
                nbar();
                CH_STAMP(14);
                if (wave == 0) {
                    const int j = lane & 7;
                    int rank = 0;
#pragma unroll
                    for (int w = 0; w < CH_NCW; w++) rank += cnt[j * CH_NCW + w];
                    const float nzr = nz[rank < 256 ? rank : 255];
                    u64 key = 0;
                    if (rank < smp_k) key = ((u64) __float_as_uint(po / nzr) << 32) | (u64) ((unsigned) rank << 11) | (u64) (unsigned) (wg * 8 + j);
#pragma unroll
                    for (int o = 4; o > 0; o >>= 1) {
                        const u64 ok2 = ((u64) (unsigned) __shfl_xor((int) (key >> 32), o, 64) << 32) | (u64) (unsigned) __shfl_xor((int) (unsigned) key, o, 64);
                        key = ok2 > key ? ok2 : key;
                    }
                    if (lane == 0) {
                        u64 * c = P.cand + (size_t) (p & 1) * 2 * grid + 2 * wg;
                        st_granule(c, tag_base | (unsigned) (p + 1), (unsigned) (key >> 32));
                        st_granule(c + 1, tag_base | (unsigned) (p + 1), (unsigned) key);
                    }
                }
                CH_STAMP(15);
            } else if (N.head_argmax) {
                // (the 8 rows' values sit in the first lanes of 8 sixteen-lane groups: they meet through LDS - a wave-wide merge costs twelve ds_bpermute round trips)
                static_assert(nest_dim<shape_head, G>::ROWS == 8 && CH_NCW >= 8, "one candidate slot per row");
                if ((tid & 15) == 0 && (tid >> 4) < 8) { ctl->am_v[tid >> 4] = best; ctl->am_i[tid >> 4] = bi; }
                nbar();
                if (tid == 0) {
                    float cv[8]; int ci[8];   // (all fourteen LDS reads in front of the merge: inside it each one was waited for on its own)
#pragma unroll
                    for (int w = 1; w < 8; w++) { cv[w] = ctl->am_v[w]; ci[w] = ctl->am_i[w]; }
#pragma unroll
                    for (int w = 1; w < 8; w++) am_merge(best, bi, cv[w], ci[w]);
                    u64 * c = P.cand + (size_t) (p & 1) * 2 * grid + 2 * wg;
                    st_granule(c, tag_base | (unsigned) (p + 1), __float_as_uint(best));
                    st_granule(c + 1, tag_base | (unsigned) (p + 1), (unsigned) bi);
                }
            }
            CH_STAMP(9);
            p = __builtin_amdgcn_readfirstlane(p + 1);
        }
    }

    // the run ends in an arg-max: workgroup 0 merges the candidates and writes the token
    if (wg == 0 && wave == 0 && N.head_argmax) {
        const nest_st st = ld_st(N.n_steps - 1);
        int token = 0;
        const bool got = SMP ? gather_sampled(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token)
                                            : gather_token(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token);
        if (got) {
            if (lane == 0) { if (st.argmax_out[0]) *gp(st.argmax_out[0]) = token; if (st.argmax_out[1]) *gp(st.argmax_out[1]) = token; }
        } else give_up();
    }
    if (wg == 0 && tid == 0) *gp(P.launch_seq) = launch + 1u;
}
