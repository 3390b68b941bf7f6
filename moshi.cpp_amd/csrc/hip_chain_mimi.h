// hip_chain_mimi.h — part of hip_chain.hip (included there, same translation unit): a Mimi transformer (the codec's 8 pre-norm layers over T = 2 latent
// frames per call: /root/reference/src/moshi/modules/transformer.h:1345-1373 around :910-1039 with LayerNorm, GELU MLP 512 -> 2048 -> 512, layer scales, RoPE and
// a ring of 250; compression.h:149-204, 277-325) as ONE persistent launch instead of 40 - the third step program, for F32 weights.
//
// Per layer five phases: LN1 + in_proj (512 -> 1536), attention (16 owner workgroups, one per (head, query row), running the stand-alone launch's own
// device function - attn_ring256_body, hip_attn_body.h - on the in_proj phase's granules), out_proj * layer_scale + residual, LN2 + linear1 + GELU (512 -> 2048), linear2 *
// layer_scale + residual (2048 -> 512). Workgroup g owns rows [g M / 256, (g + 1) M / 256) of every matrix (6 / 2 / 8 / 2 rows), a wave one row, whose
// 2 KB (8 KB) of F32 weights it requests one phase ahead into registers; the two activation columns travel between phases as {tag, value} granules.
// Arithmetic: matvec_f32_reg_kernel's to the bit - the LayerNorm statistics in its 256-thread order (waves 0-3 emulate the launch's workgroup for column 0,
// waves 4-7 for column 1: per-thread double sums over t, t + 256, wave butterfly, four waves added in index order; mean, then variance), a row's dot by one
// wave (lane = 4 consecutive k per 256, products rounded to float and summed in double, wave butterfly), GELU through the F16 table, * scale, residual + x.
// Only planned on the context that owns the device's persistent launches (the serial frame loop: there the codec is on the critical path; in the two-stream
// loop the codec graphs run on the second stream beside the LM's own persistent launch and keep one launch per node group).
#pragma once

struct mimi_mv { const char * w; const float * alpha; const float * beta; const float * out_scale; float * y; float * x_out; float eps; int pad[3]; };   // 64 B
struct mimi_at { char * kcache; char * vcache; const float * rot; const float * mask; const int32_t * index; float * out; };                          // 48 B
static_assert(sizeof(mimi_mv) == 64 && sizeof(mimi_at) == 48, "tables are copied to LDS by 16-byte lanes");

struct mimi_params {
    chain_params P;
    const u32x4 * tables;      // device: mimi_mv[4 L] | mimi_at[L]
    int n_layers;
    const float * x_in; int64_t x_cs;   // the stack's input [512, 2] (column stride in floats)
    attn_args at;              // the attention's shape (pointers replaced per layer); q / k / v strides are those of the in_proj output in memory
    int q_off, k_off, v_off;   // of token 0 / head 0's q / k / v inside the in_proj output (floats)
    size_t attn_smem;
    int ring256;               // 1: the attention is attn_ring256_body (what the stand-alone launch runs at this shape), 0: attn_decode_body at 8 waves
    int first_partial;         // 1: the first layer's in_proj ran as a launch of its own in front of the program (the planner emits the RoPE table between it and
    const float * q0, * k0, * v0;   //    its attention): that layer starts at its attention phase, q / k / v read from memory at q0 / k0 / v0
    int delay[2];              // s_sleep units before the first poll of a mat-vec phase [0], of out_proj on workgroups that own no attention part [1]
};
#define MIMI_D 512
#define MIMI_F 2048

__global__ void __launch_bounds__(CH_THREADS) mimi_tr_kernel(mimi_params N) {
    constexpr int G = 256, D = MIMI_D, F = MIMI_F, T = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int L = N.n_layers;

    float * xs = (float *) smem;                                    // [2][2048] the phase's activation columns
    float * xres = xs + T * F;                                      // [2 cols][2 rows] this workgroup's rows of the residual stream
    double * stat = (double *) (xres + 8);                          // [2 slots][8 waves]
    chain_ctl * ctl = (chain_ctl *) (stat + 16);
    char * at_smem = (char *) (ctl + 1);
    const mimi_mv * t_mv = (const mimi_mv *) (at_smem + ((N.attn_smem + 15) & ~(size_t) 15));
    const mimi_at * t_at = (const mimi_at *) (t_mv + 4 * L);

    const chain_params & P = N.P;
    auto nbar = [&]() { lds_barrier(); };
    if (tid == 0) { ctl->failed = 0; ctl->token = 0; }
    {
        const int n16 = (4 * L * (int) sizeof(mimi_mv) + L * (int) sizeof(mimi_at)) / 16;
        const GLOBAL_AS u32x4 * src = (const GLOBAL_AS u32x4 *) N.tables;
        for (int i = tid; i < n16; i += CH_THREADS) ((u32x4 *) t_mv)[i] = src[i];
    }
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    __syncthreads();

    auto ld_mv = [&](int q) {
        mimi_mv r;
        unsigned w[16];
        const unsigned * s = (const unsigned *) (t_mv + q);
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) s[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };
    auto ld_w = [&](int q) -> const char * {
        const unsigned * s = (const unsigned *) (t_mv + q);
        unsigned w[2] = { (unsigned) __builtin_amdgcn_readfirstlane((int) s[0]), (unsigned) __builtin_amdgcn_readfirstlane((int) s[1]) };
        const char * r;
        __builtin_memcpy(&r, w, 8);
        return r;
    };
    auto ld_at = [&](int q) {
        mimi_at r;
        unsigned w[12];
        const unsigned * s = (const unsigned *) (t_at + q);
#pragma unroll
        for (int i = 0; i < 12; i++) w[i] = (unsigned) __builtin_amdgcn_readfirstlane((int) s[i]);
        __builtin_memcpy(&r, w, sizeof(r));
        return r;
    };
    auto give_up = [&]() { if (lane == 0) { lds_store(&ctl->failed, 1u); *gp(P.err) = 2u; } };

    // ---- a wave's row of the NEXT phase in registers: KV x 16 bytes per lane (k = 256 it + 4 lane .. + 3), requested one phase ahead
    f32x4 wr[8];
    auto request = [&](auto kvc, const char * w, int rows) {   // rows: of this workgroup in that phase (a wave per row)
        constexpr int KV = decltype(kvc)::value;
        const int row = wg * rows + (wave < rows ? wave : rows - 1);
        const GLOBAL_AS char * src = gp(w) + (long long) row * (KV * 256 * 4);
#pragma unroll
        for (int it = 0; it < KV; it++) wr[it] = __builtin_nontemporal_load((const GLOBAL_AS f32x4 *) (src + ((long long) it * 256 + lane * 4) * 4));
    };
    if (N.first_partial) request(std::integral_constant<int, 2>(), ld_w(1), 2); else request(std::integral_constant<int, 2>(), ld_w(0), 6);

    // ---- the phase's input columns into registers: thread (c = wave / 4, t = 64 (wave % 4) + lane) holds x[c][t + 256 j], the layout of matvec_f32_reg_kernel's
    // 256-thread workgroup. From the previous phase's granules (value (c, k) at index c * K + k), or from memory (layer 0's in_proj).
    const int col = wave >> 2, t256 = (wave & 3) * 64 + lane;
    auto poll_cols = [&](auto kvc, int p, float xr[], int delay) {
        constexpr int KV = decltype(kvc)::value;
        for (int i = 0; i < delay; i++) __builtin_amdgcn_s_sleep(1);
        const unsigned tag_in = tag_base | (unsigned) p;
        const u64 * gin = P.gbuf + (size_t) ((p - 1) & 1) * CH_XF_MAX + (size_t) col * (KV * 256) + t256;
        u64 g[KV];
        unsigned spins = 0;
        for (;;) {
#pragma unroll
            for (int j = 0; j < KV; j++) g[j] = __hip_atomic_load(gin + 256 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int j = 0; j < KV; j++) ok = ok && (unsigned) (g[j] >> 32) == tag_in;
            if (__all(ok)) break;
            if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        settle_vmcnt();
#pragma unroll
        for (int j = 0; j < KV; j++) xr[j] = __uint_as_float((unsigned) g[j]);
    };
    // LayerNorm in the launch's order (two passes; per-thread double sums in j order, wave butterfly, the column's four waves in index order)
    auto col_sum = [&](double v, int slot) -> double {
        v = wave_allsum_f64(v);
        if (lane == 0) stat[slot * 8 + wave] = v;
        nbar();
        const double * s4 = stat + slot * 8 + col * 4;
        return s4[0] + s4[1] + s4[2] + s4[3];
    };
    auto layer_norm = [&](float xr[2], const mimi_mv & mv) {
        const float al[2] = { gp(mv.alpha)[t256], gp(mv.alpha)[t256 + 256] };
        float be[2] = { 0.f, 0.f };
        if (mv.beta) { be[0] = gp(mv.beta)[t256]; be[1] = gp(mv.beta)[t256 + 256]; }
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) acc += (double) xr[j];
        acc = col_sum(acc, 0);
        const float mean = (float) (acc / (double) D);
        acc = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) { const float v = xr[j] - mean; acc += (double) (v * v); }
        acc = col_sum(acc, 1);
        const float scale = 1.0f / sqrtf((float) (acc / (double) D) + mv.eps);
#pragma unroll
        for (int j = 0; j < 2; j++) { float v = ((xr[j] - mean) * scale) * al[j]; if (mv.beta) v = v + be[j]; xr[j] = v; }
    };
    // one row per wave against both columns in xs; lane 0 returns the two sums
    auto row_dot = [&](auto kvc, float out[2]) {
        constexpr int KV = decltype(kvc)::value, K = KV * 256;
        double acc[2] = { 0, 0 };
#pragma unroll
        for (int it = 0; it < KV; it++) {
            const int k = it * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float4 xv = *(const float4 *) (xs + c * K + k);
                acc[c] += (double) (wr[it][0] * xv.x);
                acc[c] += (double) (wr[it][1] * xv.y);
                acc[c] += (double) (wr[it][2] * xv.z);
                acc[c] += (double) (wr[it][3] * xv.w);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) out[c] = (float) wave_allsum_f64(acc[c]);
    };
    auto publish = [&](int p, int M, int row, const float v[2]) {
        const unsigned tag_out = tag_base | (unsigned) (p + 1);
        u64 * go = P.gbuf + (size_t) (p & 1) * CH_XF_MAX;
#pragma unroll
        for (int c = 0; c < 2; c++) st_granule(go + (size_t) c * M + row, tag_out, __float_as_uint(v[c]));
    };
    [[maybe_unused]] constexpr int grid = G;   // (CH_STAMP)
    const int owner_idx = wg >> 4;
    const bool owner = (wg & 15) == (owner_idx & 7);   // workgroup 16 i + (i & 7) runs attention part i = 2 h + t

    int p = 0;
#pragma unroll 1
    for (int l = 0; l < L; l++) {
        // -------------------------------------------------------------------- LN1 + in_proj 512 -> 1536
        if (l == 0 && N.first_partial) {
            if (wave < 2) {
#pragma unroll
                for (int c = 0; c < 2; c++) if (lane == c) xres[c * 2 + wave] = gp(N.x_in)[(int64_t) c * N.x_cs + wg * 2 + wave];
            }
            p = __builtin_amdgcn_readfirstlane(p + 1);
        } else {
            CH_STAMP(10);
            const mimi_mv mv = ld_mv(4 * l);
            float xr[2];
            if (l == 0) { xr[0] = gp(N.x_in)[(int64_t) col * N.x_cs + t256]; xr[1] = gp(N.x_in)[(int64_t) col * N.x_cs + t256 + 256]; }
            else poll_cols(std::integral_constant<int, 2>(), p, xr, N.delay[0]);
            if (l == 0 && wave < 2) {   // rows 2 wg, 2 wg + 1 of the stack's input: this workgroup's part of the residual stream
#pragma unroll
                for (int c = 0; c < 2; c++) if (lane == c) xres[c * 2 + wave] = gp(N.x_in)[(int64_t) c * N.x_cs + wg * 2 + wave];
            }
            CH_STAMP(2);
            layer_norm(xr, mv);
            xs[col * D + t256] = xr[0]; xs[col * D + t256 + 256] = xr[1];
            if (mv.x_out && wg == 0) { gp(mv.x_out)[(int64_t) col * D + t256] = xr[0]; gp(mv.x_out)[(int64_t) col * D + t256 + 256] = xr[1]; }
            nbar();
            CH_STAMP(5);
            if (wave < 6) {
                float o[2];
                row_dot(std::integral_constant<int, 2>(), o);
                const int row = wg * 6 + wave;
                if (lane == 0) { publish(p, 3 * D, row, o); gp(mv.y)[row] = o[0]; gp(mv.y)[3 * D + row] = o[1]; }
            }
            request(std::integral_constant<int, 2>(), ld_w(4 * l + 1), 2);   // out_proj
            CH_STAMP(8);
            p = __builtin_amdgcn_readfirstlane(p + 1);
        }
        // -------------------------------------------------------------------- attention: part (h, t) on its owner workgroup, everybody else moves on
        {
            CH_STAMP(10);
            if (owner) {
                const mimi_at ta = ld_at(l);
                attn_args at = N.at;
                at.kcache = ta.kcache; at.vcache = ta.vcache; at.rot = ta.rot; at.mask = ta.mask; at.index = ta.index; at.out = ta.out;
                at.q = nullptr; at.k = nullptr; at.v = nullptr;   // (element offsets come through gq; the strides are those of the in_proj output)
                at.row_split = 1; at.n_groups = 0; at.write_only = 0;
                const attn_split_ws w0 = { nullptr, nullptr, nullptr, 1, P.err, 0, 1 << 30, 1 << 30 };
                const attn_gqkv gq = { P.gbuf + (size_t) ((p - 1) & 1) * CH_XF_MAX, (int64_t) N.q_off, (int64_t) N.k_off, (int64_t) N.v_off, P.err };
                attn_gout go = { P.gbuf + (size_t) (p & 1) * CH_XF_MAX, tag_base | (unsigned) (p + 1), (int64_t) D };
#if defined(CH_LOG)
                go.log = wg == 0 && p < 512 ? &g_ch_log[0][p][0] : nullptr;
#endif
                __syncthreads();
                if (l == 0 && N.first_partial) {
                    at.q = N.q0; at.k = N.k0; at.v = N.v0;
                    if (N.ring256) attn_ring256_body<AT_GOUT>(at, at_smem, owner_idx >> 1, owner_idx & 1, 0u, attn_gqkv(), go);
                    else attn_decode_body<false, CH_NCW, AT_GOUT>(at, w0, at_smem, owner_idx >> 1, 0, owner_idx & 1, 0u, attn_gqkv(), go);
                } else if (N.ring256) attn_ring256_body<AT_GQKV | AT_GOUT>(at, at_smem, owner_idx >> 1, owner_idx & 1, tag_base | (unsigned) p, gq, go);
                else attn_decode_body<false, CH_NCW, AT_GQKV | AT_GOUT>(at, w0, at_smem, owner_idx >> 1, 0, owner_idx & 1, tag_base | (unsigned) p, gq, go);
                __syncthreads();
                CH_STAMP(8);
            }
            p = __builtin_amdgcn_readfirstlane(p + 1);
        }
        // -------------------------------------------------------------------- out_proj 512 -> 512, * layer_scale, + residual
        {
            CH_STAMP(10);
            const mimi_mv mv = ld_mv(4 * l + 1);
            float osc = 1.f;   // (the row's layer scale is asked for before the poll: read in the epilogue it is a dependent round trip on lane 0's critical path)
            if (mv.out_scale && wave < 2) osc = gp(mv.out_scale)[wg * 2 + wave];
            float xr[2];
            poll_cols(std::integral_constant<int, 2>(), p, xr, owner ? 0 : N.delay[1]);
            CH_STAMP(2);
            xs[col * D + t256] = xr[0]; xs[col * D + t256 + 256] = xr[1];
            nbar();
            CH_STAMP(5);
            if (wave < 2) {
                float o[2];
                row_dot(std::integral_constant<int, 2>(), o);
                const int row = wg * 2 + wave;
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 2; c++) { float s = o[c]; if (mv.out_scale) s = s * osc; s = xres[c * 2 + wave] + s; o[c] = s; xres[c * 2 + wave] = s; }
                    publish(p, D, row, o); gp(mv.y)[row] = o[0]; gp(mv.y)[D + row] = o[1];
                }
            }
            request(std::integral_constant<int, 2>(), ld_w(4 * l + 2), 8);   // linear1
            CH_STAMP(8);
            p = __builtin_amdgcn_readfirstlane(p + 1);
        }
        // -------------------------------------------------------------------- LN2 + linear1 512 -> 2048 + GELU
        {
            CH_STAMP(10);
            const mimi_mv mv = ld_mv(4 * l + 2);
            float xr[2];
            poll_cols(std::integral_constant<int, 2>(), p, xr, N.delay[0]);
            CH_STAMP(2);
            layer_norm(xr, mv);
            xs[col * D + t256] = xr[0]; xs[col * D + t256 + 256] = xr[1];
            if (mv.x_out && wg == 0) { gp(mv.x_out)[(int64_t) col * D + t256] = xr[0]; gp(mv.x_out)[(int64_t) col * D + t256 + 256] = xr[1]; }
            nbar();
            {
                float o[2];
                row_dot(std::integral_constant<int, 2>(), o);
                const int row = wg * 8 + wave;
                if (lane == 0) { o[0] = gelu_table(o[0]); o[1] = gelu_table(o[1]); publish(p, F, row, o); gp(mv.y)[row] = o[0]; gp(mv.y)[F + row] = o[1]; }
            }
            request(std::integral_constant<int, 8>(), ld_w(4 * l + 3), 2);   // linear2
            CH_STAMP(8);
            p = __builtin_amdgcn_readfirstlane(p + 1);
        }
        // -------------------------------------------------------------------- linear2 2048 -> 512, * layer_scale, + residual
        {
            CH_STAMP(10);
            const mimi_mv mv = ld_mv(4 * l + 3);
            float osc = 1.f;
            if (mv.out_scale && wave < 2) osc = gp(mv.out_scale)[wg * 2 + wave];
            float xr[8];
            poll_cols(std::integral_constant<int, 8>(), p, xr, N.delay[0]);
            CH_STAMP(2);
#pragma unroll
            for (int j = 0; j < 8; j++) xs[col * F + t256 + 256 * j] = xr[j];
            nbar();
            if (wave < 2) {
                float o[2];
                row_dot(std::integral_constant<int, 8>(), o);
                const int row = wg * 2 + wave;
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 2; c++) { float s = o[c]; if (mv.out_scale) s = s * osc; s = xres[c * 2 + wave] + s; o[c] = s; xres[c * 2 + wave] = s; }
                    if (l + 1 < L) publish(p, D, row, o);
                    gp(mv.y)[row] = o[0]; gp(mv.y)[D + row] = o[1];
                }
            }
            if (l + 1 < L) request(std::integral_constant<int, 2>(), ld_w(4 * l + 4), 6);   // the next layer's in_proj
            nbar();   // (xs is rewritten by the next phase's input stage)
            CH_STAMP(8);
            p = __builtin_amdgcn_readfirstlane(p + 1);
        }
    }
    if (wg == 0 && tid == 0) *gp(P.launch_seq) = launch + 1u;
}
