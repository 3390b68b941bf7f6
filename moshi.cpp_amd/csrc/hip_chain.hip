// hip_chain.hip — the persistent chain engine: a run of dependent small Q4_K mat-vecs (the chained Depth transformer of moshi.cpp,
// /root/reference/src/moshi/models/lm.h:446-553: depformer_in -> 6 x {in_proj, attention + out_proj, linear_in, linear_out} -> linears[k] ->
// arg-max -> next step's embedding, 26 mat-vecs per step, 8 / 16 / 32 steps per frame) executed by ONE launch instead of one launch per mat-vec.
//
// Why: each of those mat-vecs moves 0.6 - 3.2 MB. As separate launches they cost 2.3 - 4.9 us in the kernel plus 1.4 - 1.8 us of boundary
// (profiles/r02_frame_stamps_in_graph.txt) - 137 us per step for 47 MB, 0.04 of the HBM roofline. What a step needs is (a) its weights streamed
// once and (b) 26 all-to-all hand-offs of a <= 12 KB vector. Here:
//   * G workgroups of 8 waves stay resident for the whole run; workgroup g owns the same contiguous block of rows of every matrix;
//   * the weights of phase p + 1 are requested - straight into registers, 8 lanes per super-block, nontemporal - as soon as a wave has finished
//     the dot products of phase p, i.e. BEFORE the phase's results are handed on: they are in flight across the hand-off, which is what a kernel
//     boundary can never do. (A first form streamed them through an LDS ring with a dedicated LDS-DMA loader wave; one fill costs that wave
//     50 - 150 cycles of issue, tests/microbench/lds_dma_rate.hip, and through the shared barriers every workgroup ended up waiting for its loader:
//     gpurun_out/r3_sweep*.log.)
//   * every wave gathers the previous phase's vector itself, runs the prologue (RMS norm / attention over the 8-slot ring / plain), quantises to
//     Q8_K exactly as matvec_q4k_kernel does, dots its super-blocks (the WS = 1 arithmetic), reduces rows in the same fixed order, publishes its rows;
//   * the hand-off is data-tagged: every published float travels as one 8-byte {tag, value} granule written by one agent-scope (sc1) store and
//     polled with agent-scope loads - the data IS the flag, there is no grid barrier, no fence and no separate counter on the critical path
//     (cdna_hip_programming.md Guideline 16, form R2). tag = launch sequence number << 12 | phase index + 1: never 0, never repeated in a slot;
//   * the greedy sample (ggml's last-maximum arg-max) travels as one (value, index) candidate per workgroup; every workgroup merges them itself.
// Arithmetic is the unchained kernels' to the bit: same Q8_K rounding, same per-super-block float expression, same 16-lane row sums, same
// attention (attn_small_wave) - tests/test_chain_engine.py compares the two plans bit for bit.
#include "hip_common.h"
#include "hip_device.h"
#include "hip_mv_device.h"
#include "hip_attn_body.h"

#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <map>
#include <vector>

#define CH_NCW       8                    // consumer waves per workgroup
#define CH_THREADS   (CH_NCW * 64)
// passes of 64 super-blocks a workgroup holds in registers per phase (rows * nb <= passes * 64): fewer rows per workgroup on a larger grid
#define CH_PMAX_OF(G) ((G) >= 256 ? 2 : (G) >= 128 ? 3 : 6)
#define CH_XF_MAX    4096                 // floats: longest vector handed from one phase to the next
#define CH_PART_MAX  2048                 // super-block partial sums per workgroup and phase
#define CH_RES_MAX   256                  // rows per workgroup and phase
#define CH_ATTW      1408                 // floats of attention scratch per consumer wave: 2 heads x (q | k | v) + 2 x 8 slots x 64 products
#define CH_SPIN_MAX  (1u << 24)
#define CH_MAX_PHASES 4095                // tag = launch << 12 | phase + 1

typedef unsigned long long u64;

// diagnostic build only (-DCH_LOG, tests/microbench/chain_stamps.py): wave 0 of the first and the last workgroup log s_memrealtime (10 ns ticks)
// at the stages of every phase of the last launch
#if defined(CH_LOG)
__device__ u64 g_ch_log[2][512][16];   // [0] workgroup 0 wave 0, [1] last workgroup wave 0
#define CH_STAMP(i) do { if (wave == 0 && lane == 0 && (wg == 0 || wg == grid - 1) && p < 512) { u64 t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_ch_log[wg == 0 ? 0 : 1][p][i] = t_; } } while (0)
// (inside chain_attn_wave: stamps 11 .. 15 by wave 0 of the logging workgroups)
#define CH_ASTAMP(i) do { if (log_wg >= 0 && h0 == 0 && lane == 0 && log_p < 512) { u64 t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_ch_log[log_wg][log_p][i] = t_; } } while (0)
extern "C" __attribute__((visibility("default"))) int mi355x_chain_log_read(u64 * out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ch_log), sizeof(g_ch_log)) == hipSuccess ? 0 : -1;
}
#else
#define CH_STAMP(i) do {} while (0)
#define CH_ASTAMP(i) do {} while (0)
#endif

struct chain_phase {
    const char * w; long long row_bytes;
    int K, M, nb, rows_wg;     // rows_wg: rows per workgroup (paired: rows of EACH half)
    long long pair_F;          // > 0: W is linear_in [K, 2 F] of a gated FFN; a workgroup takes the same rows of both halves and hands on silu(l) * r
    int prologue;              // MV_PLAIN / MV_RMSNORM / MV_ATTN
    int x_chain;               // 1: x is the vector the previous phase published; 0: x is read from memory (produced before this launch)
    const float * x; const float * alpha; float eps;
    int attn;                  // MV_ATTN: index into chain_params::attns
    int q_off, k_off, v_off;   //          offsets of head 0's q / k / v inside the previous phase's vector
    int res;                   // residual: 0 none, 1 the rows this workgroup kept from an earlier phase, 2 memory
    const float * residual;
    embed_src emb; int emb_chain;   // one embedding row added in the epilogue; emb_chain: its index is the previous phase's arg-max
    int save;                  // keep this phase's rows in the workgroup for a later residual
    int argmax;                // reduce the rows to arg-max candidates (ggml_vec_argmax_f32: the LAST maximum)
    int32_t * argmax_out[2];
    float * y;                 // the ggml node's own storage (always written, plain stores: nothing inside the launch reads it)
    int n_pub;                 // values handed to the next phase: 0, M, or pair_F
    int n_in;                  // values the previous phase handed on
    int kind;                  // 0: shape read from this descriptor; 1..6: one of the compile-time shapes (shape_din .. shape_head)
    int fmt;                   // weight blocks: 0 Q4_K super-blocks (144 B / 256 weights, Q8_K activations), 1 Q8_0 (eight 34-byte blocks = 272 B / 256 weights, Q8_0 activations)
    int pad_[3];
    int32_t * prev_out[2];     // emb_chain: where the previous phase's merged arg-max (the token) is stored
    attn_args at;              // MV_ATTN: the attention whose output is x
};
static_assert(sizeof(chain_phase) % 16 == 0 && sizeof(chain_phase) <= 27 * 16, "a descriptor is staged through LDS by 16-byte lanes");
#define CH_DESC_DWORDS ((int) (sizeof(chain_phase) / 4))

struct chain_params {
    const chain_phase * phases; int n_phases;
    u64 * gbuf;                // [2][CH_XF_MAX] granules, by phase parity
    u64 * cand;                // [2][2 * grid] arg-max candidates {tag, value bits}, {tag, index}
    unsigned * launch_seq;     // bumped by workgroup 0 at the end of every launch
    unsigned * err;            // host-visible error word (bounded waits)
    int delay;                 // s_sleep units between a phase's start and its first look at a register hand-off (MI355X_CHAIN_DELAY): a poll that samples before
                               // the slowest producer's stores are visible comes back empty and costs a whole fabric round trip under everybody's polling
};

// ---- small helpers ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_load(unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(unsigned * p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The phase table and the attention descriptors are written by the host before the launch and never during it: read them through the constant
// address space (scalar loads, no vector-memory wait in front of every phase). Pointers taken out of a descriptor lose their address space, so
// they are cast back to global explicitly (a flat access would tie up both memory counters).
typedef const __attribute__((address_space(4))) struct chain_phase * cphase_ptr;
typedef const __attribute__((address_space(4))) attn_args * cattn_ptr;
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ GLOBAL_AS T * gp(T * p) { return (GLOBAL_AS T *) p; }
template <typename T> __device__ __forceinline__ const GLOBAL_AS T * gp(const T * p) { return (const GLOBAL_AS T *) p; }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// s_waitcnt vmcnt(0) as an instruction the compiler's wait-count bookkeeping SEES (the asm form above is opaque to it). Placed where a polling loop leaves:
// the loop's loads are the youngest of the wave, so nothing younger is held up, and no path remains on which one of them counts as still in flight - a path
// like that (a load issued and a use skipped under the same wave-uniform condition, or the give-up exit) makes the compiler open the NEXT phase with
// vmcnt(0), i.e. behind the weights requested in between.
__device__ __forceinline__ void settle_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0F70); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void * p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int) bytes, 0x00020000);
}
// agent-scope (sc1) 16-byte load / 8-byte store / 2-byte store: past this CU's L1, served at the device's coherence point
__device__ __forceinline__ u32x4 ld16_agent(__amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int) byte_off, 0, 16); }
__device__ __forceinline__ void st_granule(u64 * p, unsigned tag, unsigned value) {
    __hip_atomic_store(p, ((u64) tag << 32) | (u64) value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct chain_ctl {            // LDS control block
    unsigned failed;          // a bounded wait gave up: everybody leaves (checked behind every workgroup barrier)
    int token;                // merged arg-max of the previous phase
    unsigned pad[2];
    double sumsq[CH_NCW];
    float am_v[CH_NCW]; int am_i[CH_NCW];
};

// Gather `n` (even) tagged values into dst[0..n) (LDS). Every consumer wave sweeps its own 128-value slices with 16-byte agent-scope loads
// (two granules each; all of a sweep's loads are issued before the first tag is looked at) until all its tags match; values go to LDS as they
// are. Returns false when the wait gave up.
template <int NIT>
__device__ __forceinline__ bool gather_vector_n(__amdgpu_buffer_rsrc_t gb, unsigned base_bytes, int npairs, unsigned tag, float * dst, int wave, int lane, chain_ctl * ctl) {
    unsigned spins = 0;
    for (;;) {
        u32x4 g[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int pi = (it * CH_NCW + wave) * 64 + lane;
            g[it] = ld16_agent(gb, base_bytes + (unsigned) (pi < npairs ? pi : npairs - 1) * 16u);
        }
        bool ok = true;
#pragma unroll
        for (int it = 0; it < NIT; it++) ok = ok && g[it].y == tag && g[it].w == tag;
        if (__all(ok)) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int pi = (it * CH_NCW + wave) * 64 + lane;
                if (pi < npairs) *(float2 *) (dst + 2 * pi) = make_float2(__uint_as_float(g[it].x), __uint_as_float(g[it].z));
            }
            return true;
        }
        if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { settle_vmcnt(); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ bool gather_vector(__amdgpu_buffer_rsrc_t gb, unsigned base_bytes, int n, unsigned tag, float * dst, int wave, int lane, chain_ctl * ctl) {
    const int npairs = n >> 1, nit = (npairs + CH_NCW * 64 - 1) / (CH_NCW * 64);
    static_assert(CH_XF_MAX <= 4 * CH_NCW * 128, "gather_vector: four sweeps per wave cover the longest hand-off");
    switch (nit) {
        case 1:  return gather_vector_n<1>(gb, base_bytes, npairs, tag, dst, wave, lane, ctl);
        case 2:  return gather_vector_n<2>(gb, base_bytes, npairs, tag, dst, wave, lane, ctl);
        case 3:  return gather_vector_n<3>(gb, base_bytes, npairs, tag, dst, wave, lane, ctl);
        default: return gather_vector_n<4>(gb, base_bytes, npairs, tag, dst, wave, lane, ctl);
    }
}

// wave_allmax_f32 / wave_allsum_f64 (hip_device.h) for a value that is uniform inside every group of 8 lanes (for the sum: carried by ONE lane of the group,
// zero in the others): their first three steps stay inside the group and leave such a value as it is
__device__ __forceinline__ float wave_allmax_f32_groups8(float v) {
    v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ double wave_allsum_f64_groups8(double v) {
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// Attention of 2 consecutive heads by one wave for a single new token over a ring of <= 8 slots of 64: attn_small_wave<2> of hip_kernels_fused.hip
// with the new token's q / k / v taken from LDS (the in_proj rows just gathered), the ring rows passed in (requested with agent-scope loads BEFORE
// the hand-off wait: they were written by earlier phases) and the ring write done with agent-scope stores. Same arithmetic in the same order.
__device__ __forceinline__ void chain_attn_ring_loads(const attn_args & a, int h0, int lane, u32x4 kq[2], u32x4 vq[2], __amdgpu_buffer_rsrc_t kr, __amdgpu_buffer_rsrc_t vr) {
    const int sub = lane / 8, dl = (lane % 8) * 8;
    const int cc = sub < a.C ? sub : a.C - 1;
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        const int h = h0 + hh;
        kq[hh] = ld16_agent(kr, (unsigned) ((int64_t) h * a.k_nb2 + (int64_t) cc * a.k_nb1 + dl * 2));
        vq[hh] = ld16_agent(vr, (unsigned) ((int64_t) h * a.v_nb2 + (int64_t) cc * a.v_nb1 + dl * 2));
    }
}
__device__ __forceinline__ void chain_attn_wave(const attn_args & a, const float * qb, const float * kb, const float * vb, int h0, int lane, float * wbuf, float * xa,
                                                const u32x4 kq[2], const u32x4 vq[2], bool write_cache, __amdgpu_buffer_rsrc_t kr, __amdgpu_buffer_rsrc_t vr,
                                                int slot, float m, float rc, float rs, int log_wg = -1, int log_p = 0) {
    constexpr int NH = 2, LPS = 8, half = 32;
    CH_ASTAMP(11);
    const int C = a.C;
    const int sub = lane / LPS, dl = (lane % LPS) * 8;
    const int c = sub;
    const int j = lane, p = j < half ? j : j - half;
    float qr[NH], qi[NH], kr_[NH], ki[NH], vv[NH];
#pragma unroll
    for (int hh = 0; hh < NH; hh++) {
        const int h = h0 + hh;
        const float * q = qb + (int64_t) h * a.q_hs, * k = kb + (int64_t) h * a.k_hs, * v = vb + (int64_t) h * a.v_hs;
        if (a.rot) { qr[hh] = q[2 * p]; qi[hh] = q[2 * p + 1]; kr_[hh] = k[2 * p]; ki[hh] = k[2 * p + 1]; }
        else { qr[hh] = q[j]; qi[hh] = 0.f; kr_[hh] = k[j]; ki[hh] = 0.f; }
        vv[hh] = v[j];
    }
    const bool live = c < C && m > -INFINITY;
    const bool fresh = slot == c;
#pragma unroll
    for (int hh = 0; hh < NH; hh++) {
        float qo, ko;
        if (a.rot) {
            if (j < half) { qo = qr[hh] * rc - qi[hh] * rs; ko = kr_[hh] * rc - ki[hh] * rs; }
            else          { qo = qr[hh] * rs + qi[hh] * rc; ko = kr_[hh] * rs + ki[hh] * rc; }
        } else { qo = qr[hh]; ko = kr_[hh]; }
        const uint16_t kbv = f2bf(ko), vbv = f2bf(vv[hh]);
        float * wb = wbuf + hh * 192;
        wb[j] = bf2f(f2bf(qo)); wb[64 + j] = bf2f(kbv); wb[128 + j] = bf2f(vbv);
        if (write_cache && slot >= 0 && slot < C) {
            const int h = h0 + hh;
            __builtin_amdgcn_raw_buffer_store_b16((short) kbv, kr, (int) ((int64_t) h * a.k_nb2 + (int64_t) slot * a.k_nb1 + j * 2), 0, 16);
            __builtin_amdgcn_raw_buffer_store_b16((short) vbv, vr, (int) ((int64_t) h * a.v_nb2 + (int64_t) slot * a.v_nb1 + j * 2), 0, 16);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float * prod = wbuf + NH * 192;
    CH_ASTAMP(12);
    // The two heads' chains - 8-dim partial dot, 8-lane sum, wave maximum, exponential, wave sum, reciprocal, P x V products - are independent and each is a
    // string of dependent DPP reductions: written stage by stage over BOTH heads, without branches and with every LDS read in front, so that the scheduler
    // interleaves them (one after the other they took 0.6 us each, profiles/r04_chain_stamps.txt). Per head the arithmetic and its order are unchanged.
    float qv[2][8], kv[2][8], vv8[2][8];
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        const float * qf = wbuf + hh * 192, * knew = qf + 64, * vnew = qf + 128;
        const uint32_t kw[4] = { kq[hh].x, kq[hh].y, kq[hh].z, kq[hh].w }, vw[4] = { vq[hh].x, vq[hh].y, vq[hh].z, vq[hh].w };
#pragma unroll
        for (int i = 0; i < 8; i++) {
            qv[hh][i] = qf[dl + i];
            const float kn = knew[dl + i], vn = vnew[dl + i];
            const float kr8 = bf2f((uint16_t) ((i & 1) ? (kw[i >> 1] >> 16) : (kw[i >> 1] & 0xffff))), vr8 = bf2f((uint16_t) ((i & 1) ? (vw[i >> 1] >> 16) : (vw[i >> 1] & 0xffff)));
            kv[hh][i] = fresh ? kn : kr8;
            vv8[hh][i] = fresh ? vn : vr8;
        }
    }
    double acc[2];
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        double a2 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) a2 += (double) (kv[hh][i] * qv[hh][i]);
        acc[hh] = live ? a2 : 0.0;
    }
#pragma unroll
    for (int hh = 0; hh < 2; hh++) acc[hh] = group_allsum_f64(acc[hh], LPS);
    float sv[2], gmax[2], e[2];
#pragma unroll
    for (int hh = 0; hh < 2; hh++) sv[hh] = live ? (float) acc[hh] * a.scale + m : -INFINITY;
    // (sv and e are the same in all 8 lanes of a slot group: the three in-group steps of wave_allmax_f32 / wave_allsum_f64 would return their input -
    // max(v, v), and e + 0 + 0 + 0 with only lane 0 of a group carrying e - so the reductions start at the step that crosses groups; same values, same order)
#pragma unroll
    for (int hh = 0; hh < 2; hh++) gmax[hh] = wave_allmax_f32_groups8(sv[hh]);
#pragma unroll
    for (int hh = 0; hh < 2; hh++) e[hh] = sv[hh] > -INFINITY ? expf(sv[hh] - gmax[hh]) : 0.f;
    double lsum[2];
#pragma unroll
    for (int hh = 0; hh < 2; hh++) lsum[hh] = wave_allsum_f64_groups8((double) e[hh]);
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        const float inv = (float) (1.0 / lsum[hh]);
        const float pr = bf2f(f2bf(e[hh] * inv));
        float pf[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pf[i] = pr != 0.f ? vv8[hh][i] * pr : 0.f;
        float * dst = prod + hh * 512 + sub * 64 + dl;
        *(float4 *) dst = make_float4(pf[0], pf[1], pf[2], pf[3]);
        *(float4 *) (dst + 4) = make_float4(pf[4], pf[5], pf[6], pf[7]);
    }
    CH_ASTAMP(13);
    CH_ASTAMP(14);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int hq = 0; hq < 2; hq++) {
        double tot = 0;
#pragma unroll
        for (int c2 = 0; c2 < 8; c2++) tot += (double) prod[hq * 512 + c2 * 64 + lane];
        xa[hq * 64 + lane] = (float) tot;
    }
    CH_ASTAMP(15);
}

// one field of a descriptor parked in LDS, as a wave-uniform (scalar) value
template <typename T> __device__ __forceinline__ T desc_field(const unsigned * base, int dw) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "descriptor fields are 4 or 8 bytes");
    unsigned w[2] = { (unsigned) __builtin_amdgcn_readfirstlane(base[dw]), 0u };
    if (sizeof(T) == 8) w[1] = (unsigned) __builtin_amdgcn_readfirstlane(base[dw + 1]);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

// merge of arg-max candidates: larger value wins, equal values -> larger index (= the last maximum of the whole row)
__device__ __forceinline__ void am_merge(float & best, int & bi, float ov, int oi) { if (ov > best || (ov == best && oi > bi)) { best = ov; bi = oi; } }
__device__ __forceinline__ void am_wave(float & best, int & bi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64); am_merge(best, bi, ov, oi); }
}
// one wave reads every workgroup's candidate of `tag` and merges them; false when the wait gave up. (All of a sweep's loads are issued before the first
// tag is looked at: written as load - compare - load the compiler serialises the four fabric round trips of a 256-workgroup grid.)
__device__ __forceinline__ bool gather_token(__amdgpu_buffer_rsrc_t cb, unsigned base_bytes, int grid, unsigned tag, int lane, chain_ctl * ctl, int & token) {
    unsigned spins = 0;
    for (;;) {
        u32x4 c[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int g = i * 64 + lane;
            c[i] = ld16_agent(cb, base_bytes + (unsigned) (g < grid ? g : grid - 1) * 16u);
        }
        float best = -INFINITY; int bi = -1;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int g = i * 64 + lane;
            ok = ok && c[i].y == tag && c[i].w == tag;
            if (g < grid) am_merge(best, bi, __uint_as_float(c[i].x), (int) c[i].z);
        }
        if (__all(ok)) { am_wave(best, bi); token = bi < 0 ? 0 : bi; return true; }
        if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { settle_vmcnt(); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}
static_assert(CH_NCW * 32 >= 256, "gather_token: four sweeps of a wave cover the largest grid");

// one attention descriptor out of the constant-address-space table, field by field (scalar loads)
__device__ __forceinline__ attn_args load_attn(cattn_ptr p) {
    attn_args a;
    a.q = p->q; a.k = p->k; a.v = p->v;
    a.q_ts = p->q_ts; a.q_hs = p->q_hs; a.k_ts = p->k_ts; a.k_hs = p->k_hs; a.v_ts = p->v_ts; a.v_hs = p->v_hs;
    a.rot = p->rot; a.mask = p->mask; a.index = p->index; a.kcache = p->kcache; a.vcache = p->vcache;
    a.k_nb1 = p->k_nb1; a.k_nb2 = p->k_nb2; a.v_nb1 = p->v_nb1; a.v_nb2 = p->v_nb2;
    a.H = p->H; a.D = p->D; a.C = p->C; a.T = p->T; a.scale = p->scale; a.out = p->out; a.out_ts = p->out_ts; a.n_groups = p->n_groups; a.write_only = p->write_only;
    return a;
}
// dequant_elem (hip_device.h) for a row known to live in global memory
__device__ __forceinline__ float dequant_elem_g(const GLOBAL_AS char * row, int type, int64_t i) {
    switch (type) {
        case GGML_TYPE_F32:  return ((const GLOBAL_AS float *) row)[i];
        case GGML_TYPE_F16:  return h2f(((const GLOBAL_AS uint16_t *) row)[i]);
        case GGML_TYPE_BF16: return bf2f(((const GLOBAL_AS uint16_t *) row)[i]);
        case GGML_TYPE_Q8_0: { const GLOBAL_AS block_q8_0 * b = (const GLOBAL_AS block_q8_0 *) row + i / 32; return b->qs[i % 32] * h2f(b->d); }
        case GGML_TYPE_Q4_0: {
            const GLOBAL_AS block_q4_0 * b = (const GLOBAL_AS block_q4_0 *) row + i / 32;
            const int j = (int) (i % 32);
            const int q = j < 16 ? (b->qs[j] & 0x0F) : (b->qs[j - 16] >> 4);
            return (q - 8) * h2f(b->d);
        }
        case GGML_TYPE_Q4_K: {
            const GLOBAL_AS block_q4_K * b = (const GLOBAL_AS block_q4_K *) row + i / 256;
            const int j = (int) (i % 256);
            const int sub = j / 32, l = j % 32;
            uint32_t sc[2], mn[2];
            const GLOBAL_AS uint32_t * sw = (const GLOBAL_AS uint32_t *) b->scales;
            q4k_unpack_scales_w(sw[0], sw[1], sw[2], sc, mn);
            const uint32_t s = (sc[sub >> 2] >> (8 * (sub & 3))) & 0xff, m = (mn[sub >> 2] >> (8 * (sub & 3))) & 0xff;
            const uint8_t qb = b->qs[(sub >> 1) * 32 + l];
            const int q = (sub & 1) ? (qb >> 4) : (qb & 0xF);
            const float d = h2f(b->d) * (float) s, mm = h2f(b->dmin) * (float) m;
            return d * (float) q - mm;
        }
        default: return NAN;
    }
}

// four consecutive elements i0 .. i0 + 3 (i0 a multiple of 4) of a row in global memory: dequant_elem_g's values with every load of a type issued before the
// first is used (element by element through the per-type switch the compiler waits for each element's loads in turn: 8 dependent round trips)
__device__ __forceinline__ void dequant4_g(const GLOBAL_AS char * row, int type, int i0, float out[4]) {
    switch (type) {
        case GGML_TYPE_F32: { const f32x4 t = *(const GLOBAL_AS f32x4 *) (row + (int64_t) i0 * 4); out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; break; }
        case GGML_TYPE_F16: case GGML_TYPE_BF16: {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 t = *(const GLOBAL_AS u32x2 *) (row + (int64_t) i0 * 2);
            const uint16_t h[4] = { (uint16_t) (t.x & 0xffff), (uint16_t) (t.x >> 16), (uint16_t) (t.y & 0xffff), (uint16_t) (t.y >> 16) };
#pragma unroll
            for (int k = 0; k < 4; k++) out[k] = type == GGML_TYPE_F16 ? h2f(h[k]) : bf2f(h[k]);
            break;
        }
        case GGML_TYPE_Q8_0: {
            const GLOBAL_AS char * b = row + (int64_t) (i0 / 32) * 34;
            const uint16_t d = *(const GLOBAL_AS uint16_t *) b, q0 = *(const GLOBAL_AS uint16_t *) (b + 2 + (i0 % 32)), q1 = *(const GLOBAL_AS uint16_t *) (b + 4 + (i0 % 32));
            const float df = h2f(d);
            out[0] = (int) (int8_t) (q0 & 0xff) * df; out[1] = (int) (int8_t) (q0 >> 8) * df; out[2] = (int) (int8_t) (q1 & 0xff) * df; out[3] = (int) (int8_t) (q1 >> 8) * df;
            break;
        }
        case GGML_TYPE_Q4_0: {
            const GLOBAL_AS char * b = row + (int64_t) (i0 / 32) * 18;
            const int j = i0 % 32;
            const uint16_t d = *(const GLOBAL_AS uint16_t *) b, q0 = *(const GLOBAL_AS uint16_t *) (b + 2 + (j & 15)), q1 = *(const GLOBAL_AS uint16_t *) (b + 4 + (j & 15));
            const float df = h2f(d);
            const int sh = j < 16 ? 0 : 4;
            const int q[4] = { (q0 >> sh) & 0xF, (q0 >> (8 + sh)) & 0xF, (q1 >> sh) & 0xF, (q1 >> (8 + sh)) & 0xF };
#pragma unroll
            for (int k = 0; k < 4; k++) out[k] = (q[k] - 8) * df;
            break;
        }
        case GGML_TYPE_Q4_K: {
            const GLOBAL_AS char * b = row + (int64_t) (i0 / 256) * 144;
            const int j = i0 % 256, sub = j / 32, l = j % 32;
            const GLOBAL_AS uint32_t * bw = (const GLOBAL_AS uint32_t *) b;
            const uint32_t dd = bw[0], u0 = bw[1], u1 = bw[2], u2 = bw[3], qw = *(const GLOBAL_AS uint32_t *) (b + 16 + (sub >> 1) * 32 + l);
            uint32_t sc[2], mn[2];
            q4k_unpack_scales_w(u0, u1, u2, sc, mn);
            const uint32_t s = (sc[sub >> 2] >> (8 * (sub & 3))) & 0xff, m = (mn[sub >> 2] >> (8 * (sub & 3))) & 0xff;
            const float d = h2f((uint16_t) (dd & 0xffff)) * (float) s, mm = h2f((uint16_t) (dd >> 16)) * (float) m;
#pragma unroll
            for (int k = 0; k < 4; k++) { const int qb = (int) ((qw >> (8 * k)) & 0xff); const int q = (sub & 1) ? (qb >> 4) : (qb & 0xF); out[k] = d * (float) q - mm; }
            break;
        }
        default: out[0] = out[1] = out[2] = out[3] = NAN; break;
    }
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------------
// Workgroup rendezvous at the hardware barrier (LDS traffic only is waited for: loads and stores stay in flight across it). Returns false when some
// wave has given up a bounded wait: everybody leaves together.
__device__ __forceinline__ bool wg_barrier(chain_ctl * ctl) {
    lds_barrier();
    return lds_load(&ctl->failed) == 0u;
}

// Shape of a phase, either read from its descriptor (dyn_shape) or a compile-time constant (the six phase kinds of the Depth transformer of
// moshika / PersonaPlex, lm.h:446-553 + tools/moshi-config.json: depformer_in 4096 -> 1024, in_proj 1024 -> 3072, out_proj 1024 -> 1024 behind the
// attention, linear_in 1024 -> 2 x 2816 in the paired form, linear_out 2816 -> 1024, linears[k] 1024 -> 2048 without a norm). With the shape known to the compiler
// the phase body loses its generic loops, its shape-dependent branches and most of its descriptor reads: what stays in the descriptor is pointers.
struct dyn_shape { static constexpr bool S = false; static constexpr int K = 0, M = 0, PAIR = 0, PRO = 0, XCH = 0, RES = 0, EMB = 0, SAVE = 0, AM = 0, PUB = 0; };
template <int K_, int M_, int PAIR_, int PRO_, int XCH_, int RES_, int EMB_, int SAVE_, int AM_, int PUB_>
struct fix_shape { static constexpr bool S = true; static constexpr int K = K_, M = M_, PAIR = PAIR_, PRO = PRO_, XCH = XCH_, RES = RES_, EMB = EMB_, SAVE = SAVE_, AM = AM_, PUB = PUB_; };
//                   K     M     pair  prologue     x_chain res emb save argmax n_pub
typedef fix_shape<4096, 1024,    0, MV_PLAIN,   0, 0, 1, 1, 0, 1024> shape_din;
typedef fix_shape<1024, 3072,    0, MV_RMSNORM, 1, 0, 0, 0, 0, 3072> shape_inproj;
typedef fix_shape<1024, 1024,    0, MV_ATTN,    1, 1, 0, 1, 0, 1024> shape_outproj;
typedef fix_shape<1024, 5632, 2816, MV_RMSNORM, 1, 0, 0, 0, 0, 2816> shape_linin;
typedef fix_shape<2816, 1024,    0, MV_PLAIN,   1, 1, 0, 1, 0, 1024> shape_linout;
typedef fix_shape<1024, 2048,    0, MV_PLAIN,   1, 0, 0, 0, 1,    0> shape_head;     // (linears[k] takes the last layer's output as it is: no norm in front, lm.h:527-531)

template <int G>
__global__ void __launch_bounds__(CH_THREADS) matvec_chain_kernel(chain_params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    xblk * xs = (xblk *) smem;
    float * xf = (float *) (xs + 16);
    float * part = xf + CH_XF_MAX;
    float * xres = part + CH_PART_MAX;
    float * xa = xres + CH_RES_MAX;
    float * attw = xa + 1024;
    chain_ctl * ctl = (chain_ctl *) (attw + CH_NCW * CH_ATTW);
    unsigned * desc = (unsigned *) (ctl + 1);   // [3][CH_DESC_DWORDS]: the descriptors of phases p - 1, p, p + 1 (by p % 3)

    // Descriptors reach the waves through LDS: read from the table with scalar loads they cost every wave several dependent scalar-cache misses at
    // the head of every phase (~1 us of the 5.7 us phase, gpurun_out/r3_sweep10.log). Wave 7 fetches phase p + 1's descriptor with ONE vector load at
    // the head of phase p and parks it in LDS before the "blocks ready" barrier; everybody copies it to scalar registers from there.
    const GLOBAL_AS u32x4 * desc_src = (const GLOBAL_AS u32x4 *) P.phases;
    constexpr int DL = CH_DESC_DWORDS / 4;   // 16-byte lanes per descriptor
    if (tid == 0) { ctl->failed = 0; ctl->token = 0; }
    if (wave == 7 && lane < DL) ((u32x4 *) desc)[lane] = desc_src[lane];
    __syncthreads();

    const int wg = blockIdx.x, grid = gridDim.x;
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    // LDS -> scalar registers, field by field: only what a phase actually uses is read (a whole-struct copy is ~100 LDS words plus as many
    // v_readfirstlane and cost 0.7 us per phase, gpurun_out/r3_sweep11.log)
#define CH_LD(d, base, f) (d).f = desc_field<typename std::remove_reference<decltype((d).f)>::type>(base, (int) (offsetof(chain_phase, f) / 4))
    auto read_desc_weights = [&](int q, chain_phase & d) {   // what request_weights needs
        const unsigned * b = desc + (q % 3) * CH_DESC_DWORDS;
        CH_LD(d, b, w); CH_LD(d, b, row_bytes); CH_LD(d, b, M); CH_LD(d, b, nb); CH_LD(d, b, rows_wg); CH_LD(d, b, pair_F); CH_LD(d, b, fmt);
    };
    // head: what the prologue and the dot stage use; tail: what the epilogue uses, read where it starts (short-lived scalars: the kernel is at the
    // scalar-register limit, and a field kept from the head of the phase is a spill and a reload)
    auto read_desc = [&](int q, chain_phase & d) {
        const unsigned * b = desc + (q % 3) * CH_DESC_DWORDS;
        CH_LD(d, b, K); CH_LD(d, b, M); CH_LD(d, b, nb); CH_LD(d, b, rows_wg); CH_LD(d, b, pair_F); CH_LD(d, b, prologue); CH_LD(d, b, x_chain);
        CH_LD(d, b, x); CH_LD(d, b, alpha); CH_LD(d, b, eps); CH_LD(d, b, n_in); CH_LD(d, b, emb_chain); CH_LD(d, b, argmax); CH_LD(d, b, fmt);
        if (d.prologue == MV_ATTN) {
            CH_LD(d, b, q_off); CH_LD(d, b, k_off); CH_LD(d, b, v_off);
            CH_LD(d, b, at.q_hs); CH_LD(d, b, at.k_hs); CH_LD(d, b, at.v_hs); CH_LD(d, b, at.rot); CH_LD(d, b, at.mask); CH_LD(d, b, at.index);
            CH_LD(d, b, at.kcache); CH_LD(d, b, at.vcache); CH_LD(d, b, at.k_nb1); CH_LD(d, b, at.k_nb2); CH_LD(d, b, at.v_nb1); CH_LD(d, b, at.v_nb2);
            CH_LD(d, b, at.H); CH_LD(d, b, at.C); CH_LD(d, b, at.scale);
        }
    };
    auto read_desc_tail = [&](int q, chain_phase & d) {
        const unsigned * b = desc + (q % 3) * CH_DESC_DWORDS;
        CH_LD(d, b, res); CH_LD(d, b, residual); CH_LD(d, b, save); CH_LD(d, b, y); CH_LD(d, b, n_pub);
        CH_LD(d, b, emb.table); CH_LD(d, b, emb.row_bytes); CH_LD(d, b, emb.n_rows); CH_LD(d, b, emb.type); CH_LD(d, b, emb.index); CH_LD(d, b, emb.scale);
    };
    // The weights of ONE phase live in registers: pass ps covers super-blocks 64 ps .. 64 ps + 63 of the workgroup's rows, 8 lanes per super-block -
    // every lane of a group holds the 16-byte header (d, dmin, 6-bit scales: one fetch for the group) and its own 16-byte nibble chunk.
    constexpr int CH_PMAX = CH_PMAX_OF(G);
    // Q8_0 (descriptor-driven phases only): 8 lanes per 272-byte chunk of 256 weights, lane j its own 34-byte block - d (F16) + 32 int8. The block starts
    // 2 bytes off a dword for odd j, so the lane loads the 9 dwords from byte 34 j - (j odd ? 2 : 0) of the chunk (two 16-byte loads and one dword; dword
    // alignment is all global loads need): wh = dwords 0..3, wq = 4..7, we = 8. Even j: d = low half of dword 0, q = bytes 2..33; odd j: d = high half
    // of dword 0, q = dwords 1..8.
    u32x4 wh[CH_PMAX], wq[CH_PMAX];
    uint32_t we[CH_PMAX];
    auto request_weights = [&](const chain_phase & pq) {   // that phase's super-blocks of this workgroup
        if (pq.fmt == 1) {
            const bool pr = pq.pair_F > 0;
            const long long rt = pr ? pq.pair_F : (long long) pq.M;
            const long long r0 = (long long) wg * pq.rows_wg;
            const int rw = (int) (rt - r0 < pq.rows_wg ? (rt - r0 > 0 ? rt - r0 : 0) : pq.rows_wg);
            const int nseg = rw * pq.nb, nall = pr ? 2 * nseg : nseg;
            const GLOBAL_AS char * w0 = gp(pq.w) + r0 * pq.row_bytes, * w1 = gp(pq.w) + (r0 + pq.pair_F) * pq.row_bytes;
            const int j = lane & 7, boff = 34 * j - ((j & 1) ? 2 : 0);
#pragma unroll
            for (int ps = 0; ps < CH_PMAX; ps++) {
                if (ps * (CH_NCW * 8) < nall) {   // (wave-uniform)
                    const int sb = ps * (CH_NCW * 8) + wave * 8 + (lane >> 3);
                    const int sbc = sb < nall ? sb : nall - 1;
                    const GLOBAL_AS char * src = (sbc < nseg ? w0 + (long long) sbc * 272 : w1 + (long long) (sbc - nseg) * 272) + boff;
                    wh[ps] = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) src);
                    wq[ps] = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) (src + 16));
                    we[ps] = __builtin_nontemporal_load((const GLOBAL_AS uint32_t *) (src + 32));
                }
            }
            return;
        }
        const bool pr = pq.pair_F > 0;
        const long long rt = pr ? pq.pair_F : (long long) pq.M;
        const long long r0 = (long long) wg * pq.rows_wg;
        const int rw = (int) (rt - r0 < pq.rows_wg ? (rt - r0 > 0 ? rt - r0 : 0) : pq.rows_wg);
        const int nseg = rw * pq.nb, nall = pr ? 2 * nseg : nseg;
        const GLOBAL_AS u32x4 * w0 = (const GLOBAL_AS u32x4 *) (gp(pq.w) + r0 * pq.row_bytes);
        const GLOBAL_AS u32x4 * w1 = (const GLOBAL_AS u32x4 *) (gp(pq.w) + (r0 + pq.pair_F) * pq.row_bytes);
#pragma unroll
        for (int ps = 0; ps < CH_PMAX; ps++) {
            if (ps * (CH_NCW * 8) < nall) {   // (wave-uniform)
                const int sb = ps * (CH_NCW * 8) + wave * 8 + (lane >> 3);
                const int sbc = sb < nall ? sb : nall - 1;
                const GLOBAL_AS u32x4 * src = sbc < nseg ? w0 + sbc * 9 : w1 + (sbc - nseg) * 9;
                wh[ps] = __builtin_nontemporal_load(src);
                wq[ps] = __builtin_nontemporal_load(src + 1 + (lane & 7));
            }
        }
    };
    {
        chain_phase first;
        read_desc_weights(0, first);
        request_weights(first);
    }

    const __amdgpu_buffer_rsrc_t gb = make_rsrc(P.gbuf, 2u * CH_XF_MAX * 8u);
    const __amdgpu_buffer_rsrc_t cb = make_rsrc(P.cand, 2u * 2u * (unsigned) grid * 8u);
    bool alive = true;
    auto give_up = [&]() { if (lane == 0) { lds_store(&ctl->failed, 1u); *gp(P.err) = 2u; } };

    // one phase; SH = its shape (dyn_shape: everything from the descriptor). Returns false when the workgroup has to leave.
    auto phase = [&](auto shape_tag, int p) -> bool {
        using SH = decltype(shape_tag);
        CH_STAMP(10);
        chain_phase ph;
        const unsigned * dbase = desc + (p % 3) * CH_DESC_DWORDS;
        if (!SH::S) read_desc(p, ph);
        else {
            // pointers only (and what the shape does not fix)
            if (!SH::XCH) CH_LD(ph, dbase, x);
            if (SH::PRO == MV_RMSNORM) { CH_LD(ph, dbase, alpha); CH_LD(ph, dbase, eps); }
            if (SH::EMB) CH_LD(ph, dbase, emb_chain);
            if (SH::PRO == MV_ATTN) {
                CH_LD(ph, dbase, q_off); CH_LD(ph, dbase, k_off); CH_LD(ph, dbase, v_off);
                CH_LD(ph, dbase, at.q_hs); CH_LD(ph, dbase, at.k_hs); CH_LD(ph, dbase, at.v_hs); CH_LD(ph, dbase, at.rot); CH_LD(ph, dbase, at.mask); CH_LD(ph, dbase, at.index);
                CH_LD(ph, dbase, at.kcache); CH_LD(ph, dbase, at.vcache); CH_LD(ph, dbase, at.k_nb1); CH_LD(ph, dbase, at.k_nb2); CH_LD(ph, dbase, at.v_nb1); CH_LD(ph, dbase, at.v_nb2);
                CH_LD(ph, dbase, at.H); CH_LD(ph, dbase, at.C); CH_LD(ph, dbase, at.scale);
            }
        }
        // wave 7 asks for the next descriptor now; it is parked in LDS before this phase's "blocks ready" barrier
        u32x4 next_desc = (u32x4) { 0u, 0u, 0u, 0u };
        const bool stage_next = wave == 7 && p + 1 < P.n_phases;
        if (stage_next && lane < DL) next_desc = desc_src[(p + 1) * DL + lane];
        const unsigned tag_in = tag_base | (unsigned) p, tag_out = tag_base | (unsigned) (p + 1);
        const int nb = SH::S ? SH::K / 256 : ph.nb, K = SH::S ? SH::K : ph.K;
        const bool paired = SH::S ? SH::PAIR > 0 : ph.pair_F > 0;
        const long long pair_F = SH::S ? (long long) SH::PAIR : ph.pair_F;
        const int rows_wg = SH::S ? (SH::PAIR ? SH::PAIR : SH::M) / G : ph.rows_wg;
        const long long rows_total = paired ? pair_F : (long long) (SH::S ? SH::M : ph.M);
        const long long row0 = (long long) wg * rows_wg;
        const int rows = SH::S ? rows_wg : (int) (rows_total - row0 < rows_wg ? (rows_total - row0 > 0 ? rows_total - row0 : 0) : rows_wg);
        const int nblk_seg = rows * nb, nblk = paired ? 2 * nblk_seg : nblk_seg;
        const bool is_attn = (SH::S ? SH::PRO : ph.prologue) == MV_ATTN, is_rms = (SH::S ? SH::PRO : ph.prologue) == MV_RMSNORM;
        const bool x_chain = SH::S ? SH::XCH != 0 : ph.x_chain != 0;
        const bool emb_chain = SH::S ? (SH::EMB && ph.emb_chain) : ph.emb_chain != 0;
        const bool has_argmax = SH::S ? SH::AM != 0 : ph.argmax != 0;
        const int n_in = SH::S ? (SH::PRO == MV_ATTN ? 3 * SH::K : SH::K) : ph.n_in;

        CH_STAMP(0);

        // ---- the hand-off poll goes out first; the loads that do not depend on the previous phase (norm weights, x from memory, the attention's
        // ring rows) ride behind it. A phase without attention takes its blocks straight into registers: block b = 256 values = lane l's granules
        // 4 l .. 4 l + 3 = two 16-byte loads, waves take blocks w, w + 8 - no LDS staging, no rendezvous before the norm.
        const bool in_regs = x_chain && !is_attn;
        const unsigned in_base = (unsigned) ((p - 1) & 1) * (CH_XF_MAX * 8u);
        const bool has0 = wave < nb, has1 = wave + CH_NCW < nb;   // (wave-uniform) only waves that own a block poll: every poll is a fabric read
        u32x4 gq[2][2];
        gq[0][0] = gq[0][1] = gq[1][0] = gq[1][1] = (u32x4) { 0u, tag_in, 0u, tag_in };
        auto issue_poll = [&]() {
            if (has0) {
                const unsigned o0 = in_base + ((unsigned) wave * 256u + (unsigned) lane * 4u) * 8u;
                gq[0][0] = ld16_agent(gb, o0); gq[0][1] = ld16_agent(gb, o0 + 16u);
                if (has1) {
                    const unsigned o1 = o0 + CH_NCW * 256u * 8u;
                    gq[1][0] = ld16_agent(gb, o1); gq[1][1] = ld16_agent(gb, o1 + 16u);
                }
            }
        };
        if (in_regs) { for (int i = 0; i < P.delay; i++) __builtin_amdgcn_s_sleep(1); issue_poll(); }
        f32x4 al[2], xm[2];
        u32x4 kq[2], vq[2];
        int at_slot = 0; float at_m = 0.f, at_rc = 1.f, at_rs = 0.f;
        __amdgpu_buffer_rsrc_t kr = gb, vr = gb;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int b = wave + r * CH_NCW;
            al[r] = (f32x4) { 1.f, 1.f, 1.f, 1.f };
            if (is_rms && b < nb) al[r] = *(const GLOBAL_AS f32x4 *) (ph.alpha + b * 256 + lane * 4);
            xm[r] = (f32x4) { 0.f, 0.f, 0.f, 0.f };
            if (!x_chain && b < nb) xm[r] = *(const GLOBAL_AS f32x4 *) (ph.x + b * 256 + lane * 4);
        }
        if (is_attn) {
            const attn_args & at = ph.at;
            kr = make_rsrc(at.kcache, (unsigned) ((int64_t) at.H * at.k_nb2));
            vr = make_rsrc(at.vcache, (unsigned) ((int64_t) at.H * at.v_nb2));
            chain_attn_ring_loads(at, wave * 2, lane, kq, vq, kr, vr);
            const int sub = lane / 8, cc = sub < at.C ? sub : at.C - 1, pp = lane < 32 ? lane : lane - 32;
            at_slot = gp(at.index)[0];
            at_m = gp(at.mask)[cc];
            if (at.rot) { at_rc = gp(at.rot)[pp]; at_rs = gp(at.rot)[32 + pp]; }
        }
        CH_STAMP(1);

        // ---- stage 1: the activation vector -> Q8_K blocks in xs
        const float * xsrc = xf;   // LDS source of the blocks of an attention phase (xa: attention output)
        if (in_regs) {
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < 2; r++) ok = ok && gq[r][0].y == tag_in && gq[r][0].w == tag_in && gq[r][1].y == tag_in && gq[r][1].w == tag_in;
                if (__all(ok)) break;
                if (++spins > CH_SPIN_MAX || lds_load(&ctl->failed)) { give_up(); break; }
                __builtin_amdgcn_s_sleep(1);
                issue_poll();
            }
            CH_STAMP(2);
        } else if (x_chain) {
            if (!gather_vector(gb, in_base, n_in, tag_in, xf, wave, lane, ctl)) give_up();
            CH_STAMP(2);
            if (!wg_barrier(ctl)) return false;
            {
                chain_attn_wave(ph.at, xf + ph.q_off, xf + ph.k_off, xf + ph.v_off, wave * 2, lane, attw + wave * CH_ATTW, xa + wave * 128, kq, vq, wg == 0, kr, vr,
                                at_slot, at_m, at_rc, at_rs
#if defined(CH_LOG)
                                , wg == 0 ? 0 : wg == grid - 1 ? 1 : -1, p
#endif
                                );
            }
            if (!wg_barrier(ctl)) return false;
            xsrc = xa;
        }
        CH_STAMP(3);
        float v[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int b = wave + r * CH_NCW;
            f32x4 t = xm[r];
            if (in_regs) t = (f32x4) { __uint_as_float(gq[r][0].x), __uint_as_float(gq[r][0].z), __uint_as_float(gq[r][1].x), __uint_as_float(gq[r][1].z) };
            else if (x_chain && b < nb) t = *(const f32x4 *) (xsrc + b * 256 + lane * 4);
            if (b >= nb) t = (f32x4) { 0.f, 0.f, 0.f, 0.f };   // (a clamped re-read does not count towards the norm)
            v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
        }
        if (is_rms) {
            // matvec_q4k_kernel's order: per-thread squares in double, wave butterfly, waves added in index order
            {
                double acc = 0;
#pragma unroll
                for (int r = 0; r < 2; r++)
                    if (wave + r * CH_NCW < nb)   // (wave-uniform; a block that does not exist would add exact zeros)
#pragma unroll
                        for (int k = 0; k < 4; k++) acc += (double) (v[r][k] * v[r][k]);
                acc = wave_allsum_f64(acc);
                if (lane == 0) ctl->sumsq[wave] = acc;
            }
            if (!wg_barrier(ctl)) return false;
            {
                double tot = 0;
#pragma unroll
                for (int w = 0; w < CH_NCW; w++) tot += ctl->sumsq[w];
                // (K a power of two - every norm of the Depth transformer: the product is the quotient, exactly, without the f64 division's chain)
                const float mean = (K & (K - 1)) == 0 ? (float) (tot * (1.0 / (double) K)) : (float) (tot / (double) K);
                const float scale = 1.0f / sqrtf(mean + ph.eps);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (wave + r * CH_NCW >= nb) continue;
                    const float a4[4] = { al[r].x, al[r].y, al[r].z, al[r].w };
#pragma unroll
                    for (int k = 0; k < 4; k++) v[r][k] = a4[k] * (v[r][k] * scale);
                }
            }
        }
        {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int b = wave + r * CH_NCW;
                if (b < nb) { if (!SH::S && ph.fmt == 1) quantize_block_q80((xblk80 *) (xs + b), v[r], lane); else quantize_block_q8k(xs + b, v[r], lane); }
            }
        }
        if (stage_next && lane < DL) ((u32x4 *) (desc + ((p + 1) % 3) * CH_DESC_DWORDS))[lane] = next_desc;
        CH_STAMP(4);
        if (!wg_barrier(ctl)) return false;
        CH_STAMP(5);

        // ---- stage 2: super-block dots out of the registers (the WS = 1 arithmetic of matvec_q4k_kernel: 8 lanes per super-block)
        {
            if (!SH::S && ph.fmt == 1) {
                // Q8_0: lane j dots its 32-wide block against the activation's block j (vec_dot_q8_0_q8_0: sumi * (d_w * d_x)); the chunk's eight terms are
                // then added in block order, starting from 0 - q80_q80_sb_dot's float sequence - by the group's first lane through row-shift DPP reads
#pragma unroll
                for (int ps = 0; ps < CH_PMAX; ps++) {
                    if (ps * (CH_NCW * 8) >= nblk) break;
                    const int sb = ps * (CH_NCW * 8) + wave * 8 + (lane >> 3);
                    const int sbc = sb < nblk ? sb : nblk - 1;
                    const int j8 = lane & 7;
                    const bool odd = (j8 & 1) != 0;
                    const xblk80 * xb = (const xblk80 *) (xs + (sbc % nb));
                    const uint32_t D[9] = { wh[ps].x, wh[ps].y, wh[ps].z, wh[ps].w, wq[ps].x, wq[ps].y, wq[ps].z, wq[ps].w, we[ps] };
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
                    if (j8 == 0 && sb < nblk) part[sb] = sumf;
                }
            } else
#pragma unroll
            for (int ps = 0; ps < CH_PMAX; ps++) {
                if (ps * (CH_NCW * 8) >= nblk) break;
                const int sb = ps * (CH_NCW * 8) + wave * 8 + (lane >> 3);
                const int sbc = sb < nblk ? sb : nblk - 1;
                const int j8 = lane & 7, g32 = j8 >> 1, hf = j8 & 1;
                const xblk * xb = xs + (sbc % nb);
                const uint32_t hw[4] = { wh[ps].x, wh[ps].y, wh[ps].z, wh[ps].w };
                uint32_t sc[2], mn[2];
                q4k_unpack_scales_w(hw[1], hw[2], hw[3], sc, mn);
                const u32x4 ylo = *(const u32x4 *) (xb->q + 64 * g32 + 16 * hf), yhi = *(const u32x4 *) (xb->q + 64 * g32 + 32 + 16 * hf);
                const uint32_t qw[4] = { wq[ps].x, wq[ps].y, wq[ps].z, wq[ps].w }, yl[4] = { ylo.x, ylo.y, ylo.z, ylo.w }, yh[4] = { yhi.x, yhi.y, yhi.z, yhi.w };
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
                if (j8 == 0 && sb < nblk) {
                    const float d = h2f((uint16_t) (hw[0] & 0xffff)) * xb->d, dmin = h2f((uint16_t) (hw[0] >> 16)) * xb->d;
                    part[sb] = d * (float) isum - dmin * (float) msum;
                }
            }
            // workgroup 0 wrote the new K / V ring rows of an attention phase with agent-scope stores: they are drained HERE - nothing else of this wave
            // is in flight at this point, the stores went out a microsecond ago - so that every publication of this phase (st_granule below) and of all
            // later ones follows them (Guideline 16 R1: sc1 payload, vmcnt(0) in every storing wave, then the signal). Readers touch the rows 26+ phases on.
            if (is_attn && wg == 0) wait_vmcnt<0>();
            // the next phase's weights go out now: this wave's registers are free again, and the request has the whole hand-off to land in
            if (p + 1 < P.n_phases) { chain_phase nx; read_desc_weights(p + 1, nx); request_weights(nx); }
            CH_STAMP(6);
            // the token this phase's epilogue needs (the previous phase's arg-max): wave 0 merges the candidates
            if (emb_chain && wave == 0) {
                int token = 0;
                if (!gather_token(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_in, lane, ctl, token)) give_up();
                if (lane == 0) {
                    ctl->token = token;
                    if (wg == 0) {
                        const unsigned * db = desc + (p % 3) * CH_DESC_DWORDS;
                        CH_LD(ph, db, prev_out[0]); CH_LD(ph, db, prev_out[1]);
                        if (ph.prev_out[0]) *gp(ph.prev_out[0]) = token;
                        if (ph.prev_out[1]) *gp(ph.prev_out[1]) = token;
                    }
                }
            }
        }
        if (!wg_barrier(ctl)) return false;   // partial sums and token visible
        CH_STAMP(7);
        if (!SH::S) read_desc_tail(p, ph);
        else {
            CH_LD(ph, dbase, y);
            if (SH::EMB) { CH_LD(ph, dbase, emb.table); CH_LD(ph, dbase, emb.row_bytes); CH_LD(ph, dbase, emb.n_rows); CH_LD(ph, dbase, emb.type); CH_LD(ph, dbase, emb.index); CH_LD(ph, dbase, emb.scale); }
        }
        const int res = SH::S ? SH::RES : ph.res;
        const bool save = SH::S ? SH::SAVE != 0 : ph.save != 0;
        const int n_pub = SH::S ? SH::PUB : ph.n_pub;
        const bool has_emb = SH::S ? SH::EMB != 0 : ph.emb.table != nullptr;

        // ---- stage 3: fixed-order row sums, epilogue, publication
        const unsigned pub_base = (unsigned) (p & 1) * CH_XF_MAX;
        float best = -INFINITY; int bi = -1;
        {
            if (paired) {
                for (int rr = tid >> 4; rr < rows; rr += CH_NCW * 4) {
                    float sl = 0.f, sr = 0.f;
                    for (int j = tid & 15; j < nb; j += 16) { sl += part[rr * nb + j]; sr += part[(rows + rr) * nb + j]; }
                    sl = row16_allsum_f32(sl); sr = row16_allsum_f32(sr);
                    if ((tid & 15) == 0) {
                        const float g = (sl / (1.0f + expf(-sl))) * sr;
                        if (n_pub) st_granule(P.gbuf + pub_base + row0 + rr, tag_out, __float_as_uint(g));
                        gp(ph.y)[row0 + rr] = sl; gp(ph.y)[pair_F + row0 + rr] = sr;
                    }
                }
            } else {
                float emb_scale = 1.f; const GLOBAL_AS char * emb_row = nullptr;
                if (has_emb) {
                    int64_t r = emb_chain ? (int64_t) ctl->token : (int64_t) *gp(ph.emb.index);
                    if (r < 0 || r >= ph.emb.n_rows) r = 0;
                    emb_row = gp(ph.emb.table) + r * ph.emb.row_bytes;
                    if (ph.emb.scale) emb_scale = *gp(ph.emb.scale);
                }
                for (int rr = tid >> 4; rr < rows; rr += CH_NCW * 4) {
                    float sum = 0.f;
                    for (int j = tid & 15; j < nb; j += 16) sum += part[rr * nb + j];
                    sum = row16_allsum_f32(sum);
                    if ((tid & 15) == 0) {
                        const int64_t row = row0 + rr;
                        if (res == 1) sum = xres[rr] + sum;
                        else if (res == 2) sum = gp(ph.residual)[row] + sum;
                        else if (emb_row) {
                            float e = dequant_elem_g(emb_row, ph.emb.type, row);
                            if (ph.emb.scale) e = e * emb_scale;
                            sum = sum + e;
                        }
                        if (save) xres[rr] = sum;
                        if (n_pub) st_granule(P.gbuf + pub_base + row, tag_out, __float_as_uint(sum));
                        gp(ph.y)[row] = sum;
                        if (sum >= best) { best = sum; bi = (int) row; }   // rows ascend per thread: '>=' keeps the last maximum
                    }
                }
            }
        }
        CH_STAMP(8);
        if (has_argmax) {
            {
                am_wave(best, bi);
                if (lane == 0) { ctl->am_v[wave] = best; ctl->am_i[wave] = bi; }
            }
            if (!wg_barrier(ctl)) return false;
            if (tid == 0) {
                for (int w = 1; w < CH_NCW; w++) am_merge(best, bi, ctl->am_v[w], ctl->am_i[w]);
                u64 * c = P.cand + (size_t) (p & 1) * 2 * grid + 2 * wg;
                st_granule(c, tag_out, __float_as_uint(best));
                st_granule(c + 1, tag_out, (unsigned) bi);
            }
        }
        CH_STAMP(9);
        return true;
    };
    for (int p = 0; p < P.n_phases && alive; p++) {
        const int kind = (int) __builtin_amdgcn_readfirstlane(desc[(p % 3) * CH_DESC_DWORDS + offsetof(chain_phase, kind) / 4]);
        switch (kind) {
            case 1:  alive = phase(shape_din(), p); break;
            case 2:  alive = phase(shape_inproj(), p); break;
            case 3:  alive = phase(shape_outproj(), p); break;
            case 4:  alive = phase(shape_linin(), p); break;
            case 5:  alive = phase(shape_linout(), p); break;
            case 6:  alive = phase(shape_head(), p); break;
            default: alive = phase(dyn_shape(), p); break;
        }
    }
    if (!alive) return;

    // the chain ends in an arg-max: workgroup 0 merges the candidates and writes the token
    chain_phase last;
    { const unsigned * b = desc + ((P.n_phases - 1) % 3) * CH_DESC_DWORDS; CH_LD(last, b, argmax); CH_LD(last, b, argmax_out[0]); CH_LD(last, b, argmax_out[1]); }
    if (wg == 0 && wave == 0 && last.argmax) {
        const int p = P.n_phases;
        int token = 0;
        if (gather_token(cb, (unsigned) ((p - 1) & 1) * (2u * (unsigned) grid * 8u), grid, tag_base | (unsigned) p, lane, ctl, token)) {
            if (lane == 0) { if (last.argmax_out[0]) *gp(last.argmax_out[0]) = token; if (last.argmax_out[1]) *gp(last.argmax_out[1]) = token; }
        } else { give_up(); return; }
    }
    if (wg == 0 && tid == 0) *gp(P.launch_seq) = launch + 1u;
}

#include "hip_chain_nest.h"
#include "hip_chain_nest80.h"
#include "hip_chain_mimi.h"

// ---- host side -----------------------------------------------------------------------------------------------------------------
static int chain_env(const char * name, int def) { const char * v = getenv(name); return v ? atoi(v) : def; }
bool k_chain_default_on() { static const int on = chain_env("MI355X_CHAIN", 1); return on != 0; }
static size_t chain_smem() { return 16 * XBLK_BYTES + (size_t) (CH_XF_MAX + CH_PART_MAX + CH_RES_MAX + 1024 + CH_NCW * CH_ATTW) * 4 + sizeof(chain_ctl) + 3 * sizeof(chain_phase); }
// The chain kernel's workgroups spin on each other's granules: ALL of them must be resident at once. The grid is therefore a function of the device (and
// of the compute units the caller's stream may use): the largest of 256 / 128 / 64 - capped by MI355X_CHAIN_GRID - that
// hipOccupancyMaxActiveBlocksPerMultiprocessor x usable CUs can hold; 0 when not even 64 fit (partitioned or CU-masked devices): the planner then
// keeps one launch per mat-vec. MI355X_CHAIN_GRID_FORCE (tests) skips the check.
static int chain_grid_for(int usable_cus) {
    static const int want = chain_env("MI355X_CHAIN_GRID", 256), force = chain_env("MI355X_CHAIN_GRID_FORCE", 0);
    const int cap = want < 8 ? 8 : want > 256 ? 256 : want;
    if (force) return cap;
    static std::map<int, int> memo;
    auto it = memo.find(usable_cus);
    if (it != memo.end()) return it->second;
    static bool granted = false;
    if (!granted) {
        HIP_CHECK(hipFuncSetAttribute((const void *) matvec_chain_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *) matvec_chain_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *) matvec_chain_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        granted = true;
    }
    int grid = 0;
    const int cands[4] = { cap, 256, 128, 64 };
    for (int k = 0; k < 4 && !grid; k++) {
        const int g = cands[k];
        if (g > cap) continue;
        int per_cu = 0;
        const void * fn = g == 256 ? (const void *) matvec_chain_kernel<256> : g == 128 ? (const void *) matvec_chain_kernel<128> : (const void *) matvec_chain_kernel<64>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, CH_THREADS, chain_smem()) != hipSuccess) per_cu = 0;
        if ((long long) per_cu * usable_cus >= g) grid = g;
    }
    if (chain_env("MI355X_CHAIN_VERBOSE", 0)) fprintf(stderr, "chain engine: %d usable CUs -> grid %d\n", usable_cus, grid);
    memo[usable_cus] = grid;
    return grid;
}

struct chain_plan {
    chain_params P;
    int grid; size_t smem;
    std::vector<chain_phase> phases;
    std::vector<attn_args> attns;
    // the run as a compile-time step program (hip_chain_nest.h): the phases are steps x { din, L x { in_proj, out_proj, linear_in, linear_out }, head } at
    // the Depth transformer's shapes
    bool nest = false;
    nest_params NP;
    size_t nest_smem = 0;
    std::vector<char> nest_tables;
    // ... or as the Q8_0 step program of the tts-shaped Depth transformer (hip_chain_nest80.h): the run then also holds attention and low-rank embedding steps
    bool mimi = false;           // a Mimi transformer as one launch (hip_chain_mimi.h)
    mimi_params MM;
    bool nest80 = false;
    nest80_params N80;
    int64_t n80_weight_bytes = 0;
    int n80_len = 0;
};

static bool overlaps(const void * a, size_t an, const void * b, size_t bn) {
    return (const char *) a < (const char *) b + bn && (const char *) b < (const char *) a + an;
}

// Greedy prefix: the first `len` of the n consecutive mat-vecs that one launch can execute. Fills `out` / `attns` (host form) and returns len
// (0: none). Rules: every phase but the first waits on its predecessor (x is the predecessor's output, or its embedding index is the
// predecessor's arg-max) - that is what lets two hand-off buffers alternate by phase parity; anything a phase reads that an earlier phase of
// the run writes must travel by one of the in-launch mechanisms (hand-off vector, kept rows, arg-max candidates) or the run is cut there.
static int chain_analyse(const mv_args * mv, int n, int G, std::vector<chain_phase> & out, std::vector<attn_args> & attns) {
    out.clear(); attns.clear();
    static const int q80_on = chain_env("MI355X_CHAIN_Q80", 1);
    std::vector<int> res_from;   // phase whose rows a phase adds as its residual (-1: none / memory)
    auto ybytes = [&](int j) { return (size_t) out[(size_t) j].M * 4; };
    for (int i = 0; i < n && i < CH_MAX_PHASES; i++) {   // (the hand-off tag keeps 12 bits for the phase index + 1)
        const mv_args & a = mv[i];
        chain_phase ph;
        memset(&ph, 0, sizeof(ph));
        // ---- the mat-vec itself
        if (a.special) break;   // (an attention / low-rank embedding step: only a step program takes those)
        if ((a.wtype != GGML_TYPE_Q4_K && a.wtype != GGML_TYPE_Q8_0) || a.ncols != 1 || a.K % 256 != 0 || a.K > 4096 || a.pair_F != 0 || a.x_out || a.out_scale || a.out_act) break;
        if (!a.ticket && (a.argmax_out[0] || a.argmax_out[1])) break;
        if (a.row_bytes != (a.K / 256) * (a.wtype == GGML_TYPE_Q8_0 ? 272 : 144) || ((uintptr_t) a.w & 15)) break;
        if (a.wtype == GGML_TYPE_Q8_0 && !q80_on) break;
        ph.fmt = a.wtype == GGML_TYPE_Q8_0 ? 1 : 0;
        if (!(a.prologue == MV_PLAIN || a.prologue == MV_RMSNORM || a.prologue == MV_GATE_SILU || a.prologue == MV_ATTN)) break;
        if ((a.M * (a.K / 256) + 63) / 64 > 512) break;   // the large matrices keep their own LDS-tile kernel
        ph.w = a.w; ph.row_bytes = a.row_bytes; ph.K = (int) a.K; ph.M = (int) a.M; ph.nb = (int) (a.K / 256);
        ph.rows_wg = (int) ((a.M + G - 1) / G);
        ph.prologue = a.prologue; ph.x = a.x; ph.alpha = a.alpha; ph.eps = a.eps; ph.y = a.y;
        ph.attn = -1;
        if (ph.rows_wg > CH_RES_MAX || ph.rows_wg * ph.nb > CH_PART_MAX) break;
        bool bad = false;
        // ---- where x comes from
        if (a.prologue != MV_ATTN) {
            const size_t xn = (size_t) (a.prologue == MV_GATE_SILU ? 2 * a.K : a.K) * 4;
            int x_from = -1;
            for (int j = 0; j < i; j++) if (overlaps(a.x, xn, out[(size_t) j].y, ybytes(j))) { if (x_from >= 0) bad = true; x_from = j; }
            if (x_from >= 0) {
                const chain_phase & pj = out[(size_t) x_from];
                if (x_from != i - 1 || a.x != pj.y || xn != ybytes(x_from) || pj.argmax || pj.pair_F) bad = true;
            }
            if (bad) break;
            if (a.prologue == MV_GATE_SILU) {
                // silu(h[:K]) * h[K:] with h = the previous phase's output: that phase takes the paired form and hands on the K gate values
                if (x_from < 0) break;
                chain_phase & pj = out[(size_t) x_from];
                if (pj.res || pj.emb.table || pj.argmax || res_from.back() >= 0) break;
                const int rw = (int) ((a.K + G - 1) / G);
                if (rw > CH_RES_MAX || 2 * rw * pj.nb > CH_PART_MAX) break;
                pj.pair_F = a.K; pj.rows_wg = rw;
                ph.prologue = MV_PLAIN;
            }
            ph.x_chain = x_from >= 0;
        } else {
            // q / k / v are slices of the previous phase's output (in_proj); workgroup 0 writes the new row to the ring
            if (i == 0 || !a.attn) break;
            const attn_args & at = *a.attn;
            const chain_phase & pj = out[(size_t) i - 1];
            if (at.T != 1 || at.D != 64 || at.H != 2 * CH_NCW || at.C < 1 || at.C > 8 || (int64_t) at.H * at.D != a.K || pj.argmax || pj.pair_F) break;
            const float * lo = pj.y, * hi = pj.y + pj.M;
            auto inside = [&](const float * q, int64_t hs) { return hs >= 0 && q >= lo && q + (at.H - 1) * hs + at.D <= hi; };
            if (!inside(at.q, at.q_hs) || !inside(at.k, at.k_hs) || !inside(at.v, at.v_hs)) break;
            ph.q_off = (int) (at.q - lo); ph.k_off = (int) (at.k - lo); ph.v_off = (int) (at.v - lo);
            ph.attn = (int) attns.size();
            ph.x_chain = 1;
        }
        // ---- epilogue
        int rf = -1;
        if (a.residual) {
            for (int j = 0; j < i; j++) if (overlaps(a.residual, (size_t) a.M * 4, out[(size_t) j].y, ybytes(j))) rf = j;
            if (rf >= 0) {
                const chain_phase & pj = out[(size_t) rf];
                if (a.residual != pj.y || pj.M != ph.M || pj.rows_wg != ph.rows_wg || pj.pair_F || pj.argmax) break;
                ph.res = 1;
            } else ph.res = 2;
            ph.residual = a.residual;
        }
        if (a.res_embed.table) {
            ph.emb = a.res_embed;
            for (int j = 0; j < i; j++) {
                const chain_phase & pj = out[(size_t) j];
                if (pj.argmax && ((const void *) a.res_embed.index == (const void *) pj.argmax_out[0] || (const void *) a.res_embed.index == (const void *) pj.argmax_out[1])) {
                    if (j != i - 1) bad = true;
                    ph.emb_chain = 1;
                }
                if (a.res_embed.scale && overlaps(a.res_embed.scale, 4, pj.y, ybytes(j))) bad = true;
            }
            if (bad) break;
        }
        if (a.ticket) { ph.argmax = 1; ph.argmax_out[0] = a.argmax_out[0]; ph.argmax_out[1] = a.argmax_out[1]; }
        if (i > 0) {
            const chain_phase & pj = out[(size_t) i - 1];
            if (ph.x_chain == ph.emb_chain) break;            // exactly one hand-off from the predecessor
            if (pj.argmax != ph.emb_chain) break;             // an arg-max is consumed by the very next phase (or ends the run)
        } else if (ph.x_chain || ph.emb_chain) break;
        if (ph.attn >= 0) attns.push_back(*a.attn);
        out.push_back(ph);
        res_from.push_back(rf);
    }
    int len = (int) out.size();
    // kept rows: a residual taken from the run must be the rows of the most recent phase that keeps any; cut the run where that fails
    for (;;) {
        for (int i = 0; i < len; i++) out[(size_t) i].save = 0;
        for (int i = 0; i < len; i++) if (out[(size_t) i].res == 1) out[(size_t) res_from[(size_t) i]].save = 1;
        int cut = len;
        for (int i = 0; i < len && cut == len; i++) {
            const chain_phase & ph = out[(size_t) i];
            if (ph.res == 1) {
                int src = -1;
                for (int j = i - 1; j >= 0; j--) if (out[(size_t) j].save) { src = j; break; }
                if (src != res_from[(size_t) i]) cut = i;
            }
            if ((ph.pair_F ? 2 : 1) * ph.rows_wg * ph.nb > CH_PMAX_OF(G) * CH_NCW * 8) cut = i;   // a phase's super-blocks must fit the registers
        }
        while (cut > 0 && out[(size_t) cut - 1].pair_F) cut--;   // a paired phase needs its consumer
        if (cut == len) break;
        len = cut;
    }
    out.resize((size_t) len);
    int na = 0;
    for (int i = 0; i < len; i++) {
        chain_phase & ph = out[(size_t) i];
        const bool next_x = i + 1 < len && out[(size_t) i + 1].x_chain;
        ph.n_pub = next_x ? (ph.pair_F ? (int) ph.pair_F : ph.M) : 0;
        if (ph.n_pub > CH_XF_MAX || (ph.n_pub & 1)) return chain_analyse(mv, i, G, out, attns);   // (cannot hand this vector on: end the run in front of it)
        if (ph.attn >= 0) na++;
    }
    attns.resize((size_t) na);
    return len;
}

// ---- the Q8_0 step program (hip_chain_nest80.h) --------------------------------------------------------------------------------------------------
// Does the run START with whole steps of the tts-shaped Depth transformer - [low-rank embedding], depformer_in + last, L x { in_proj, attention, out_proj,
// linear_in, linear_out (gated) }, linears[k] + arg-max - in Q8_0 at the instantiated widths? Returns the entries taken (0: no) and fills `d`.
struct n80_desc {
    int n_steps = 0, L = 0, KIN = 0, FF = 0, len = 0;
    std::vector<nest_ph> ph; std::vector<n80_at> at; std::vector<n80_st> st; std::vector<const char *> dsets;
    attn_args at0; int q_off = 0, k_off = 0, v_off = 0; const float * xin = nullptr; int64_t weight_bytes = 0;
};
static bool n80_is_mv(const mv_args & a, int64_t K, int64_t M, int pro) {
    return !a.special && a.wtype == GGML_TYPE_Q8_0 && a.ncols == 1 && a.K == K && a.M == M && a.prologue == pro && a.row_bytes == (K / 256) * 272 && !((uintptr_t) a.w & 15) &&
           !a.pair_F && !a.x_out && !a.out_scale && !a.out_act && !a.res_embed.table && !a.beta;
}
static int nest80_match(const mv_args * mv, int n, n80_desc & d) {
    static const int on = chain_env("MI355X_CHAIN_NEST80", 1);
    if (!on) return 0;
    int i = 0;
    const int32_t * prev_tok[2] = { nullptr, nullptr };
    while (i < n && d.n_steps < N80_STEPS_MAX) {
        const int i0 = i;
        n80_st st; memset(&st, 0, sizeof(st));
        auto fail = [&]() { i = i0; };
        if (mv[i].special == 2) {
            const lowrank_embed_args & lr = *mv[i].lr;
            if (lr.K != 128 || lr.M != 1024 || lr.w_row_bytes != 136 || ((uintptr_t) lr.w & 7) || d.n_steps == 0) { fail(); break; }
            if (!((const void *) lr.index == (const void *) prev_tok[0] || (const void *) lr.index == (const void *) prev_tok[1])) { fail(); break; }   // the token the previous step's arg-max wrote
            st.lr_table = lr.table; st.lr_row_bytes = lr.row_bytes; st.lr_n_rows = lr.n_rows; st.lr_type = lr.type; st.lr_index = lr.index; st.lr_w = lr.w;
            st.lr_w_row_bytes = lr.w_row_bytes; st.lr_out = lr.out; st.emb_chain = 1;
            i++;
        } else if (d.n_steps > 0) break;   // (later steps take their embedding from the chain)
        if (i >= n) { fail(); break; }
        const mv_args & din = mv[i];
        if (din.special || din.wtype != GGML_TYPE_Q8_0 || din.M != 1024 || din.prologue != MV_PLAIN || !din.residual || din.ticket || din.K % 256 || din.K > 4096) { fail(); break; }
        if (d.n_steps == 0) { d.KIN = (int) din.K; d.xin = din.x; }
        if (!n80_is_mv(din, d.KIN, 1024, MV_PLAIN) || din.x != d.xin) { fail(); break; }
        if (st.lr_table ? din.residual != st.lr_out : false) { fail(); break; }
        if (!st.lr_table) st.res_mem = din.residual;
        st.din_y = din.y;
        {
            size_t k = 0;
            while (k < d.dsets.size() && d.dsets[k] != din.w) k++;
            if (k == d.dsets.size()) { if (k >= N80_SETS_MAX) { fail(); break; } d.dsets.push_back(din.w); d.weight_bytes += din.M * din.row_bytes; }
            st.din_set = (int) k;
        }
        i++;
        const float * xprev = din.y;
        int l = 0;
        bool ok = true;
        std::vector<nest_ph> ph; std::vector<n80_at> ats;
        int64_t wb = 0;
        for (;; l++) {
            if (i >= n) { ok = false; break; }
            if (n80_is_mv(mv[i], 1024, 2048, MV_PLAIN) && mv[i].ticket && mv[i].x == xprev && !mv[i].residual) break;   // linears[k]
            if (i + 5 > n) { ok = false; break; }
            const mv_args & ip = mv[i], & as = mv[i + 1], & op = mv[i + 2], & li = mv[i + 3], & lo = mv[i + 4];
            if (!n80_is_mv(ip, 1024, 3072, MV_RMSNORM) || ip.x != xprev || ip.residual || ip.ticket) { ok = false; break; }
            if (as.special != 1) { ok = false; break; }
            const attn_args & at = *as.attn;
            if (at.T != 1 || at.D != 64 || at.H != 16 || at.C <= 8 || at.C > 64 || at.n_groups > 1 || at.out_ts < 0) { ok = false; break; }   // (attn_ring64_body's shapes: what the stand-alone launch runs there)
            auto inside = [&](const float * qq, int64_t hs) { return hs >= 0 && qq >= ip.y && qq + 15 * hs + 64 <= ip.y + 3072; };
            if (!inside(at.q, at.q_hs) || !inside(at.k, at.k_hs) || !inside(at.v, at.v_hs) || at.out_ts < 0) { ok = false; break; }
            if (d.n_steps == 0 && l == 0) { d.at0 = at; d.q_off = (int) (at.q - ip.y); d.k_off = (int) (at.k - ip.y); d.v_off = (int) (at.v - ip.y); }
            if ((int) (at.q - ip.y) != d.q_off || (int) (at.k - ip.y) != d.k_off || (int) (at.v - ip.y) != d.v_off || at.q_hs != d.at0.q_hs || at.k_hs != d.at0.k_hs ||
                at.v_hs != d.at0.v_hs || at.C != d.at0.C || at.scale != d.at0.scale || at.k_nb1 != d.at0.k_nb1 || at.k_nb2 != d.at0.k_nb2 || at.v_nb1 != d.at0.v_nb1 ||
                at.v_nb2 != d.at0.v_nb2 || (at.rot != nullptr) != (d.at0.rot != nullptr)) { ok = false; break; }
            if (!n80_is_mv(op, 1024, 1024, MV_PLAIN) || op.x != at.out || op.residual != xprev || op.ticket) { ok = false; break; }
            if (li.special || li.wtype != GGML_TYPE_Q8_0 || li.K != 1024 || li.prologue != MV_RMSNORM) { ok = false; break; }
            const int FF = (int) (li.M / 2);
            if (d.n_steps == 0 && l == 0) d.FF = FF;
            if (FF != d.FF || !n80_is_mv(li, 1024, 2 * (int64_t) FF, MV_RMSNORM) || li.x != op.y || li.residual || li.ticket) { ok = false; break; }
            if (!n80_is_mv(lo, FF, 1024, MV_GATE_SILU) || lo.x != li.y || lo.residual != op.y || lo.ticket) { ok = false; break; }
            ph.push_back({ ip.w, ip.alpha, ip.y, ip.eps, 0 }); ph.push_back({ op.w, nullptr, op.y, 0.f, 0 });
            ph.push_back({ li.w, li.alpha, li.y, li.eps, 0 }); ph.push_back({ lo.w, nullptr, lo.y, 0.f, 0 });
            ats.push_back({ at.kcache, at.vcache, at.rot, at.mask, at.index, at.out });
            wb += ip.M * ip.row_bytes + op.M * op.row_bytes + li.M * li.row_bytes + lo.M * lo.row_bytes;
            xprev = lo.y;
            i += 5;
        }
        if (!ok || l < 1 || (d.n_steps > 0 && l != d.L)) { fail(); break; }
        const mv_args & hd = mv[i];
        ph.push_back({ hd.w, nullptr, hd.y, 0.f, 0 });
        wb += hd.M * hd.row_bytes;
        st.argmax_out[0] = hd.argmax_out[0]; st.argmax_out[1] = hd.argmax_out[1];
        st.prev_out[0] = const_cast<int32_t *>(prev_tok[0]); st.prev_out[1] = const_cast<int32_t *>(prev_tok[1]);
        prev_tok[0] = hd.argmax_out[0]; prev_tok[1] = hd.argmax_out[1];
        i++;
        d.L = l;
        d.ph.insert(d.ph.end(), ph.begin(), ph.end()); d.at.insert(d.at.end(), ats.begin(), ats.end()); d.st.push_back(st);
        d.weight_bytes += wb;
        d.n_steps++;
    }
    if (d.n_steps == 0) return 0;
    if (!((d.KIN == 2048 && d.FF == 2048))) return 0;   // the instantiated widths (depth_nest80_kernel<KIN, FF>)
    d.len = i;
    return i;
}
static size_t nest80_attn_smem(const attn_args &) { return ATTN_RING64_SMEM; }
static size_t nest80_smem_bytes(const n80_desc & d) {
    return 16 * XBLK_BYTES + (size_t) (1024 + 16 + 1024) * 4 + 128 + 16 + sizeof(chain_ctl) + ((nest80_attn_smem(d.at0) + 15) & ~(size_t) 15) +
           d.ph.size() * sizeof(nest_ph) + d.at.size() * sizeof(n80_at) + d.st.size() * sizeof(n80_st) + N80_SETS_MAX * 8 + d.dsets.size() * 1024 * 4;
}
static size_t nest80_tables_bytes(const n80_desc & d) { return GGML_PAD(d.ph.size() * sizeof(nest_ph) + d.at.size() * sizeof(n80_at) + d.st.size() * sizeof(n80_st) + N80_SETS_MAX * 8, 256); }
static bool nest80_resident(const n80_desc & d, int usable_cus) {
    const size_t smem = nest80_smem_bytes(d);
    if (smem > 159 * 1024) return false;   // (the attention body brings a few hundred bytes of static LDS of its own)
    static bool granted = false;
    if (!granted) { HIP_CHECK(hipFuncSetAttribute((const void *) depth_nest80_kernel<2048, 2048>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); granted = true; }
    static const int force = chain_env("MI355X_CHAIN_GRID_FORCE", 0);
    int per_cu = 0;
    return force || (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) depth_nest80_kernel<2048, 2048>, CH_THREADS, smem) == hipSuccess &&
                     (long long) per_cu * usable_cus >= 256);
}

// ---- a Mimi transformer as one launch (hip_chain_mimi.h) -----------------------------------------------------------------------------------------------
// Does the run START with L x { LN + in_proj, attention (T = 2), out_proj * scale + residual, LN + linear1 + GELU, linear2 * scale + residual } in F32 at
// the codec's widths (512 / 8 heads of 64 / 2048)? Returns the entries taken (0: no).
struct mimi_desc { int L = 0, len = 0; bool first_partial = false; attn_args at_first; std::vector<mimi_mv> mv; std::vector<mimi_at> at; attn_args at0; int q_off = 0, k_off = 0, v_off = 0; const float * x_in = nullptr; int64_t x_cs = 0, weight_bytes = 0; };
static bool mimi_is_mv(const mv_args & a, int64_t K, int64_t M, int pro) {
    return !a.special && a.wtype == GGML_TYPE_F32 && a.ncols == 2 && a.K == K && a.M == M && a.prologue == pro && a.row_bytes == K * 4 && !((uintptr_t) a.w & 15) && a.y_cs == M &&
           !a.pair_F && !a.ticket && !a.res_embed.table && !a.argmax_out[0];
}
static int mimi_match(const mv_args * mv, int n, mimi_desc & d) {
    static const int on = chain_env("MI355X_CHAIN_MIMI", 1);
    if (!on) return 0;
    const int D = MIMI_D, F = MIMI_F;
    static const int verbose = chain_env("MI355X_CHAIN_VERBOSE", 0) > 1;
    int i = 0;
    const float * xprev = nullptr; int64_t xprev_cs = 0;
    // The first layer's in_proj may have been launched already: the planner emits the RoPE table (a node the first attention depends on) between that mat-vec
    // and its attention, which splits the run there. The program then starts at the attention phase of its first layer, q / k / v taken from memory.
    static const mv_args none = {};
    d.first_partial = n >= 4 && mv[0].special == 1;
    while (i + (d.first_partial && d.L == 0 ? 4 : 5) <= n && d.L < 16) {
        const bool part = d.first_partial && d.L == 0;
        const mv_args & ip = part ? none : mv[i], & as = mv[i + (part ? 0 : 1)], & op = mv[i + (part ? 1 : 2)], & l1 = mv[i + (part ? 2 : 3)], & l2 = mv[i + (part ? 3 : 4)];
        if (part) {
            const attn_args & at = *as.attn;
            if (at.T != 2 || at.D != 64 || at.H != 8 || at.C < 128 || at.C > 1024 || at.n_groups > 1 || at.out_ts != D || at.q_ts != 3 * D || at.k_ts != 3 * D || at.v_ts != 3 * D ||
                at.q_hs != 64 || at.k_hs != 64 || at.v_hs != 64) break;
            if (!mimi_is_mv(op, D, D, MV_PLAIN) || op.x != at.out || op.x_cs != D || !op.residual || op.out_act) break;
            if (!mimi_is_mv(l1, D, F, MV_LAYERNORM) || !l1.alpha || l1.x != op.y || l1.x_cs != D || l1.residual || l1.out_scale || l1.out_act != 1) break;
            if (!mimi_is_mv(l2, F, D, MV_PLAIN) || l2.x != l1.y || l2.x_cs != F || l2.residual != op.y || l2.r_cs != D || l2.out_act) break;
            d.at0 = at; d.q_off = d.k_off = d.v_off = 0; d.at_first = at;
            d.x_in = op.residual; d.x_cs = op.r_cs;
            d.mv.push_back({ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, { 0, 0, 0 } });
            d.mv.push_back({ op.w, nullptr, nullptr, op.out_scale, op.y, nullptr, 0.f, { 0, 0, 0 } });
            d.mv.push_back({ l1.w, l1.alpha, l1.beta, nullptr, l1.y, l1.x_out, l1.eps, { 0, 0, 0 } });
            d.mv.push_back({ l2.w, nullptr, nullptr, l2.out_scale, l2.y, nullptr, 0.f, { 0, 0, 0 } });
            d.at.push_back({ at.kcache, at.vcache, at.rot, at.mask, at.index, at.out });
            d.weight_bytes += op.M * op.row_bytes + l1.M * l1.row_bytes + l2.M * l2.row_bytes;
            xprev = l2.y; xprev_cs = D;
            d.L++;
            i += 4;
            continue;
        }
        if (!mimi_is_mv(ip, D, 3 * D, MV_LAYERNORM) || !ip.alpha || ip.residual || ip.out_scale || ip.out_act) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 1); break; }
        if (d.L == 0) { d.x_in = ip.x; d.x_cs = ip.x_cs; } else if (ip.x != xprev || ip.x_cs != xprev_cs) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 2); break; }
        if (as.special != 1) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 3); break; }
        const attn_args & at = *as.attn;
        if (at.T != 2 || at.D != 64 || at.H != 8 || at.C < 128 || at.C > 1024 || at.n_groups > 1 || at.out_ts != D) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 4); break; }   // (C >= 128, D = 64: the stand-alone launch runs 8 waves there)
        if (at.q_ts != 3 * D || at.k_ts != 3 * D || at.v_ts != 3 * D || at.q_hs != 64 || at.k_hs != 64 || at.v_hs != 64) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 5); break; }
        if (at.q < ip.y || at.k < ip.y || at.v < ip.y || at.q + 512 > ip.y + 3 * D || at.k + 512 > ip.y + 3 * D || at.v + 512 > ip.y + 3 * D) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 6); break; }
        if (d.L == 0 || (d.first_partial && d.L == 1)) { d.at0 = at; d.q_off = (int) (at.q - ip.y); d.k_off = (int) (at.k - ip.y); d.v_off = (int) (at.v - ip.y); }
        if ((int) (at.q - ip.y) != d.q_off || (int) (at.k - ip.y) != d.k_off || (int) (at.v - ip.y) != d.v_off || at.C != d.at0.C || at.scale != d.at0.scale ||
            at.k_nb1 != d.at0.k_nb1 || at.k_nb2 != d.at0.k_nb2 || at.v_nb1 != d.at0.v_nb1 || at.v_nb2 != d.at0.v_nb2 || (at.rot != nullptr) != (d.at0.rot != nullptr)) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 7); break; }
        const float * xin = d.L == 0 ? d.x_in : xprev; const int64_t xin_cs = d.L == 0 ? d.x_cs : xprev_cs;
        if (!mimi_is_mv(op, D, D, MV_PLAIN) || op.x != at.out || op.x_cs != D || op.residual != xin || op.r_cs != xin_cs || op.out_act) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 8); break; }
        if (!mimi_is_mv(l1, D, F, MV_LAYERNORM) || !l1.alpha || l1.x != op.y || l1.x_cs != D || l1.residual || l1.out_scale || l1.out_act != 1) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 9); break; }
        if (!mimi_is_mv(l2, F, D, MV_PLAIN) || l2.x != l1.y || l2.x_cs != F || l2.residual != op.y || l2.r_cs != D || l2.out_act) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 10); break; }
        if (d.L == 0 && d.x_cs < D) { if (verbose && d.L == 0) fprintf(stderr, "mimi_match: rule %d fails at layer 0\n", 11); break; }
        d.mv.push_back({ ip.w, ip.alpha, ip.beta, nullptr, ip.y, ip.x_out, ip.eps, { 0, 0, 0 } });
        d.mv.push_back({ op.w, nullptr, nullptr, op.out_scale, op.y, nullptr, 0.f, { 0, 0, 0 } });
        d.mv.push_back({ l1.w, l1.alpha, l1.beta, nullptr, l1.y, l1.x_out, l1.eps, { 0, 0, 0 } });
        d.mv.push_back({ l2.w, nullptr, nullptr, l2.out_scale, l2.y, nullptr, 0.f, { 0, 0, 0 } });
        d.at.push_back({ at.kcache, at.vcache, at.rot, at.mask, at.index, at.out });
        d.weight_bytes += ip.M * ip.row_bytes + op.M * op.row_bytes + l1.M * l1.row_bytes + l2.M * l2.row_bytes;
        xprev = l2.y; xprev_cs = D;
        d.L++;
        i += 5;
    }
    if (d.L < 2) return 0;
    d.len = i;
    return i;
}
static size_t mimi_attn_smem(const attn_args & at) { return ATTN_RING256_SMEM + (size_t) at.C * 4 + (size_t) at.T * at.D * 4 * 3 + (size_t) CH_NCW * 64 * 8 * 8 + 16 + (size_t) at.T * at.C * 4 + 64; }   // attn_smem_bytes, 8 waves
static size_t mimi_smem_bytes(const mimi_desc & d) {
    return (size_t) (2 * MIMI_F + 8) * 4 + 16 * 8 + sizeof(chain_ctl) + ((mimi_attn_smem(d.at0) + 15) & ~(size_t) 15) + d.mv.size() * sizeof(mimi_mv) + d.at.size() * sizeof(mimi_at);
}
static size_t mimi_tables_bytes(const mimi_desc & d) { return GGML_PAD(d.mv.size() * sizeof(mimi_mv) + d.at.size() * sizeof(mimi_at), 256); }
static bool mimi_resident(const mimi_desc & d, int usable_cus) {
    const size_t smem = mimi_smem_bytes(d);
    if (smem > 159 * 1024) return false;
    static bool granted = false;
    if (!granted) { HIP_CHECK(hipFuncSetAttribute((const void *) mimi_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); granted = true; }
    static const int force = chain_env("MI355X_CHAIN_GRID_FORCE", 0);
    int per_cu = 0;
    return force || (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) mimi_tr_kernel, CH_THREADS, smem) == hipSuccess && (long long) per_cu * usable_cus >= 256);
}

// ---- the Depth step program in the reference's sampling mode (tools/moshi-sts.cpp:106-107; moshi_sample_token, sampling.h:4-64) --------------------------
// With temp > 0 a sampler sits between linears[k] and the next step's embedding row (mv_args::special = 3 in launch order). The run
//     { depformer_in .. linears[k] (plain logits), sampler } x steps
// is the greedy run - linears[k] with an arg-max epilogue whose token the next step embeds - up to the rule that picks the token, so it is analysed AS that
// run: the sampler entries are taken out, each logits mat-vec in front of one gets the sampler's token outputs as its arg-max outputs, and chain_analyse /
// nest_build do the rest. Only the step program implements the rule (hip_chain_nest.h, head_argmax = 2): there is no descriptor-driven fall-back, so such a
// run is accepted only where nest_build succeeds; otherwise the samplers keep their launches and cut the run into per-step chains as before.
struct smp_strip { std::vector<mv_args> mv; std::vector<const sample_args *> smp; std::vector<int> end_of; bool any = false; };
static unsigned g_sampled_ticket;   // (never dereferenced: marks a logits mat-vec whose token a sampler picks)
static void sampled_strip(const mv_args * mv, int n, smp_strip & o) {
    for (int i = 0; i < n; i++) {
        const mv_args & a = mv[i];
        if (!a.special) { o.mv.push_back(a); o.smp.push_back(nullptr); o.end_of.push_back(i + 1); continue; }
        if (a.special != 3 || o.mv.empty() || o.smp.back() || !a.smp) break;
        mv_args & hd = o.mv.back();
        const sample_args & sa = *a.smp;
        if (hd.ticket || hd.argmax_out[0] || hd.argmax_out[1] || sa.logits != hd.y || sa.n != hd.M || sa.n != 2048 || sa.k < 1 || sa.k > 256 || !sa.noise || !sa.out) break;
        hd.ticket = &g_sampled_ticket; hd.argmax_out[0] = sa.out; hd.argmax_out[1] = sa.out2;
        o.smp.back() = a.smp; o.end_of.back() = i + 1; o.any = true;
    }
}
static void chain_fill(chain_plan * c);
static bool nest_build(chain_plan * c, char * tables_dev, char * din_dev, int usable_cus);
// entries of the run a sampled step program takes (0: none); *st (optional) receives the stripped run cut to exactly that
static int nest_sampled_try(const mv_args * mv, int n, int usable_cus, smp_strip * st) {
    if (n < 7 || mv[0].special || chain_grid_for(usable_cus) != 256) return 0;
    smp_strip o;
    sampled_strip(mv, n, o);
    if (!o.any) return 0;
    chain_plan tmp;
    tmp.grid = 256; tmp.smem = 0;
    memset(&tmp.P, 0, sizeof(tmp.P));
    int len = chain_analyse(o.mv.data(), (int) o.mv.size(), 256, tmp.phases, tmp.attns);
    while (len > 0 && !o.smp[(size_t) len - 1]) len--;          // the run ends behind a sampled head
    if (len <= 0) return 0;
    if (chain_analyse(o.mv.data(), len, 256, tmp.phases, tmp.attns) != len) return 0;
    for (int i = 0; i < len; i++) if ((tmp.phases[(size_t) i].argmax != 0) != (o.smp[(size_t) i] != nullptr)) return 0;   // every token of the run is a sampled one
    chain_fill(&tmp);
    if (!nest_build(&tmp, nullptr, nullptr, usable_cus)) return 0;
    const int taken = o.end_of[(size_t) len - 1];
    if (st) { o.mv.resize((size_t) len); o.smp.resize((size_t) len); o.end_of.resize((size_t) len); *st = std::move(o); }
    return taken;
}

int k_chain_accept(const mv_args * mv, int n, int usable_cus, bool allow_step_program) {
    if (n >= 10 && ((!mv[0].special && mv[0].wtype == GGML_TYPE_F32 && mv[0].ncols == 2) || (mv[0].special == 1 && mv[0].attn->T == 2)) && chain_grid_for(usable_cus) == 256) {
        mimi_desc d;
        const int len = mimi_match(mv, n, d);
        if (len > 0 && mimi_resident(d, usable_cus)) return len;
    }
    if (n > 0 && (mv[0].special == 2 || (!mv[0].special && mv[0].wtype == GGML_TYPE_Q8_0)) && chain_grid_for(usable_cus) == 256) {
        n80_desc d;
        const int len = nest80_match(mv, n, d);
        if (len > 0 && nest80_resident(d, usable_cus)) return len;
    }
    if (n > 0 && mv[0].special) return 0;
    if (allow_step_program) { const int len = nest_sampled_try(mv, n, usable_cus, nullptr); if (len > 0) return len; }
    static const int min_len = chain_env("MI355X_CHAIN_MIN", 4);
    if (n < min_len) return 0;
    const int G = chain_grid_for(usable_cus);
    if (G <= 0) return 0;   // the grid cannot be resident on this device / stream: one launch per mat-vec
    std::vector<chain_phase> ph; std::vector<attn_args> at;
    const int len = chain_analyse(mv, n, G, ph, at);
    // Q8_0 runs pay off only when long: the four-phase runs between two attention LAUNCHES of the tts-shaped Depth transformer (32-slot ring: its attention
    // is not recomputed inside out_proj) measured slower as chains than as launches - 198 vs 222 frames/s, the resident grid also starves the codec stream
    // beside it (profiles/r05_ab_q8_0_chain_tts.txt). Whole Depth steps (short ring, attention inside the chain) are taken.
    static const int min_q80 = chain_env("MI355X_CHAIN_MIN_Q80", 12);
    bool any_q80 = false;
    for (int i = 0; i < len; i++) any_q80 = any_q80 || ph[(size_t) i].fmt == 1;
    if (any_q80 && len < min_q80) return 0;
    return len >= min_len ? len : 0;
}

static size_t chain_tables_bytes(int n, int) { return GGML_PAD((size_t) n * sizeof(chain_phase), 256); }
static size_t chain_state_bytes(int grid) { return 256 + 2 * (size_t) CH_XF_MAX * 8 + 2 * 2 * (size_t) grid * 8; }
// what a step program adds: its compact tables and the [steps][1024] granules of the hoisted depformer_in products (upper bounds: reserved for every run)
static size_t nest_tables_bytes(int n) { return GGML_PAD((size_t) n * (sizeof(nest_ph) + sizeof(nest_at)) + NEST_STEPS_MAX * sizeof(nest_st), 256); }
static size_t nest_din_bytes() { return (size_t) NEST_STEPS_MAX * 1024 * 8; }
size_t k_chain_ws_size(const mv_args * mv, int n, int usable_cus, bool allow_step_program) {
    if (n >= 10 && ((!mv[0].special && mv[0].wtype == GGML_TYPE_F32 && mv[0].ncols == 2) || (mv[0].special == 1 && mv[0].attn->T == 2))) {
        mimi_desc d;
        if (mimi_match(mv, n, d) == n) return mimi_tables_bytes(d) + chain_state_bytes(256);
    }
    {
        n80_desc d;
        if (n > 0 && (mv[0].special == 2 || (!mv[0].special && mv[0].wtype == GGML_TYPE_Q8_0)) && nest80_match(mv, n, d) == n)
            return nest80_tables_bytes(d) + chain_state_bytes(256) + (size_t) N80_SETS_MAX * 1024 * 8;
    }
    std::vector<chain_phase> ph; std::vector<attn_args> at;
    const int G = chain_grid_for(usable_cus);
    smp_strip st;
    if (allow_step_program && nest_sampled_try(mv, n, usable_cus, &st) == n) { mv = st.mv.data(); n = (int) st.mv.size(); }
    const int len = chain_analyse(mv, n, G, ph, at);
    return chain_tables_bytes(len, (int) at.size()) + chain_state_bytes(G) + nest_tables_bytes(len) + nest_din_bytes();
}

// which compile-time shape (1..6, see shape_din .. shape_head) a phase matches exactly, 0: none - its shape is read from the descriptor
template <class SH> static bool shape_is(const chain_phase & ph, int grid) {
    const int rows_total = SH::PAIR ? SH::PAIR : SH::M;
    return ph.K == SH::K && ph.M == SH::M && ph.pair_F == SH::PAIR && ph.prologue == SH::PRO && ph.x_chain == SH::XCH && ph.res == SH::RES &&
           (ph.emb.table != nullptr) == (SH::EMB != 0) && ph.save <= SH::SAVE && ph.argmax == SH::AM && ph.n_pub == SH::PUB &&
           rows_total % grid == 0 && ph.rows_wg == rows_total / grid && ph.n_in == (SH::XCH ? (SH::PRO == MV_ATTN ? 3 * SH::K : SH::K) : ph.n_in) &&
           ph.row_bytes == (SH::K / 256) * 144 && ph.fmt == 0;
}
static int chain_shape_kind(const chain_phase & ph, int grid) {
    static const int on = chain_env("MI355X_CHAIN_SHAPES", 0x7e);   // bit k: phase kind k may use its compile-time shape
    if (!on || !(grid == 64 || grid == 128 || grid == 256)) return 0;
    int k = 0;
    if (shape_is<shape_din>(ph, grid)) k = 1;
    else if (shape_is<shape_inproj>(ph, grid)) k = 2;
    else if (shape_is<shape_outproj>(ph, grid)) k = 3;
    else if (shape_is<shape_linin>(ph, grid)) k = 4;
    else if (shape_is<shape_linout>(ph, grid)) k = 5;
    else if (shape_is<shape_head>(ph, grid)) k = 6;
    return (on >> k) & 1 ? k : 0;
}

static size_t nest_smem_bytes(int n_steps, int L) {
    const int n_ph = n_steps * (4 * L + 1), n_at = n_steps * L;
    return 16 * XBLK_BYTES + (size_t) (NEST_XF + NEST_PART + 16 + 1024 + CH_NCW * CH_ATTW) * 4 + sizeof(chain_ctl) +
           (size_t) n_ph * sizeof(nest_ph) + (size_t) n_at * sizeof(nest_at) + (size_t) n_steps * sizeof(nest_st) + (size_t) n_steps * 1024 * 4;
}
// Is the run (kinds assigned) a step program - steps x { din, L x { in_proj, out_proj, linear_in, linear_out }, head } at the compile-time shapes, one
// transformer_out, one attention shape? Then fill c->NP / c->nest_tables (device addresses inside `tables_dev` / `din_dev`).
static bool nest_no(int why) {
    if (chain_env("MI355X_CHAIN_VERBOSE", 0)) fprintf(stderr, "chain engine: not a step program (rule %d of nest_build)\n", why);
    return false;
}
static bool nest_build(chain_plan * c, char * tables_dev, char * din_dev, int usable_cus) {
    static const int on = chain_env("MI355X_CHAIN_NEST", 1);
    const int n = (int) c->phases.size();
    if (chain_env("MI355X_CHAIN_VERBOSE", 0) > 1)
        for (int i = 0; i < n && i < 60; i++) {
            const chain_phase & ph = c->phases[(size_t) i];
            fprintf(stderr, "  phase %3d kind %d K %d M %d pair %lld pro %d xch %d res %d emb %d save %d am %d pub %d n_in %d rows_wg %d\n", i, ph.kind, ph.K, ph.M, (long long) ph.pair_F, ph.prologue,
                    ph.x_chain, ph.res, ph.emb.table != nullptr, ph.save, ph.argmax, ph.n_pub, ph.n_in, ph.rows_wg);
        }
    if (!on || c->grid != 256 || n < 6 || c->phases[0].kind != 1) return nest_no(1);
    int per = 1;
    while (per < n && c->phases[(size_t) per].kind != 1) per++;
    if ((per - 2) % 4 != 0 || per < 6 || n % per != 0) return nest_no(2);
    const int L = (per - 2) / 4, S = n / per;
    if (S > NEST_STEPS_MAX) return nest_no(3);
    const chain_phase & d0 = c->phases[0];
    const attn_args * a0 = nullptr;
    auto fits = [](int64_t v) { return v >= 0 && v < (1ll << 30); };
    for (int s = 0; s < S; s++) {
        const chain_phase * st = &c->phases[(size_t) s * per];
        if (st[0].kind != 1 || st[0].x != d0.x || !st[0].save || st[0].res || (s == 0 && st[0].emb_chain)) return nest_no(4);
        for (int l = 0; l < L; l++)
            for (int i = 0; i < 4; i++) if (st[1 + 4 * l + i].kind != 2 + i) return nest_no(5);
        {   // linears[k]: with the arg-max epilogue (greedy), or as plain logits for a sampler launch behind the run (temp > 0: one step per run)
            const chain_phase & hd = st[per - 1];
            const bool plain_head = hd.kind == 0 && hd.K == 1024 && hd.M == 2048 && hd.prologue == MV_PLAIN && hd.x_chain && !hd.res && !hd.emb.table && !hd.argmax &&
                                    hd.n_pub == 0 && !hd.pair_F && hd.rows_wg == 8 && hd.row_bytes == 4 * 144 && S == 1;
            if (hd.kind != 6 && !plain_head) return nest_no(6);
        }
        if (s > 0 && !st[0].emb_chain) return nest_no(7);   // (a head's arg-max inside the run is consumed by the next step)
        for (int l = 0; l < L; l++) {
            const chain_phase & op = st[1 + 4 * l + 1], & lo = st[1 + 4 * l + 3];
            if (!op.save || op.res != 1 || lo.res != 1) return nest_no(8);
            if (l + 1 < L && !lo.save) return nest_no(9);
            const attn_args & at = op.at;
            if (!a0) a0 = &at;
            if (at.H != a0->H || at.D != a0->D || at.C != a0->C || at.T != 1 || at.scale != a0->scale || at.q_hs != a0->q_hs || at.k_hs != a0->k_hs || at.v_hs != a0->v_hs ||
                at.k_nb1 != a0->k_nb1 || at.k_nb2 != a0->k_nb2 || at.v_nb1 != a0->v_nb1 || at.v_nb2 != a0->v_nb2) return nest_no(10);
            const chain_phase & o0 = c->phases[2];
            if (op.q_off != o0.q_off || op.k_off != o0.k_off || op.v_off != o0.v_off) return nest_no(11);
        }
    }
    if (!a0 || !fits(a0->q_hs) || !fits(a0->k_hs) || !fits(a0->v_hs) || !fits(a0->k_nb1) || !fits(a0->k_nb2) || !fits(a0->v_nb1) || !fits(a0->v_nb2)) return nest_no(12);
    {   // nest_attn_pair's layout: 16 heads of 64 straight out of the in_proj vector (q | k | v), no rotary embedding (depformer_pos_emb "none", lm_default.h:97-103),
        // a ring of <= 8 slots whose rows are whole dwords
        const chain_phase & o0 = c->phases[2];
        if (a0->H != 2 * CH_NCW || a0->D != 64 || a0->C < 1 || a0->C > 8 || a0->q_hs != 64 || a0->k_hs != 64 || a0->v_hs != 64 || o0.q_off != 0 || o0.k_off != 1024 ||
            o0.v_off != 2048 || a0->k_nb1 % 4 || a0->k_nb2 % 4 || a0->v_nb1 % 4 || a0->v_nb2 % 4 || (int64_t) a0->H * a0->k_nb2 >= (1ll << 30) || (int64_t) a0->H * a0->v_nb2 >= (1ll << 30)) return nest_no(15);
        for (int s = 0; s < S; s++) for (int l = 0; l < L; l++) if (c->phases[(size_t) s * per + 1 + 4 * l + 1].at.rot) return nest_no(16);
    }
    c->nest_smem = nest_smem_bytes(S, L);
    if (c->nest_smem > 160 * 1024) return nest_no(13);
    {   // the whole grid must be resident with THIS kernel's footprint too
        static bool granted = false;
        if (!granted) {
            HIP_CHECK(hipFuncSetAttribute((const void *) depth_nest_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void *) depth_nest_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            granted = true;
        }
        static const int force = chain_env("MI355X_CHAIN_GRID_FORCE", 0);
        int per_cu = 0;
        if (!force && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) depth_nest_kernel<256, true>, CH_THREADS, c->nest_smem) != hipSuccess ||   // (the larger of the two instances)
                       (long long) per_cu * usable_cus < 256)) return nest_no(14);
    }
    const int n_ph = S * (4 * L + 1), n_at = S * L;
    c->nest_tables.assign((size_t) n_ph * sizeof(nest_ph) + (size_t) n_at * sizeof(nest_at) + (size_t) S * sizeof(nest_st), 0);
    nest_ph * tp = (nest_ph *) c->nest_tables.data();
    nest_at * ta = (nest_at *) (tp + n_ph);
    nest_st * ts = (nest_st *) (ta + n_at);
    for (int s = 0; s < S; s++) {
        const chain_phase * st = &c->phases[(size_t) s * per];
        nest_st & r = ts[s];
        r.din_w = st[0].w; r.din_y = st[0].y; r.emb = st[0].emb; r.emb_chain = st[0].emb_chain;
        r.prev_out[0] = st[0].prev_out[0]; r.prev_out[1] = st[0].prev_out[1];
        r.argmax_out[0] = st[per - 1].argmax_out[0]; r.argmax_out[1] = st[per - 1].argmax_out[1];
        for (int i = 1; i < per; i++) {
            nest_ph & q = tp[s * (4 * L + 1) + i - 1];
            q.w = st[i].w; q.alpha = st[i].alpha; q.y = st[i].y; q.eps = st[i].eps;
        }
        for (int l = 0; l < L; l++) {
            const attn_args & at = st[1 + 4 * l + 1].at;
            nest_at & q = ta[s * L + l];
            q.kcache = at.kcache; q.vcache = at.vcache; q.rot = at.rot; q.mask = at.mask; q.index = at.index;
        }
    }
    nest_params & N = c->NP;
    memset(&N, 0, sizeof(N));
    N.P = c->P;
    N.tables = (const u32x4 *) tables_dev;
    N.n_steps = S; N.n_layers = L;
    N.head_argmax = c->phases[(size_t) per - 1].argmax;
    N.din_buf = (u64 *) din_dev; N.din_x = d0.x;
    const chain_phase & o0 = c->phases[2];
    N.q_off = o0.q_off; N.k_off = o0.k_off; N.v_off = o0.v_off;
    N.q_hs = (int) a0->q_hs; N.k_hs = (int) a0->k_hs; N.v_hs = (int) a0->v_hs;
    N.k_nb1 = (int) a0->k_nb1; N.k_nb2 = (int) a0->k_nb2; N.v_nb1 = (int) a0->v_nb1; N.v_nb2 = (int) a0->v_nb2; N.C = a0->C; N.scale = a0->scale;
    {   // MI355X_NEST_DELAY="a,b,c,d,e": s_sleep units before the first poll of in_proj / out_proj / linear_in / linear_out / linears[k]
        static const int dflt[5] = { 20, 0, 20, 20, 20 };   // tests/microbench/nest_delay_sweep.sh: LM step 2 310 us with no delay, 2 120 - 2 135 with 20 - 24 units
        for (int i = 0; i < 5; i++) N.delay[i] = dflt[i];
        if (const char * e = getenv("MI355X_NEST_DELAY")) {
            int d[5]; const int got = sscanf(e, "%d,%d,%d,%d,%d", &d[0], &d[1], &d[2], &d[3], &d[4]);
            for (int i = 0; i < got && i < 5; i++) N.delay[i] = d[i] < 0 ? 0 : d[i] > 200 ? 200 : d[i];
        }
    }
    if (chain_env("MI355X_CHAIN_VERBOSE", 0)) fprintf(stderr, "chain engine: step program, %d steps x %d layers, %zu bytes of LDS\n", S, L, c->nest_smem);
    return true;
}

chain_plan * k_chain_create(hipStream_t s, const mv_args * mv, int n, void * ws, unsigned * err, int usable_cus, bool allow_step_program) {
    chain_plan * c = new chain_plan;
    c->grid = chain_grid_for(usable_cus);
    if (n >= 10 && ((!mv[0].special && mv[0].wtype == GGML_TYPE_F32 && mv[0].ncols == 2) || (mv[0].special == 1 && mv[0].attn->T == 2)) && c->grid == 256) {
        mimi_desc d;
        if (mimi_match(mv, n, d) == n) {
            char * base = (char *) ws;
            char * state = base + mimi_tables_bytes(d);
            std::vector<char> tab(mimi_tables_bytes(d), 0);
            memcpy(tab.data(), d.mv.data(), d.mv.size() * sizeof(mimi_mv));
            memcpy(tab.data() + d.mv.size() * sizeof(mimi_mv), d.at.data(), d.at.size() * sizeof(mimi_at));
            c->nest_tables.swap(tab);
            HIP_CHECK(hipMemcpyAsync(base, c->nest_tables.data(), c->nest_tables.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemsetAsync(state, 0, chain_state_bytes(256), s));
            mimi_params & N = c->MM;
            memset(&N, 0, sizeof(N));
            N.P.phases = nullptr; N.P.n_phases = d.len;
            N.P.launch_seq = (unsigned *) state; N.P.gbuf = (u64 *) (state + 256); N.P.cand = N.P.gbuf + 2 * CH_XF_MAX; N.P.err = err; N.P.delay = 0;
            N.tables = (const u32x4 *) base;
            N.n_layers = d.L; N.x_in = d.x_in; N.x_cs = d.x_cs;
            N.at = d.at0; N.at.q = N.at.k = N.at.v = nullptr;
            N.first_partial = d.first_partial ? 1 : 0;
            if (d.first_partial) { N.q0 = d.at_first.q; N.k0 = d.at_first.k; N.v0 = d.at_first.v; }
            N.q_off = d.q_off; N.k_off = d.k_off; N.v_off = d.v_off;
            N.attn_smem = mimi_attn_smem(d.at0);
            N.ring256 = chain_env("MI355X_ATTN_RING256", 1) && d.at0.C > 64 && d.at0.C <= 256 ? 1 : 0;   // (the stand-alone launch's choice, k_attn_decode)
            {
                static const int dflt[2] = { 30, 200 };   // tests/microbench/mimi_delay_sweep.sh (profiles/r05_mimi_delay_sweep.txt): flat from 20 / 100 up
                N.delay[0] = dflt[0]; N.delay[1] = dflt[1];
                if (const char * e = getenv("MI355X_MIMI_DELAY")) { int v[2]; const int got = sscanf(e, "%d,%d", &v[0], &v[1]); for (int i = 0; i < got && i < 2; i++) N.delay[i] = v[i] < 0 ? 0 : v[i] > 600 ? 600 : v[i]; }
            }
            c->P = N.P;
            c->smem = mimi_smem_bytes(d);
            c->mimi = true; c->n80_weight_bytes = d.weight_bytes; c->n80_len = d.len;
            if (chain_env("MI355X_CHAIN_VERBOSE", 0)) fprintf(stderr, "chain engine: Mimi transformer program, %d layers, ring of %d, %zu bytes of LDS, %d plan steps\n", d.L, d.at0.C, c->smem, d.len);
            return c;
        }
    }
    {
        n80_desc d;
        if (n > 0 && (mv[0].special == 2 || (!mv[0].special && mv[0].wtype == GGML_TYPE_Q8_0)) && c->grid == 256 && nest80_match(mv, n, d) == n) {
            // the Q8_0 step program: tables | hand-off state | din granules
            char * base = (char *) ws;
            char * state = base + nest80_tables_bytes(d);
            char * din = state + chain_state_bytes(256);
            std::vector<char> tab(nest80_tables_bytes(d), 0);
            char * t = tab.data();
            memcpy(t, d.ph.data(), d.ph.size() * sizeof(nest_ph)); t += d.ph.size() * sizeof(nest_ph);
            memcpy(t, d.at.data(), d.at.size() * sizeof(n80_at)); t += d.at.size() * sizeof(n80_at);
            memcpy(t, d.st.data(), d.st.size() * sizeof(n80_st)); t += d.st.size() * sizeof(n80_st);
            memcpy(t, d.dsets.data(), d.dsets.size() * 8);
            c->nest_tables.swap(tab);
            HIP_CHECK(hipMemcpyAsync(base, c->nest_tables.data(), c->nest_tables.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemsetAsync(state, 0, chain_state_bytes(256) + (size_t) N80_SETS_MAX * 1024 * 8, s));
            nest80_params & N = c->N80;
            memset(&N, 0, sizeof(N));
            N.P.phases = nullptr; N.P.n_phases = d.len;
            N.P.launch_seq = (unsigned *) state; N.P.gbuf = (u64 *) (state + 256); N.P.cand = N.P.gbuf + 2 * CH_XF_MAX; N.P.err = err; N.P.delay = 0;
            N.tables = (const u32x4 *) base;
            N.n_steps = d.n_steps; N.n_layers = d.L; N.n_sets = (int) d.dsets.size();
            N.din_buf = (u64 *) din; N.din_x = d.xin;
            N.at = d.at0; N.at.q = N.at.k = N.at.v = nullptr; N.at.n_groups = 0; N.at.write_only = 0; N.at.row_split = 0;
            N.q_off = d.q_off; N.k_off = d.k_off; N.v_off = d.v_off;
            N.attn_smem = nest80_attn_smem(d.at0);
            {   // MI355X_NEST80_DELAY="a,b,c": first-poll delays (s_sleep units) of the mat-vec phases / of out_proj on non-owner workgroups / of the head owners' granule poll
                static const int dflt[4] = { 20, 60, 0, 0 };   // tests/microbench/nest80_delay_sweep.sh (gpurun_out/r05_nest80_delay_sweep.txt): flat between 20 and 30 / 40 and 100; the owners poll at once
                for (int i = 0; i < 4; i++) N.delay[i] = dflt[i];
                if (const char * e = getenv("MI355X_NEST80_DELAY")) {
                    int v[3]; const int got = sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
                    for (int i = 0; i < got && i < 3; i++) N.delay[i] = v[i] < 0 ? 0 : v[i] > 400 ? 400 : v[i];
                }
            }
            c->P = N.P;
            c->smem = nest80_smem_bytes(d);
            c->nest80 = true; c->n80_weight_bytes = d.weight_bytes; c->n80_len = d.len;
            if (chain_env("MI355X_CHAIN_VERBOSE", 0))
                fprintf(stderr, "chain engine: Q8_0 step program, %d steps x %d layers, %d depformer_in sets, ring of %d, %zu bytes of LDS, %d plan steps\n", d.n_steps, d.L, N.n_sets,
                        d.at0.C, c->smem, d.len);
            return c;
        }
    }
    GGML_ASSERT(c->grid > 0 && n <= CH_MAX_PHASES);
    smp_strip sampled;   // (sampling mode: the run as the greedy run it is analysed as, see nest_sampled_try)
    const bool is_sampled = allow_step_program && nest_sampled_try(mv, n, usable_cus, &sampled) == n;
    if (is_sampled) { mv = sampled.mv.data(); n = (int) sampled.mv.size(); }
    const int len = chain_analyse(mv, n, c->grid, c->phases, c->attns);
    GGML_ASSERT(len == n && "k_chain_create: pass exactly the run k_chain_accept took");
    char * base = (char *) ws;
    chain_phase * d_ph = (chain_phase *) base;
    char * state = base + chain_tables_bytes(n, (int) c->attns.size());
    char * nest_tab = state + chain_state_bytes(c->grid);
    char * nest_din = nest_tab + nest_tables_bytes(n);
    chain_fill(c);
    // the table is uploaded from the plan's own vector (pageable memory: the runtime stages it before returning); the host copy lives as long as the plan
    HIP_CHECK(hipMemcpyAsync(d_ph, c->phases.data(), (size_t) n * sizeof(chain_phase), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(state, 0, chain_state_bytes(c->grid) + nest_tables_bytes(n) + nest_din_bytes(), s));
    c->P.phases = d_ph; c->P.n_phases = n;
    c->P.launch_seq = (unsigned *) state;
    c->P.gbuf = (u64 *) (state + 256);
    c->P.cand = c->P.gbuf + 2 * CH_XF_MAX;
    c->P.err = err;
    {
        static const int delay = chain_env("MI355X_CHAIN_DELAY", 20);
        c->P.delay = delay < 0 ? 0 : delay > 200 ? 200 : delay;
    }
    c->smem = chain_smem();
    GGML_ASSERT(c->smem <= 160 * 1024);
    c->nest = allow_step_program && nest_build(c, nest_tab, nest_din, usable_cus);
    if (is_sampled) {
        GGML_ASSERT(c->nest && "k_chain_create: a sampled run is only ever accepted as a step program");
        // the samplers' arguments, step by step: the token rule of the linears[k] phases (head_argmax = 2)
        const int per = n / c->NP.n_steps;
        nest_st * ts = (nest_st *) (c->nest_tables.data() + (size_t) c->NP.n_steps * ((4 * c->NP.n_layers + 1) * sizeof(nest_ph) + c->NP.n_layers * sizeof(nest_at)));
        for (int st = 0; st < c->NP.n_steps; st++) {
            const sample_args * sa = sampled.smp[(size_t) (st + 1) * per - 1];
            GGML_ASSERT(sa);
            ts[st].noise = sa->noise; ts[st].smp_scale = sa->scale; ts[st].smp_k = sa->k;
        }
        c->NP.head_argmax = 2;
    }
    if (c->nest) HIP_CHECK(hipMemcpyAsync(nest_tab, c->nest_tables.data(), c->nest_tables.size(), hipMemcpyHostToDevice, s));
    return c;
}
// per-phase fields that follow from the analysed run: the attention a phase recomputes, what it is handed, its compile-time shape, where a chained token is stored
static void chain_fill(chain_plan * c) {
    int ai = 0;
    for (size_t i = 0; i < c->phases.size(); i++) {
        chain_phase & ph = c->phases[i];
        if (ph.attn >= 0) ph.at = c->attns[(size_t) ai++];
        ph.n_in = i > 0 ? c->phases[i - 1].n_pub : 0;
        ph.kind = chain_shape_kind(ph, c->grid);
        if (ph.emb_chain) { ph.prev_out[0] = c->phases[i - 1].argmax_out[0]; ph.prev_out[1] = c->phases[i - 1].argmax_out[1]; }
    }
}
void k_chain_free(chain_plan * c) { delete c; }
int k_chain_length(const chain_plan * c) { return c->P.n_phases; }
int64_t k_chain_weight_bytes(const chain_plan * c) { if (c->nest80 || c->mimi) return c->n80_weight_bytes; int64_t b = 0; for (auto & ph : c->phases) b += (int64_t) ph.M * ph.row_bytes; return b; }

bool k_chain_is_step_program(const chain_plan * c) { return c->nest || c->nest80 || c->mimi; }

void k_chain_launch(hipStream_t s, const chain_plan * c) {
    if (c->mimi) { mimi_tr_kernel<<<256, CH_THREADS, c->smem, s>>>(c->MM); return; }
    if (c->nest80) { depth_nest80_kernel<2048, 2048><<<256, CH_THREADS, c->smem, s>>>(c->N80); return; }
    if (c->nest) {
        if (c->NP.head_argmax == 2) depth_nest_kernel<256, true><<<c->grid, CH_THREADS, c->nest_smem, s>>>(c->NP);
        else depth_nest_kernel<256><<<c->grid, CH_THREADS, c->nest_smem, s>>>(c->NP);
        return;
    }
    // (the template argument only matters to phases with a compile-time shape; any other grid runs every phase from its descriptor)
    if (c->grid == 256) matvec_chain_kernel<256><<<c->grid, CH_THREADS, c->smem, s>>>(c->P);
    else if (c->grid == 128) matvec_chain_kernel<128><<<c->grid, CH_THREADS, c->smem, s>>>(c->P);
    else matvec_chain_kernel<64><<<c->grid, CH_THREADS, c->smem, s>>>(c->P);
}

// ---- RVQ encode: the levels of one residual stack as one launch -----------------------------------------------------------------------------------------------
// vq_level_kernel's arithmetic level by level (score = num[c] / (float(sum_j (double) ((e_cj - x_j)^2)) + add_c), the LAST maximum wins; x <- x - e_best), with the
// level-to-level dependency kept inside the launch: every workgroup holds the residual (lane l: elements 4 l .. 4 l + 3, the same in every wave), scores its 16
// centroids, publishes ONE candidate as two {tag, value} granules, merges all candidates itself and fetches the winning centroid row - no second hand-off, no
// launch boundary. The next level's centroid rows are requested before the candidates are polled (they do not depend on the chain).
#define VQC_CPW 4
#define VQC_LEVELS_MAX 32
struct vqc_level { const char * emb; long long emb_row_bytes; const float * add_c; const float * num; float * resid_out; float * idx_f; int32_t * idx_i; long long pad; };
static_assert(sizeof(vqc_level) == 64, "level records are read by 16-byte lanes");
struct vqc_params { const vqc_level * levels; int n_levels, NC; const char * resid; long long resid_stride; u64 * cand; unsigned * launch_seq; unsigned * err; int delay; };
__global__ void __launch_bounds__(256) vq_chain_kernel(vqc_params P) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_code;
    __shared__ unsigned s_failed;
    __shared__ __attribute__((aligned(16))) vqc_level s_lv[VQC_LEVELS_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grid = (int) gridDim.x, wg = (int) blockIdx.x;
    for (int i = tid; i < P.n_levels * 4; i += 256) ((u32x4 *) s_lv)[i] = ((const GLOBAL_AS u32x4 *) P.levels)[i];
    if (tid == 0) s_failed = 0u;
    const unsigned launch = *gp(P.launch_seq);
    const unsigned tag_base = launch << 12;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = *(const GLOBAL_AS float *) (gp(P.resid) + (long long) (lane * 4 + j) * P.resid_stride);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t cb = make_rsrc(P.cand, 2u * 2u * (unsigned) grid * 8u);
    const int c0 = (wg * 4 + wave) * VQC_CPW;
    f32x4 e[VQC_CPW];
    auto request = [&](int l) {
        const char * emb = *(const char * const *) &s_lv[l].emb; const long long rb = *(const long long *) &s_lv[l].emb_row_bytes;
#pragma unroll
        for (int u = 0; u < VQC_CPW; u++) {
            const int c = c0 + u < P.NC ? c0 + u : P.NC - 1;
            e[u] = *(const GLOBAL_AS f32x4 *) (gp(emb) + (long long) c * rb + lane * 16);
        }
    };
    request(0);
#pragma unroll 1
    for (int l = 0; l < P.n_levels; l++) {
        const vqc_level lv = s_lv[l];
        const float addc = gp(lv.add_c)[0];
        float best = -INFINITY; int bi = -1;
#pragma unroll
        for (int u = 0; u < VQC_CPW; u++) {
            double acc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { const float d = e[u][j] - x[j]; acc += (double) (d * d); }
            acc = wave_allsum_f64(acc);
            const int c = c0 + u;
            if (c < P.NC) {
                const float v = gp(lv.num)[c] / ((float) acc + addc);
                if (v >= best) { best = v; bi = c; }   // centroids ascend: '>=' keeps the last maximum (ggml_vec_argmax_f32)
            }
        }
        if (l + 1 < P.n_levels) request(l + 1);
        if (lane == 0) { sv[wave] = best; si[wave] = bi; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const unsigned tag = tag_base | (unsigned) (l + 1);
        if (tid == 0) {
            for (int w = 1; w < 4; w++) if (sv[w] > best || (sv[w] == best && si[w] > bi)) { best = sv[w]; bi = si[w]; }
            u64 * c = P.cand + (size_t) (l & 1) * 2 * grid + 2 * wg;
            st_granule(c, tag, __float_as_uint(best));
            st_granule(c + 1, tag, (unsigned) bi);
        }
        if (wave == 0) {
            for (int i = 0; i < P.delay; i++) __builtin_amdgcn_s_sleep(1);
            int token = 0;
            unsigned spins = 0;
            const unsigned base = (unsigned) (l & 1) * (2u * (unsigned) grid * 8u);
            for (;;) {
                u32x4 c[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { const int g = i * 64 + lane; c[i] = ld16_agent(cb, base + (unsigned) (g < grid ? g : grid - 1) * 16u); }
                float bv = -INFINITY; int bc = -1;
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int g = i * 64 + lane;
                    ok = ok && c[i].y == tag && c[i].w == tag;
                    if (g < grid) am_merge(bv, bc, __uint_as_float(c[i].x), (int) c[i].z);
                }
                if (__all(ok)) { am_wave(bv, bc); token = bc < 0 ? 0 : bc; break; }
                if (++spins > CH_SPIN_MAX || lds_load(&s_failed)) { if (lane == 0) { lds_store(&s_failed, 1u); *gp(P.err) = 2u; } break; }
                __builtin_amdgcn_s_sleep(1);
            }
            settle_vmcnt();
            if (lane == 0) s_code = token;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const int code = __builtin_amdgcn_readfirstlane(s_code);
        if (wg == 0 && tid == 0) { *gp(lv.idx_i) = code; *gp(lv.idx_f) = (float) code; }
        if (lv.resid_out) {
            const f32x4 q = *(const GLOBAL_AS f32x4 *) (gp(lv.emb) + (long long) code * lv.emb_row_bytes + lane * 16);
#pragma unroll
            for (int j = 0; j < 4; j++) x[j] = x[j] - q[j];
            if (wg == 0 && wave == 0) *(GLOBAL_AS f32x4 *) (gp(lv.resid_out) + lane * 4) = (f32x4) { x[0], x[1], x[2], x[3] };
        }
    }
    if (wg == 0 && tid == 0) *gp(P.launch_seq) = launch + 1u;
}

struct vq_chain_plan { vqc_params P; int grid; std::vector<vqc_level> levels; };
static bool vq_chain_shape(const vq_level_args * lv, int n) {
    if (n < 2 || n > VQC_LEVELS_MAX) return false;
    for (int i = 0; i < n; i++) {
        const vq_level_args & a = lv[i];
        if (a.D != 256 || a.NC != lv[0].NC || a.NC < 16 || a.NC > 4096 || ((uintptr_t) a.emb & 15) || (a.emb_row_bytes & 15) || !a.idx_i || !a.idx_f) return false;
        if (i > 0 && ((const void *) a.resid != (const void *) lv[i - 1].resid_out || a.resid_stride != 4)) return false;   // level i quantises what level i - 1 left
        if (i + 1 < n && (!a.resid_out || ((uintptr_t) a.resid_out & 15))) return false;
    }
    return true;
}
bool k_vq_chain_accept(const vq_level_args * lv, int n, int usable_cus) {
    static const int on = chain_env("MI355X_VQ_CHAIN", 1);
    if (!on || !vq_chain_shape(lv, n)) return false;
    const int grid = (lv[0].NC + 4 * VQC_CPW - 1) / (4 * VQC_CPW);
    if (grid > 256) return false;
    static const int force = chain_env("MI355X_CHAIN_GRID_FORCE", 0);
    int per_cu = 0;   // the workgroups wait for each other: the whole grid must be resident on the stream's compute units
    return force || (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) vq_chain_kernel, 256, 0) == hipSuccess && (long long) per_cu * usable_cus >= grid);
}
size_t k_vq_chain_ws_size(int n) { return GGML_PAD((size_t) n * sizeof(vqc_level), 256) + 256 + 2 * 2 * 256 * 8; }
vq_chain_plan * k_vq_chain_create(hipStream_t s, const vq_level_args * lv, int n, void * ws, unsigned * err) {
    GGML_ASSERT(vq_chain_shape(lv, n));
    vq_chain_plan * c = new vq_chain_plan;
    c->grid = (lv[0].NC + 4 * VQC_CPW - 1) / (4 * VQC_CPW);
    for (int i = 0; i < n; i++) c->levels.push_back({ lv[i].emb, (long long) lv[i].emb_row_bytes, lv[i].add_c, lv[i].num, lv[i].resid_out, lv[i].idx_f, lv[i].idx_i, 0 });
    char * base = (char *) ws;
    char * state = base + GGML_PAD((size_t) n * sizeof(vqc_level), 256);
    HIP_CHECK(hipMemcpyAsync(base, c->levels.data(), (size_t) n * sizeof(vqc_level), hipMemcpyHostToDevice, s));   // (from the plan's own vector, which lives as long as the plan)
    HIP_CHECK(hipMemsetAsync(state, 0, 256 + 2 * 2 * 256 * 8, s));
    memset(&c->P, 0, sizeof(c->P));
    c->P.levels = (const vqc_level *) base; c->P.n_levels = n; c->P.NC = lv[0].NC;
    c->P.resid = lv[0].resid; c->P.resid_stride = (long long) lv[0].resid_stride;
    c->P.launch_seq = (unsigned *) state; c->P.cand = (u64 *) (state + 256); c->P.err = err;
    static const int delay = chain_env("MI355X_VQ_CHAIN_DELAY", 16);
    c->P.delay = delay < 0 ? 0 : delay > 200 ? 200 : delay;
    return c;
}
void k_vq_chain_launch(hipStream_t s, const vq_chain_plan * c) { vq_chain_kernel<<<c->grid, 256, 0, s>>>(c->P); }
void k_vq_chain_free(vq_chain_plan * c) { delete c; }
