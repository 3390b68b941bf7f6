// hip_device.h — device-side helpers shared by the gfx950 kernels: strided addressing, scalar type
// conversion with ggml's rounding rules, wave64 reductions, block-format decoding.
#pragma once

#include "hip_common.h"

__device__ __forceinline__ char * at(const tdesc & t, int64_t i0, int64_t i1, int64_t i2, int64_t i3) {
    return t.data + i0 * t.nb[0] + i1 * t.nb[1] + i2 * t.nb[2] + i3 * t.nb[3];
}
// element counts on this path are far below 2^31 (checked on the host): 32-bit division is ~5x cheaper than 64-bit
__device__ __forceinline__ void unravel(const tdesc & t, int64_t i, int64_t & i0, int64_t & i1, int64_t & i2, int64_t & i3) {
    uint32_t r = (uint32_t) i;
    const uint32_t n0 = (uint32_t) t.ne[0], n1 = (uint32_t) t.ne[1], n2 = (uint32_t) t.ne[2];
    uint32_t q = r / n0; i0 = r - q * n0; r = q;
    q = r / n1; i1 = r - q * n1; r = q;
    q = r / n2; i2 = r - q * n2; i3 = q;
}
__device__ __forceinline__ void row_coords(const tdesc & t, int64_t rr, int64_t & i1, int64_t & i2, int64_t & i3) {
    uint32_t r = (uint32_t) rr;
    const uint32_t n1 = (uint32_t) t.ne[1], n2 = (uint32_t) t.ne[2];
    uint32_t q = r / n1; i1 = r - q * n1; r = q;
    q = r / n2; i2 = r - q * n2; i3 = q;
}
__device__ __forceinline__ int64_t wrap(int64_t i, int64_t n) { return n == 1 ? 0 : (i < n ? i : (int64_t) ((uint32_t) i % (uint32_t) n)); }
__host__ __device__ __forceinline__ int elem_size(int type) {
    switch (type) {
        case GGML_TYPE_F32: case GGML_TYPE_I32: return 4;
        case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_I16: return 2;
        case GGML_TYPE_I64: case GGML_TYPE_F64: return 8;
        default: return 1;
    }
}

// ---- scalar conversions -------------------------------------------------------------------------------
__device__ __forceinline__ float h2f(uint16_t h) { _Float16 v; __builtin_memcpy(&v, &h, 2); return (float) v; }
__device__ __forceinline__ uint16_t f2h(float f) { _Float16 v = (_Float16) f; uint16_t h; __builtin_memcpy(&h, &v, 2); return h; }
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t) h << 16); }
// round-to-nearest-even, NaN stays NaN (same integer recipe as the oracle)
__device__ __forceinline__ uint16_t f2bf(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t) ((u >> 16) | 64);
    return (uint16_t) ((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
}
__device__ __forceinline__ int nearest_int_dev(float fval) {
    const float val = fval + 12582912.f;
    return (int) (__float_as_uint(val) & 0x007fffffu) - 0x00400000;
}
__device__ __forceinline__ float ld_as_f32(const char * p, int type) {
    switch (type) {
        case GGML_TYPE_F32:  return *(const float *) p;
        case GGML_TYPE_F16:  return h2f(*(const uint16_t *) p);
        case GGML_TYPE_BF16: return bf2f(*(const uint16_t *) p);
        case GGML_TYPE_I32:  return (float) *(const int32_t *) p;
        default:             return NAN;
    }
}
__device__ __forceinline__ void st_from_f32(char * p, int type, float v) {
    switch (type) {
        case GGML_TYPE_F32:  *(float *) p = v; break;
        case GGML_TYPE_F16:  *(uint16_t *) p = f2h(v); break;
        case GGML_TYPE_BF16: *(uint16_t *) p = f2bf(v); break;
        case GGML_TYPE_I32:  *(int32_t *) p = (int32_t) v; break;
        default: break;
    }
}

// ---- activations ---------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f32(float x) {
    return 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x)));
}
// ggml's CPU backend evaluates gelu through an F16 lookup table: f16(gelu(f16(x)))
__device__ __forceinline__ float gelu_table(float x) {
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    return h2f(f2h(gelu_f32(h2f(f2h(x)))));
}
__device__ __forceinline__ float apply_unary(int uop, float x) {
    switch (uop) {
        case GGML_UNARY_OP_NEG:  return -x;
        case GGML_UNARY_OP_SILU: return x / (1.0f + expf(-x));
        case GGML_UNARY_OP_ELU:  return x > 0.f ? x : expm1f(x);
        case GGML_UNARY_OP_GELU: return gelu_table(x);
        case GGML_UNARY_OP_RELU: return x > 0.f ? x : 0.f;
        case GGML_UNARY_OP_TANH: return tanhf(x);
        case GGML_UNARY_OP_SIGMOID: return 1.f / (1.f + expf(-x));
        case GGML_UNARY_OP_EXP:  return expf(x);
        default: return NAN;
    }
}

// ---- wave64 all-reduce on the DPP path (no LDS round trips: __shfl_xor lowers to ds_bpermute, ~100 cycles a step) ----
#define DPP_QUAD_XOR1   0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2   0x4E   // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i inside each group of 8
#define DPP_ROW_MIRROR  0x140  // lane i <-> 15 - i inside each row of 16
// (bound_ctrl = true: every permutation used here reads a valid lane for every lane, so the flag changes no value - but with it, full row / bank masks and
// an unused `old`, the compiler folds the move into the consuming VOP2: v_add_f32_dpp / v_max_u32_dpp ... instead of v_mov_b32_dpp + the op)
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// maximum over the 64 lanes of a NON-NEGATIVE float (no NaN), result in every lane: non-negative floats order like their bit patterns, and an unsigned
// integer maximum needs no canonicalisation of its operands (fmaxf costs three v_max_f32 per step: one per operand to quiet signalling NaNs, one for the
// maximum) and folds into the DPP move: one v_max_u32_dpp per step
__device__ __forceinline__ float wave_allmax_nonneg_f32(float v) {
    unsigned u = __float_as_uint(v);
    auto umax = [](unsigned a, unsigned b) { return a > b ? a : b; };
    u = umax(u, (unsigned) dpp_i32<DPP_QUAD_XOR1>((int) u));
    u = umax(u, (unsigned) dpp_i32<DPP_QUAD_XOR2>((int) u));
    u = umax(u, (unsigned) dpp_i32<DPP_HALF_MIRROR>((int) u));
    u = umax(u, (unsigned) dpp_i32<DPP_ROW_MIRROR>((int) u));
    const unsigned r0 = (unsigned) __builtin_amdgcn_readlane((int) u, 0), r1 = (unsigned) __builtin_amdgcn_readlane((int) u, 16);
    const unsigned r2 = (unsigned) __builtin_amdgcn_readlane((int) u, 32), r3 = (unsigned) __builtin_amdgcn_readlane((int) u, 48);
    return __uint_as_float(umax(umax(r0, r1), umax(r2, r3)));
}
// max over the 64 lanes, result in every lane
__device__ __forceinline__ float wave_allmax_f32(float v) {
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// sum over the 64 lanes in double, result in every lane (fixed order)
__device__ __forceinline__ double wave_allsum_f64(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int wave_allmax_i32(int v) {
    v = max(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = max(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = max(v, dpp_i32<DPP_HALF_MIRROR>(v));
    v = max(v, dpp_i32<DPP_ROW_MIRROR>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// Workgroup barrier for data exchanged through LDS only. __syncthreads() also waits for every outstanding global load of the
// wave (vmcnt(0) on gfx9), which would serialise a prologue behind the weight tile / ring rows it deliberately left in flight;
// this one waits for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- cross-workgroup hand-off without cache-wide fences -------------------------------------------------------------
// An agent-scope release/acquire fence writes back / invalidates the whole 4 MB L2 of the XCD, which costs many microseconds in
// the middle of a weight stream. Handed-off values are instead written and read with agent-coherent (sc1) accesses, which go
// past the non-coherent caches (MI355X_MICROARCH.md "Correctness boundaries", second form). The data itself is published with
// xchg_agent_wait below, not st_agent: see the note there.
template <typename V> __device__ __forceinline__ void st_agent(V * p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V> __device__ __forceinline__ V ld_agent(const V * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
// Cross-workgroup hand-offs publish their data with RETURNING agent-scope exchanges: such an exchange is complete at the device's coherence
// point when its old value is back, and consuming that value (the empty asm) makes the compiler emit the returning form and wait for it.
// A plain agent-scope store followed by drain_stores() and the counter increment is NOT enough: the store is acknowledged by this XCD's L2,
// and about once per 1e5 hand-offs a consumer on another XCD saw the counter before the data (tests/microbench/soak.py: two runs of the same
// 3300 frames diverged). An agent-scope release fence by the incrementing thread also fixes it, but its L2 write-back request costs ~10 % of
// the frame at long context; the exchanges cost one round trip.
__device__ __forceinline__ void xchg_agent_wait(float * p, float v) { const float o = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("" :: "v"(o)); }
__device__ __forceinline__ void xchg_agent_wait(int * p, int v) { const int o = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("" :: "v"(o)); }
__device__ __forceinline__ void xchg_agent_wait(double * p, double v) { const double o = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("" :: "v"(o)); }

// sum over each aligned group of `n` lanes (n = 8 or 16 on the DPP path, any power of two otherwise), result in all of them
__device__ __forceinline__ double group_allsum_f64(double v, int n) {
    if (n == 16 || n == 8) {
        v += dpp_f64<DPP_QUAD_XOR1>(v);
        v += dpp_f64<DPP_QUAD_XOR2>(v);
        v += dpp_f64<DPP_HALF_MIRROR>(v);
        if (n == 16) v += dpp_f64<DPP_ROW_MIRROR>(v);
        return v;
    }
    for (int o = n >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over each aligned group of 4 lanes, result in all 4
__device__ __forceinline__ int quad_allsum_i32(int v) { v += dpp_i32<DPP_QUAD_XOR1>(v); v += dpp_i32<DPP_QUAD_XOR2>(v); return v; }
// sum over each aligned group of 16 lanes (one DPP row), result in all 16
__device__ __forceinline__ float row16_allsum_f32(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    return v;
}

// ---- wave64 reductions ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f32(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- block formats -------------------------------------------------------------------------------------
// 6-bit (scale, min) pairs of a Q4_K super-block -> sc[0..7] in the bytes of s01, mins in the bytes of m01
__device__ __forceinline__ void q4k_unpack_scales(const uint8_t * scales, uint32_t sc[2], uint32_t mn[2]) {
    const uint32_t kmask1 = 0x3f3f3f3fu, kmask2 = 0x0f0f0f0fu, kmask3 = 0x03030303u;
    const uint32_t u0 = ((const uint32_t *) scales)[0], u1 = ((const uint32_t *) scales)[1], u2 = ((const uint32_t *) scales)[2];
    sc[0] = u0 & kmask1;
    sc[1] = (u2 & kmask2) | (((u0 >> 6) & kmask3) << 4);
    mn[0] = u1 & kmask1;
    mn[1] = ((u2 >> 4) & kmask2) | (((u1 >> 6) & kmask3) << 4);
}

// the same from the three scale words held in registers
__device__ __forceinline__ void q4k_unpack_scales_w(uint32_t u0, uint32_t u1, uint32_t u2, uint32_t sc[2], uint32_t mn[2]) {
    const uint32_t kmask1 = 0x3f3f3f3fu, kmask2 = 0x0f0f0f0fu, kmask3 = 0x03030303u;
    sc[0] = u0 & kmask1;
    sc[1] = (u2 & kmask2) | (((u0 >> 6) & kmask3) << 4);
    mn[0] = u1 & kmask1;
    mn[1] = ((u2 >> 4) & kmask2) | (((u1 >> 6) & kmask3) << 4);
}

__device__ __forceinline__ float dequant_elem(const char * row, int type, int64_t i) {
    switch (type) {
        case GGML_TYPE_F32:  return ((const float *) row)[i];
        case GGML_TYPE_F16:  return h2f(((const uint16_t *) row)[i]);
        case GGML_TYPE_BF16: return bf2f(((const uint16_t *) row)[i]);
        case GGML_TYPE_Q8_0: { const block_q8_0 * b = (const block_q8_0 *) row + i / 32; return b->qs[i % 32] * h2f(b->d); }
        case GGML_TYPE_Q4_0: {
            const block_q4_0 * b = (const block_q4_0 *) row + i / 32;
            const int j = (int) (i % 32);
            const int q = j < 16 ? (b->qs[j] & 0x0F) : (b->qs[j - 16] >> 4);
            return (q - 8) * h2f(b->d);
        }
        case GGML_TYPE_Q4_K: {
            const block_q4_K * b = (const block_q4_K *) row + i / 256;
            const int j = (int) (i % 256);
            const int sub = j / 32, l = j % 32;
            uint32_t sc[2], mn[2];
            q4k_unpack_scales(b->scales, sc, mn);
            const uint32_t s = (sc[sub >> 2] >> (8 * (sub & 3))) & 0xff, m = (mn[sub >> 2] >> (8 * (sub & 3))) & 0xff;
            const uint8_t qb = b->qs[(sub >> 1) * 32 + l];
            const int q = (sub & 1) ? (qb >> 4) : (qb & 0xF);
            const float d = h2f(b->d) * (float) s, mm = h2f(b->dmin) * (float) m;
            return d * (float) q - mm;
        }
        default: return NAN;
    }
}

// signed 4x8-bit dot with 32-bit accumulate (v_dot4_i32_i8)
__device__ __forceinline__ int dot4_i8(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// one Q4_K super-block against 256 q8_K-quantised activations: d8*(d*sum_j sc_j*<q4_j,q8_j> - dmin*sum_j m_j*bsum_j)
// q8 must be 4-byte aligned, bsums 2-byte aligned
__device__ __forceinline__ float q4k_q8k_block_dot(const block_q4_K * xb, const int8_t * q8, const int16_t * bsums, float d8) {
    uint32_t sc[2], mn[2];
    q4k_unpack_scales(xb->scales, sc, mn);
    const uint32_t * qs = (const uint32_t *) xb->qs;
    const int * y = (const int *) q8;
    int isum = 0, msum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int lo = 0, hi = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t q = qs[j * 8 + t];
            lo = dot4_i8((int) (q & 0x0F0F0F0Fu), y[j * 16 + t], lo);
            hi = dot4_i8((int) ((q >> 4) & 0x0F0F0F0Fu), y[j * 16 + 8 + t], hi);
        }
        const int s0 = (int) ((sc[j >> 1] >> (16 * (j & 1))) & 0xff), s1 = (int) ((sc[j >> 1] >> (16 * (j & 1) + 8)) & 0xff);
        const int m0 = (int) ((mn[j >> 1] >> (16 * (j & 1))) & 0xff), m1 = (int) ((mn[j >> 1] >> (16 * (j & 1) + 8)) & 0xff);
        // 6-bit scales times sums bounded by 32 * 15 * 127 < 2^23: exact in the full-rate 24-bit multiplier (v_mul_lo_u32 is quarter rate)
        isum += __mul24(s0, lo) + __mul24(s1, hi);
        msum += __mul24(m0, (int) bsums[4 * j] + (int) bsums[4 * j + 1]) + __mul24(m1, (int) bsums[4 * j + 2] + (int) bsums[4 * j + 3]);
    }
    const float d = h2f(xb->d) * d8, dmin = h2f(xb->dmin) * d8;
    return d * (float) isum - dmin * (float) msum;
}
