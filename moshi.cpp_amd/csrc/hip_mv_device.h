// hip_mv_device.h — device pieces shared by the block-quantised mat-vec kernels (hip_kernels_fused.hip) and the persistent
// chain engine (hip_chain.hip): the padded Q8_K activation record and its quantiser.
#pragma once

#include "hip_device.h"

#define XBLK_BYTES 304   // 256 q8 + 16 bsums (int16) + d (f32) + pad: 76-dword stride => conflict-free b128 reads

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct xblk { int8_t q[256]; int16_t bsums[16]; float d; float pad[3]; };
static_assert(sizeof(xblk) == XBLK_BYTES && offsetof(xblk, bsums) == 256 && offsetof(xblk, d) == 288, "xblk layout");

// quantise the 256 values held by one wave (4 per lane, contiguous) to a Q8_K block in LDS
__device__ __forceinline__ void quantize_block_q8k(xblk * dst, const float v[4], int lane) {
    // the signed value of largest magnitude (ggml: iscale = -127 / max); when +a and -a tie the sign is immaterial
    // (the sign comes from one ballot instead of a second wave-wide maximum: the largest signed value equals amax exactly when some element IS +amax)
    // (the magnitudes' maximum on their bit patterns: non-negative floats order like unsigned integers - one v_max_u32_dpp per reduction step where fmaxf
    // costs a move and three v_max_f32. A NaN input would win here where fmaxf ignores it; the result is meaningless either way)
    const unsigned a0 = __float_as_uint(v[0]) & 0x7fffffffu, a1 = __float_as_uint(v[1]) & 0x7fffffffu, a2 = __float_as_uint(v[2]) & 0x7fffffffu, a3 = __float_as_uint(v[3]) & 0x7fffffffu;
    const unsigned a01 = a0 > a1 ? a0 : a1, a23 = a2 > a3 ? a2 : a3;
    const float amax = wave_allmax_nonneg_f32(__uint_as_float(a01 > a23 ? a01 : a23));
    const float smax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    const float mx = __ballot(smax == amax) != 0ull ? amax : -amax;
    int q[4] = { 0, 0, 0, 0 };
    float d = 0.f;
    if (amax != 0.f) {
        const float iscale = -127.f / mx;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int qi = nearest_int_dev(iscale * v[k]); q[k] = qi < 127 ? qi : 127; }
        d = 1.f / iscale;
    }
    *(uint32_t *) (dst->q + lane * 4) = (uint32_t) (q[0] & 0xff) | ((uint32_t) (q[1] & 0xff) << 8) | ((uint32_t) (q[2] & 0xff) << 16) | ((uint32_t) (q[3] & 0xff) << 24);
    const int s4 = quad_allsum_i32(q[0] + q[1] + q[2] + q[3]);
    if ((lane & 3) == 0) dst->bsums[lane >> 2] = (int16_t) s4;
    if (lane == 0) dst->d = d;
}

// Q8_0 activations of the same 304-byte record (weights Q8_0 / Q4_0: ggml quantises the activation row to Q8_0 - 32-wide blocks, F16 scale)
struct xblk80 { int8_t q[256]; float d[8]; int16_t bsums[8]; };   // d = the F16-rounded scale of each 32-wide block, bsums = sum of its q
static_assert(sizeof(xblk80) == XBLK_BYTES, "xblk80 layout");

// quantise the 256 values held by one wave (4 per lane, contiguous) to eight Q8_0 blocks in LDS (quantize_row_q8_0_ref:
// d = amax / 127, q = roundf(x / d), d stored as F16)
__device__ __forceinline__ void quantize_block_q80(xblk80 * dst, const float v[4], int lane) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, dpp_f32<DPP_QUAD_XOR1>(amax));
    amax = fmaxf(amax, dpp_f32<DPP_QUAD_XOR2>(amax));
    amax = fmaxf(amax, dpp_f32<DPP_HALF_MIRROR>(amax));   // the 8 lanes of one 32-wide block
    const float d = amax / 127.f;
    const float id = d != 0.f ? 1.0f / d : 0.0f;
    int q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = (int) roundf(v[k] * id);
    *(uint32_t *) (dst->q + lane * 4) = (uint32_t) (q[0] & 0xff) | ((uint32_t) (q[1] & 0xff) << 8) | ((uint32_t) (q[2] & 0xff) << 16) | ((uint32_t) (q[3] & 0xff) << 24);
    int s = q[0] + q[1] + q[2] + q[3];
    s += dpp_i32<DPP_QUAD_XOR1>(s);
    s += dpp_i32<DPP_QUAD_XOR2>(s);
    s += dpp_i32<DPP_HALF_MIRROR>(s);
    if ((lane & 7) == 0) { dst->d[lane >> 3] = h2f(f2h(d)); dst->bsums[lane >> 3] = (int16_t) s; }
}

