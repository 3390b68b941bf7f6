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
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    const float smax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    amax = wave_allmax_f32(amax);
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
