// hip_kernels_generic.hip — one gfx950 kernel per ggml operator, covering every op the
// moshi.cpp graphs emit with arbitrary strides. These are the always-correct fallbacks; the per-frame
// hot path goes through the fused kernels in hip_kernels_fused.hip wherever the planner matches.
//
// Numerics follow the ggml CPU semantics the oracle restates (oracle/oracle.cpp): double accumulators
// for norms / soft_max / sums / float dots, activations rounded to the weight's dot type in mul_mat.
#include "hip_common.h"
#include "hip_device.h"
#include <type_traits>

#define BLOCK 256

static inline int nblocks(int64_t n, int bs = BLOCK) { return (int) ((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------------
__global__ void binary_kernel(int op, tdesc dst, tdesc a, tdesc b, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t i0, i1, i2, i3;
    unravel(dst, i, i0, i1, i2, i3);
    const float x = *(const float *) at(a, i0, i1, i2, i3);
    const float y = *(const float *) at(b, wrap(i0, b.ne[0]), wrap(i1, b.ne[1]), wrap(i2, b.ne[2]), wrap(i3, b.ne[3]));
    float r;
    switch (op) {
        case GGML_OP_ADD: r = x + y; break;
        case GGML_OP_SUB: r = x - y; break;
        case GGML_OP_MUL: r = x * y; break;
        default:          r = x / y; break;
    }
    *(float *) at(dst, i0, i1, i2, i3) = r;
}
void k_binary(hipStream_t s, int op, tdesc dst, tdesc a, tdesc b) {
    const int64_t n = td_nelements(dst);
    if (n == 0) return;
    GGML_ASSERT(dst.type == GGML_TYPE_F32 && a.type == GGML_TYPE_F32 && b.type == GGML_TYPE_F32);
    binary_kernel<<<nblocks(n), BLOCK, 0, s>>>(op, dst, a, b, n);
}

__global__ void unary_kernel(int uop, tdesc dst, tdesc a, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t i0, i1, i2, i3;
    unravel(dst, i, i0, i1, i2, i3);
    const float x = *(const float *) at(a, i0, i1, i2, i3);
    *(float *) at(dst, i0, i1, i2, i3) = apply_unary(uop, x);
}
void k_unary(hipStream_t s, int uop, tdesc dst, tdesc a) {
    const int64_t n = td_nelements(dst);
    if (n == 0) return;
    GGML_ASSERT(dst.type == GGML_TYPE_F32 && a.type == GGML_TYPE_F32);
    unary_kernel<<<nblocks(n), BLOCK, 0, s>>>(uop, dst, a, n);
}

__global__ void scale_kernel(tdesc dst, tdesc a, float sc, float bias, float mn, float mx, int clamp, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t i0, i1, i2, i3;
    unravel(dst, i, i0, i1, i2, i3);
    const float x = *(const float *) at(a, i0, i1, i2, i3);
    *(float *) at(dst, i0, i1, i2, i3) = clamp ? (x < mn ? mn : x > mx ? mx : x) : x * sc + bias;
}
void k_scale(hipStream_t s, tdesc dst, tdesc a, float scale, float bias) {
    const int64_t n = td_nelements(dst);
    if (n) scale_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, a, scale, bias, 0, 0, 0, n);
}
void k_clamp(hipStream_t s, tdesc dst, tdesc a, float mn, float mx) {
    const int64_t n = td_nelements(dst);
    if (n) scale_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, a, 0, 0, mn, mx, 1, n);
}

// logical-order, type-converting copy; shapes may differ as long as the element counts match
__global__ void cpy_kernel(tdesc dst, tdesc src, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t s0, s1, s2, s3, d0, d1, d2, d3;
    unravel(src, i, s0, s1, s2, s3);
    unravel(dst, i, d0, d1, d2, d3);
    const char * sp = at(src, s0, s1, s2, s3);
    char * dp = at(dst, d0, d1, d2, d3);
    if (src.type == dst.type) {
        switch (elem_size(src.type)) {
            case 4: *(uint32_t *) dp = *(const uint32_t *) sp; break;
            case 2: *(uint16_t *) dp = *(const uint16_t *) sp; break;
            case 8: *(uint64_t *) dp = *(const uint64_t *) sp; break;
            default: *dp = *sp; break;
        }
    } else {
        st_from_f32(dp, dst.type, ld_as_f32(sp, src.type));
    }
}
// ---- weight (re)quantisation on the device: ggml_cast / cpy of F32 / F16 / BF16 rows to Q8_0 / Q4_0 / Q4_K (WeightLoader's load-time cast,
// ---- src/loader.h:160-187), the same float operations in the same order as the host quantisers of ggml_core.cpp, hence the same bytes ------
// element (i0, row r) of the source rows as F32; rows are the flattened dims 1..3
struct qsrc { const char * base; int64_t nb0, nb1, nb2, nb3, ne1, ne2; int type; };
__device__ __forceinline__ float qsrc_at(const qsrc & q, int64_t i0, int64_t r) {
    const int64_t i1 = r % q.ne1, i2 = (r / q.ne1) % q.ne2, i3 = r / (q.ne1 * q.ne2);
    return ld_as_f32(q.base + i0 * q.nb0 + i1 * q.nb1 + i2 * q.nb2 + i3 * q.nb3, q.type);
}
// one thread per 32-wide block
template <int DST>
__global__ void quantize32_kernel(qsrc q, char * out, int64_t nb_row, int64_t nblocks_total) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks_total) return;
    const int64_t r = b / nb_row, i0 = (b % nb_row) * 32;
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = qsrc_at(q, i0 + j, r);
    if (DST == GGML_TYPE_Q8_0) {
        block_q8_0 * y = (block_q8_0 *) out + b;
        float amax = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) { const float v = fabsf(x[j]); if (v > amax) amax = v; }
        const float d = amax / 127.f, id = d ? 1.f / d : 0.f;
        y->d = f2h(d);
#pragma unroll
        for (int j = 0; j < 32; j++) y->qs[j] = (int8_t) roundf(x[j] * id);
    } else {
        block_q4_0 * y = (block_q4_0 *) out + b;
        float amax = 0, mx = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); mx = v; } }
        const float d = mx / -8.f, id = d ? 1.f / d : 0.f;
        y->d = mx == 0.f ? (uint16_t) 0x8000 : f2h(d);   // an all-zero block: 0 / -8 = -0.0 on the host, sign included
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float x0 = x[j] * id, x1 = x[16 + j] * id;
            const int a = (int) (x0 + 8.5f), c = (int) (x1 + 8.5f);
            y->qs[j] = (uint8_t) ((a > 15 ? 15 : a) | ((c > 15 ? 15 : c) << 4));
        }
    }
}
// Q4_K (quantize_row_q4_K_ref + make_qkx2_quants, as restated in ggml_core.cpp): eight lanes per super-block, one per 32-weight sub-block; the
// 21-step scale search runs per lane in the reference's order, the super-block maxima meet over the 8 lanes, the nibbles are packed through LDS
__global__ void __launch_bounds__(256) quantize_q4k_kernel(qsrc q, block_q4_K * out, int64_t nb_row, int64_t nsb_total) {
    __shared__ uint8_t Ls[32][256];
    const int tid = threadIdx.x, j = tid & 7, sbl = tid >> 3;                       // sub-block, super-block within the workgroup
    const int64_t sb = (int64_t) blockIdx.x * 32 + sbl;
    const bool valid = sb < nsb_total;
    const int64_t sbc = valid ? sb : nsb_total - 1;
    const int64_t r = sbc / nb_row, i0 = (sbc % nb_row) * 256 + 32 * j;
    float x[32], w[32];
    uint8_t L[32], Laux[32];
#pragma unroll
    for (int l = 0; l < 32; l++) x[l] = qsrc_at(q, i0 + l, r);
    float sum_x2 = 0;
#pragma unroll
    for (int l = 0; l < 32; l++) sum_x2 += x[l] * x[l];
    const float av_x = sqrtf(sum_x2 / 32);
#pragma unroll
    for (int l = 0; l < 32; l++) w[l] = av_x + fabsf(x[l]);
    // make_qkx2_quants(32, 15, x, w, L, &the_min, Laux, -1, 0.1, 20, false)
    float scale, the_min;
    {
        const int nmax = 15;
        float mn = x[0], mx = x[0], sum_w = w[0], sum_x = sum_w * x[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; sum_w += w[i]; sum_x += w[i] * x[i]; }
        if (mn > 0) mn = 0;
        if (mx == mn) {
#pragma unroll
            for (int i = 0; i < 32; ++i) L[i] = 0;
            scale = 0.f; the_min = -mn;
        } else {
            float iscale = nmax / (mx - mn);
            scale = 1 / iscale;
            float best_mad = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                int l = nearest_int_dev(iscale * (x[i] - mn));
                L[i] = (uint8_t) (l < 0 ? 0 : l > nmax ? nmax : l);
                float diff = scale * L[i] + mn - x[i];
                diff = diff * diff;
                best_mad += w[i] * diff;
            }
            for (int is = 0; is <= 20; ++is) {
                iscale = (-1.f + 0.1f * is + nmax) / (mx - mn);
                float sum_l = 0, sum_l2 = 0, sum_xl = 0;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    int l = nearest_int_dev(iscale * (x[i] - mn));
                    l = l < 0 ? 0 : l > nmax ? nmax : l;
                    Laux[i] = (uint8_t) l;
                    sum_l += w[i] * l; sum_l2 += w[i] * l * l; sum_xl += w[i] * l * x[i];
                }
                const float D = sum_w * sum_l2 - sum_l * sum_l;
                if (D > 0) {
                    float this_scale = (sum_w * sum_xl - sum_x * sum_l) / D, this_min = (sum_l2 * sum_x - sum_l * sum_xl) / D;
                    if (this_min > 0) { this_min = 0; this_scale = sum_xl / sum_l2; }
                    float mad = 0;
#pragma unroll
                    for (int i = 0; i < 32; ++i) { float diff = this_scale * Laux[i] + this_min - x[i]; diff = diff * diff; mad += w[i] * diff; }
                    if (mad < best_mad) {
#pragma unroll
                        for (int i = 0; i < 32; ++i) L[i] = Laux[i];
                        best_mad = mad; scale = this_scale; mn = this_min;
                    }
                }
            }
            the_min = -mn;
        }
    }
    // super-block maxima over the 8 lanes (max is order-free)
    float max_scale = scale > 0 ? scale : 0.f, max_min = the_min > 0 ? the_min : 0.f;
    max_scale = fmaxf(max_scale, dpp_f32<DPP_QUAD_XOR1>(max_scale)); max_min = fmaxf(max_min, dpp_f32<DPP_QUAD_XOR1>(max_min));
    max_scale = fmaxf(max_scale, dpp_f32<DPP_QUAD_XOR2>(max_scale)); max_min = fmaxf(max_min, dpp_f32<DPP_QUAD_XOR2>(max_min));
    max_scale = fmaxf(max_scale, dpp_f32<DPP_HALF_MIRROR>(max_scale)); max_min = fmaxf(max_min, dpp_f32<DPP_HALF_MIRROR>(max_min));
    const float inv_scale = max_scale > 0 ? 63.f / max_scale : 0.f, inv_min = max_min > 0 ? 63.f / max_min : 0.f;
    int ls = nearest_int_dev(inv_scale * scale) & 0xff, lm = nearest_int_dev(inv_min * the_min) & 0xff;   // (uint8_t) nearest_int(.)
    if (ls > 63) ls = 63;
    if (lm > 63) lm = 63;
    const uint16_t dh = f2h(max_scale / 63.f), dmh = f2h(max_min / 63.f);
    const float d = h2f(dh) * ls;
    if (d) {
        const float dm = h2f(dmh) * lm;
#pragma unroll
        for (int l = 0; l < 32; l++) { int v = nearest_int_dev((x[l] + dm) / d); L[l] = (uint8_t) (v < 0 ? 0 : v > 15 ? 15 : v); }
    }
#pragma unroll
    for (int l = 0; l < 32; l++) Ls[sbl][32 * j + l] = L[l];
    __shared__ uint8_t sls[32][8], slm[32][8];
    sls[sbl][j] = (uint8_t) ls; slm[sbl][j] = (uint8_t) lm;
    __syncthreads();
    if (!valid) return;
    block_q4_K * y = out + sb;
    // lane j packs 16 bytes of qs: group g = j / 2 (sub-blocks 2g | 2g + 1), half h = j % 2
    {
        const int g = j >> 1, h = j & 1;
#pragma unroll
        for (int l = 0; l < 16; l++) y->qs[32 * g + 16 * h + l] = (uint8_t) (Ls[sbl][64 * g + 16 * h + l] | (Ls[sbl][64 * g + 32 + 16 * h + l] << 4));
    }
    if (j < 4) {
        y->scales[j]     = (uint8_t) (sls[sbl][j] | ((sls[sbl][j + 4] >> 4) << 6));
        y->scales[j + 4] = (uint8_t) (slm[sbl][j] | ((slm[sbl][j + 4] >> 4) << 6));
        y->scales[j + 8] = (uint8_t) ((sls[sbl][j + 4] & 0xF) | ((slm[sbl][j + 4] & 0xF) << 4));
    }
    if (j == 0) { y->d = dh; y->dmin = dmh; }
}
static bool k_quantize_rows(hipStream_t s, tdesc dst, tdesc src) {
    if (!(src.type == GGML_TYPE_F32 || src.type == GGML_TYPE_F16 || src.type == GGML_TYPE_BF16)) return false;
    if (!(dst.type == GGML_TYPE_Q8_0 || dst.type == GGML_TYPE_Q4_0 || dst.type == GGML_TYPE_Q4_K)) return false;
    const int64_t K = src.ne[0], rows = src.ne[1] * src.ne[2] * src.ne[3], blk = ggml_blck_size((enum ggml_type) dst.type);
    if (K % blk != 0 || dst.ne[0] != K || td_nelements(dst) != td_nelements(src)) return false;
    // the destination must be dense rows (it is a freshly allocated weight tensor)
    const int64_t rb = (int64_t) ggml_row_size((enum ggml_type) dst.type, K);
    if (dst.nb[1] != rb || (dst.ne[2] > 1 && dst.nb[2] != rb * dst.ne[1]) || (dst.ne[3] > 1 && dst.nb[3] != rb * dst.ne[1] * dst.ne[2])) return false;
    if (dst.ne[1] != src.ne[1] || dst.ne[2] != src.ne[2] || dst.ne[3] != src.ne[3]) return false;
    qsrc q = { src.data, src.nb[0], src.nb[1], src.nb[2], src.nb[3], src.ne[1], src.ne[2], src.type };
    const int64_t nb_row = K / blk, total = nb_row * rows;
    if (dst.type == GGML_TYPE_Q4_K) quantize_q4k_kernel<<<(unsigned) ((total + 31) / 32), 256, 0, s>>>(q, (block_q4_K *) dst.data, nb_row, total);
    else if (dst.type == GGML_TYPE_Q8_0) quantize32_kernel<GGML_TYPE_Q8_0><<<nblocks(total), BLOCK, 0, s>>>(q, dst.data, nb_row, total);
    else quantize32_kernel<GGML_TYPE_Q4_0><<<nblocks(total), BLOCK, 0, s>>>(q, dst.data, nb_row, total);
    return true;
}

void k_cpy(hipStream_t s, tdesc dst, tdesc src) {
    const int64_t n = td_nelements(src);
    if (n == 0) return;
    GGML_ASSERT(n == td_nelements(dst));
    const bool q = ggml_is_quantized((enum ggml_type) src.type) || ggml_is_quantized((enum ggml_type) dst.type);
    if (q) {
        if (src.type != dst.type && k_quantize_rows(s, dst, src)) return;
        // otherwise only the byte-identical contiguous case is handled on the device
        GGML_ASSERT(src.type == dst.type && "this (re)quantising copy is not implemented on the device; cast on the host device");
        const size_t bytes = ggml_row_size((enum ggml_type) src.type, src.ne[0]) * (size_t) (src.ne[1] * src.ne[2] * src.ne[3]);
        HIP_CHECK(hipMemcpyAsync(dst.data, src.data, bytes, hipMemcpyDeviceToDevice, s));
        return;
    }
    cpy_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, src, n);
}

__global__ void concat_kernel(tdesc dst, tdesc a, tdesc b, int dim, int es, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t c[4];
    unravel(dst, i, c[0], c[1], c[2], c[3]);
    const bool in_a = c[dim] < a.ne[dim];
    const char * sp;
    if (in_a) sp = at(a, c[0], c[1], c[2], c[3]);
    else { c[dim] -= a.ne[dim]; sp = at(b, c[0], c[1], c[2], c[3]); c[dim] += a.ne[dim]; }
    char * dp = at(dst, c[0], c[1], c[2], c[3]);
    if (es == 4) *(uint32_t *) dp = *(const uint32_t *) sp; else if (es == 2) *(uint16_t *) dp = *(const uint16_t *) sp; else *dp = *sp;
}
void k_concat(hipStream_t s, tdesc dst, tdesc a, tdesc b, int dim) {
    const int64_t n = td_nelements(dst);
    if (n) concat_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, a, b, dim, (int) ggml_type_size((enum ggml_type) dst.type), n);
}

__global__ void repeat_kernel(tdesc dst, tdesc a, int es, int pad, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t i0, i1, i2, i3;
    unravel(dst, i, i0, i1, i2, i3);
    char * dp = at(dst, i0, i1, i2, i3);
    if (pad) {
        const bool in = i0 < a.ne[0] && i1 < a.ne[1] && i2 < a.ne[2] && i3 < a.ne[3];
        *(float *) dp = in ? *(const float *) at(a, i0, i1, i2, i3) : 0.f;
        return;
    }
    const char * sp = at(a, wrap(i0, a.ne[0]), wrap(i1, a.ne[1]), wrap(i2, a.ne[2]), wrap(i3, a.ne[3]));
    if (es == 4) *(uint32_t *) dp = *(const uint32_t *) sp; else if (es == 2) *(uint16_t *) dp = *(const uint16_t *) sp; else *dp = *sp;
}
void k_repeat(hipStream_t s, tdesc dst, tdesc a) {
    const int64_t n = td_nelements(dst);
    if (n) repeat_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, a, (int) ggml_type_size((enum ggml_type) dst.type), 0, n);
}
void k_pad(hipStream_t s, tdesc dst, tdesc a) {
    const int64_t n = td_nelements(dst);
    if (n) repeat_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, a, 4, 1, n);
}

__global__ void arange_kernel(float * dst, float start, float step, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = start + step * (float) i;
}
void k_arange(hipStream_t s, tdesc dst, float start, float step) {
    const int64_t n = dst.ne[0];
    if (n) arange_kernel<<<nblocks(n), BLOCK, 0, s>>>((float *) dst.data, start, step, n);
}

// ---------------------------------------------------------------------------------------------------
// reductions (one block per row)
// ---------------------------------------------------------------------------------------------------
__device__ inline double block_sum_f64(double v, double * sh) {
    const int tid = threadIdx.x;
    v = wave_sum_f64(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) r += sh[w];
    return r;
}
__device__ inline float block_max_f32(float v, float * sh) {
    const int tid = threadIdx.x;
    v = wave_max_f32(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) r = fmaxf(r, sh[w]);
    return r;
}

__global__ void sum_rows_kernel(tdesc dst, tdesc a) {
    __shared__ double sh[BLOCK / 64];
    int64_t i1, i2, i3;
    row_coords(a, blockIdx.x, i1, i2, i3);
    double acc = 0;
    for (int64_t i0 = threadIdx.x; i0 < a.ne[0]; i0 += blockDim.x) acc += (double) *(const float *) at(a, i0, i1, i2, i3);
    acc = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) *(float *) at(dst, 0, i1, i2, i3) = (float) acc;
}
void k_sum_rows(hipStream_t s, tdesc dst, tdesc a) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    if (rows) sum_rows_kernel<<<(int) rows, BLOCK, 0, s>>>(dst, a);
}
__global__ void sum_all_kernel(tdesc dst, tdesc a, int64_t n) {
    __shared__ double sh[BLOCK / 64];
    double acc = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        int64_t i0, i1, i2, i3;
        unravel(a, i, i0, i1, i2, i3);
        acc += (double) *(const float *) at(a, i0, i1, i2, i3);
    }
    acc = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) *(float *) dst.data = (float) acc;
}
void k_sum_all(hipStream_t s, tdesc dst, tdesc a) { sum_all_kernel<<<1, BLOCK, 0, s>>>(dst, a, td_nelements(a)); }

// ggml's CPU argmax keeps overwriting the index while the running maximum equals the element (ggml_vec_argmax_f32), so the LAST
// maximum of a row wins; an all -inf row yields ne0 - 1
__global__ void argmax_kernel(tdesc dst, tdesc a) {
    __shared__ float shv[BLOCK];
    __shared__ int shi[BLOCK];
    const int64_t i1 = blockIdx.x;
    float best = -INFINITY; int bi = -1;
    for (int64_t i0 = threadIdx.x; i0 < a.ne[0]; i0 += blockDim.x) {
        const float v = *(const float *) at(a, i0, i1, 0, 0);
        if (v >= best) { best = v; bi = (int) i0; }
    }
    shv[threadIdx.x] = best; shi[threadIdx.x] = bi;
    __syncthreads();
    for (int st = BLOCK / 2; st > 0; st >>= 1) {
        if ((int) threadIdx.x < st) {
            const float v = shv[threadIdx.x + st]; const int j = shi[threadIdx.x + st];
            if (v > shv[threadIdx.x] || (v == shv[threadIdx.x] && j > shi[threadIdx.x])) { shv[threadIdx.x] = v; shi[threadIdx.x] = j; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ((int32_t *) dst.data)[i1] = shi[0] < 0 ? 0 : shi[0];
}
void k_argmax(hipStream_t s, tdesc dst, tdesc a) { if (a.ne[1]) argmax_kernel<<<(int) a.ne[1], BLOCK, 0, s>>>(dst, a); }

// rank-by-counting sort: rank_i = #{j : v_j before v_i}; ties keep the lower index first (stable)
__global__ void argsort_kernel(tdesc dst, tdesc a, int desc) {
    __shared__ float tile[BLOCK];
    int64_t i1, i2, i3;
    row_coords(a, blockIdx.y, i1, i2, i3);
    const int64_t n = a.ne[0];
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const float vi = i < n ? *(const float *) at(a, i, i1, i2, i3) : 0.f;
    int rank = 0;
    for (int64_t base = 0; base < n; base += BLOCK) {
        const int64_t j = base + threadIdx.x;
        __syncthreads();
        tile[threadIdx.x] = j < n ? *(const float *) at(a, j, i1, i2, i3) : 0.f;
        __syncthreads();
        const int lim = (int) (n - base < BLOCK ? n - base : BLOCK);
        for (int t = 0; t < lim; t++) {
            const float vj = tile[t];
            const bool before = desc ? (vj > vi) : (vj < vi);
            rank += (before || (vj == vi && base + t < i)) ? 1 : 0;
        }
    }
    if (i < n) *(int32_t *) at(dst, rank, i1, i2, i3) = (int32_t) i;
}
void k_argsort(hipStream_t s, tdesc dst, tdesc a, int desc) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    if (rows == 0 || a.ne[0] == 0) return;
    argsort_kernel<<<dim3(nblocks(a.ne[0]), (unsigned) rows), BLOCK, 0, s>>>(dst, a, desc);
}

// w / b (optional, [n]): the weight product and the bias sum that follow a LayerNorm (ggml_norm -> ggml_mul -> ggml_add, one row) applied in the same
// launch - the same float operations in the same order as the three launches (y = (x - mean) * scale; y * w; + b)
__global__ void norm_kernel(tdesc dst, tdesc a, float eps, int rms, const float * w = nullptr, const float * b = nullptr, float * out = nullptr) {
    __shared__ double sh[BLOCK / 64];
    int64_t i1, i2, i3;
    row_coords(a, blockIdx.x, i1, i2, i3);
    const float * x = (const float *) at(a, 0, i1, i2, i3);
    float * y = (float *) at(dst, 0, i1, i2, i3);
    const int64_t n = a.ne[0];
    float mean = 0.f;
    if (!rms) {
        double acc = 0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += (double) x[i];
        mean = (float) (block_sum_f64(acc, sh) / (double) n);
    }
    double acc2 = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { const float v = x[i] - mean; acc2 += (double) (v * v); }
    const float var = (float) (block_sum_f64(acc2, sh) / (double) n);
    const float scale = 1.0f / sqrtf(var + eps);
    if (w == nullptr) { for (int64_t i = threadIdx.x; i < n; i += blockDim.x) y[i] = (x[i] - mean) * scale; return; }
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = (x[i] - mean) * scale;
        v = v * w[i];
        if (b) v = v + b[i];
        out[i] = v;
    }
}
void k_norm(hipStream_t s, tdesc dst, tdesc a, float eps, int rms) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    GGML_ASSERT(a.nb[0] == 4 && dst.nb[0] == 4);
    if (rows) norm_kernel<<<(int) rows, BLOCK, 0, s>>>(dst, a, eps, rms);
}
// one row: out = norm(a) * w (+ b)
void k_norm_affine(hipStream_t s, tdesc a, float eps, int rms, const float * w, const float * b, float * out) {
    GGML_ASSERT(a.nb[0] == 4 && a.ne[1] * a.ne[2] * a.ne[3] == 1);
    norm_kernel<<<1, BLOCK, 0, s>>>(a, a, eps, rms, w, b, out);
}

__global__ void soft_max_kernel(tdesc dst, tdesc a, tdesc mask, int has_mask, float scale) {
    __shared__ double shd[BLOCK / 64];
    __shared__ float shf[BLOCK / 64];
    int64_t i1, i2, i3;
    row_coords(a, blockIdx.x, i1, i2, i3);
    const float * x = (const float *) at(a, 0, i1, i2, i3);
    float * y = (float *) at(dst, 0, i1, i2, i3);
    const char * mp = has_mask ? at(mask, 0, i1, i2 % mask.ne[2], i3 % mask.ne[3]) : nullptr;
    const int64_t n = a.ne[0];
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = x[i] * scale;
        if (has_mask) v += mask.type == GGML_TYPE_F16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i];
        mx = fmaxf(mx, v);
    }
    mx = block_max_f32(mx, shf);
    double sum = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = x[i] * scale;
        if (has_mask) v += mask.type == GGML_TYPE_F16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i];
        const float e = expf(v - mx);
        y[i] = e;
        sum += (double) e;
    }
    sum = block_sum_f64(sum, shd);
    const float inv = (float) (1.0 / sum);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) y[i] *= inv;
}
void k_soft_max(hipStream_t s, tdesc dst, tdesc a, tdesc mask, int has_mask, float scale) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    if (rows) soft_max_kernel<<<(int) rows, BLOCK, 0, s>>>(dst, a, mask, has_mask, scale);
}

// ---------------------------------------------------------------------------------------------------
// gather / scatter of rows
// ---------------------------------------------------------------------------------------------------
__global__ void get_rows_kernel(tdesc dst, tdesc a, tdesc idx) {
    const int64_t i10 = blockIdx.x, i11 = blockIdx.y, i12 = blockIdx.z;
    const int64_t r = *(const int32_t *) at(idx, i10, i11, i12, 0);
    const char * row = at(a, 0, r, i11, i12);
    float * out = (float *) at(dst, 0, i10, i11, i12);
    const int64_t n = a.ne[0];
    if (r < 0 || r >= a.ne[1]) { for (int64_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = NAN; return; }
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        if (a.type == GGML_TYPE_I32) ((int32_t *) out)[i] = ((const int32_t *) row)[i];
        else out[i] = dequant_elem(row, a.type, i);
    }
}
void k_get_rows(hipStream_t s, tdesc dst, tdesc a, tdesc idx) {
    if (td_nelements(idx) == 0) return;
    get_rows_kernel<<<dim3((unsigned) idx.ne[0], (unsigned) idx.ne[1], (unsigned) idx.ne[2]), BLOCK, 0, s>>>(dst, a, idx);
}

__global__ void set_rows_kernel(tdesc dst, tdesc src, tdesc idx) {
    const int64_t i = blockIdx.x, i02 = blockIdx.y, i03 = blockIdx.z;
    const char * ip = at(idx, i, i02 % idx.ne[1], i03 % idx.ne[2], 0);
    const int64_t r = idx.type == GGML_TYPE_I64 ? *(const int64_t *) ip : (int64_t) *(const int32_t *) ip;
    if (r < 0 || r >= dst.ne[1]) return;
    const float * sp = (const float *) at(src, 0, i, i02, i03);
    char * out = at(dst, 0, r, i02, i03);
    const int es = elem_size(dst.type);
    for (int64_t k = threadIdx.x; k < src.ne[0]; k += blockDim.x) st_from_f32(out + k * es, dst.type, sp[k]);
}
void k_set_rows(hipStream_t s, tdesc dst, tdesc src, tdesc idx) {
    if (td_nelements(src) == 0) return;
    GGML_ASSERT(!ggml_is_quantized((enum ggml_type) dst.type));
    const int bs = src.ne[0] >= 256 ? 256 : 64;
    set_rows_kernel<<<dim3((unsigned) src.ne[1], (unsigned) src.ne[2], (unsigned) src.ne[3]), bs, 0, s>>>(dst, src, idx);
}

// ---------------------------------------------------------------------------------------------------
// convolution helpers and positional embedding
// ---------------------------------------------------------------------------------------------------
__global__ void im2col_kernel(tdesc dst, tdesc x, int64_t K, int s0, int p0, int d0, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t c, iow, in, z;
    unravel(dst, i, c, iow, in, z);
    const int64_t ic = c / K, ik = c % K;
    const int64_t iiw = iow * s0 + ik * d0 - p0;
    const float v = (iiw < 0 || iiw >= x.ne[0]) ? 0.f : *(const float *) at(x, iiw, ic, in, 0);
    st_from_f32(at(dst, c, iow, in, 0), dst.type, v);
}
void k_im2col(hipStream_t s, tdesc dst, tdesc x, int64_t K, int s0, int p0, int d0) {
    const int64_t n = td_nelements(dst);
    GGML_ASSERT(x.type == GGML_TYPE_F32);
    if (n) im2col_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, x, K, s0, p0, d0, n);
}

// conv_transpose_1d in two steps (the weight tensor is read exactly once, coalesced):
//  A. P[s][l][n] = sum over the s-th slice of ic of x[l, ic] * w[ic][n], n = oc*K + k (w's memory order), double partials
//  B. y[t, oc] = sum over l ascending of (float) sum_s P[s][l][oc*K + (t - l*s0)]  — the CPU reference's accumulation order
#define CT_LT 8
__global__ void __launch_bounds__(256) convtr_partial_kernel(tdesc w, tdesc x, double * P, int N, int L, int IC, int ic_per_split, int pre_elu) {
    extern __shared__ float ct_xs[];   // [ic in this split][CT_LT]: the activation tile, already rounded the way the dot sees it
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int l0 = blockIdx.y * CT_LT;
    const int split = blockIdx.z;
    const int ic0 = split * ic_per_split, ic1 = min(IC, ic0 + ic_per_split), nic = ic1 - ic0;
    const bool f16 = w.type == GGML_TYPE_F16;
    for (int idx = threadIdx.x; idx < nic * CT_LT; idx += blockDim.x) {
        const int i = idx % CT_LT, icl = idx / CT_LT, l = l0 + i;
        float xv = 0.f;
        if (l < L) {
            xv = *(const float *) (x.data + (int64_t) l * x.nb[0] + (int64_t) (ic0 + icl) * x.nb[1]);
            if (pre_elu) xv = xv > 0.f ? xv : expm1f(xv);
            if (f16) xv = h2f(f2h(xv));
        }
        ct_xs[idx] = xv;
    }
    __syncthreads();
    if (n >= N) return;
    double acc[CT_LT];
#pragma unroll
    for (int i = 0; i < CT_LT; i++) acc[i] = 0;
    const char * wp = w.data + (int64_t) n * w.nb[0] + (int64_t) ic0 * w.nb[2];   // (k, oc) are contiguous: n-th element of an ic slab
    int icl = 0;
    for (; icl + 8 <= nic; icl += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const char * wq = wp + (int64_t) (icl + u) * w.nb[2];
            wv[u] = f16 ? h2f(*(const uint16_t *) wq) : *(const float *) wq;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < CT_LT; i++) acc[i] += (double) (ct_xs[(icl + u) * CT_LT + i] * wv[u]);
    }
    for (; icl < nic; icl++) {
        const char * wq = wp + (int64_t) icl * w.nb[2];
        const float wv = f16 ? h2f(*(const uint16_t *) wq) : *(const float *) wq;
#pragma unroll
        for (int i = 0; i < CT_LT; i++) acc[i] += (double) (ct_xs[icl * CT_LT + i] * wv);
    }
#pragma unroll
    for (int i = 0; i < CT_LT; i++) { const int l = l0 + i; if (l < L) P[((int64_t) split * L + l) * N + n] = acc[i]; }
}
__global__ void convtr_overlap_add_kernel(tdesc dst, const double * P, int N, int L, int K, int s0, int nsplit, int64_t n_out) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int t = (int) (i % dst.ne[0]), oc = (int) (i / dst.ne[0]);
    float acc = 0.f;
    int l0 = (t - (K - 1) + s0 - 1) / s0;
    if (l0 < 0) l0 = 0;
    for (int l = l0; l < L && l * s0 <= t; l++) {
        const int k = t - l * s0;
        if (k >= K) continue;
        double v = 0;
        for (int sp = 0; sp < nsplit; sp++) v += P[((int64_t) sp * L + l) * N + oc * K + k];
        acc += (float) v;
    }
    *(float *) at(dst, t, oc, 0, 0) = acc;
}
size_t k_conv_transpose_1d_ws_size(const struct ggml_tensor * w, const struct ggml_tensor * x) {
    return (size_t) 16 * (size_t) x->ne[0] * (size_t) (w->ne[0] * w->ne[1]) * 8 + 256;
}
int k_conv_transpose_1d_partial(hipStream_t s, tdesc w, tdesc x, void * ws, int pre_elu) {
    GGML_ASSERT(x.type == GGML_TYPE_F32 && (w.type == GGML_TYPE_F32 || w.type == GGML_TYPE_F16));
    const int K = (int) w.ne[0], OC = (int) w.ne[1], IC = (int) w.ne[2], L = (int) x.ne[0];
    GGML_ASSERT(w.nb[1] == w.nb[0] * K && "kernel taps and output channels must be contiguous");
    const int N = OC * K;
    const int nb = (N + 255) / 256, lt = (L + CT_LT - 1) / CT_LT;
    int nsplit = 512 / (nb * lt);
    if (nsplit > 16) nsplit = 16;
    if (nsplit > IC / 32) nsplit = IC / 32;
    if (nsplit < 1) nsplit = 1;
    const int per = (IC + nsplit - 1) / nsplit;
    convtr_partial_kernel<<<dim3(nb, lt, nsplit), 256, (size_t) per * CT_LT * 4, s>>>(w, x, (double *) ws, N, L, IC, per, pre_elu);
    return nsplit;
}
void k_conv_transpose_1d(hipStream_t s, tdesc dst, tdesc w, tdesc x, int s0, void * ws) {
    const int nsplit = k_conv_transpose_1d_partial(s, w, x, ws, 0);
    const int K = (int) w.ne[0], OC = (int) w.ne[1], L = (int) x.ne[0];
    const int64_t n_out = dst.ne[0] * dst.ne[1];
    convtr_overlap_add_kernel<<<nblocks(n_out), BLOCK, 0, s>>>(dst, (const double *) ws, OC * K, L, K, s0, nsplit, n_out);
}

// ---- conv_scatter (hip_common.h): the next convolution's im2col panel written by this launch ----
// one finished output element (position l, channel c)
__device__ __forceinline__ void conv_scatter_elem(const conv_scatter & sc, int l, int c, float v) {
    if (sc.elu) v = v > 0.f ? v : expm1f(v);
    const uint16_t h = f2h(v);
    const int t = l + sc.TP;
    for (int kk = t % sc.s0; kk < sc.Kw; kk += sc.s0) {
        const int ow = (t - kk) / sc.s0;
        if (kk <= t && ow < sc.M) sc.panel[(int64_t) ow * sc.K + c * sc.Kw + kk] = h;
    }
}
// the panel columns that come from the consumer's carried tail: (ow, kk) with ow * s0 + kk < TP, every channel
__device__ __forceinline__ void conv_scatter_tail(const conv_scatter & sc) {
    const int n = sc.C * sc.TP;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = i / sc.TP, t = i - c * sc.TP;
        const uint16_t h = f2h(sc.prev[i]);
        for (int kk = t % sc.s0; kk < sc.Kw; kk += sc.s0) {
            const int ow = (t - kk) / sc.s0;
            if (kk <= t && ow < sc.M) sc.panel[(int64_t) ow * sc.K + c * sc.Kw + kk] = h;
        }
    }
}
// streaming tail of conv_transpose_1d: y = convtr(x); y[:PT] += prev[-PT:]; prev = y; out = (y + bias)[: len - PT]
// (moshi_streaming_conv_transpose_1d, conv.h:282-309). Thread t (< L*s0) also owns position t + L*s0 when t < PT, so the old
// tail value it needs is read before the same thread overwrites it.
// NS: the number of ic splits when it is a power of two <= 16 (the two-tap fast path then requests exactly 2 x 2 x NS partials: with NS fixed at 16 and the
// surplus clamped to split 0, the 480-frame layer's threads each read 64 doubles where 16 carry information and the kernel spilled registers); 0: any
template <int NS>
__global__ void convtr_finish_kernel(tdesc out, float * prev, const float * bias, const double * P, int K, int OC, int L, int s0, int nsplit, conv_scatter sc) {
    const int N = OC * K, OLf = (L - 1) * s0 + K, PT = K - s0, keep = OLf - PT;   // keep == L * s0
    if (sc.panel && sc.TP > 0 && blockIdx.x == gridDim.x - 1) { conv_scatter_tail(sc); return; }   // (the extra workgroup)
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) keep * OC) return;
    const int t = (int) (i % keep), oc = (int) (i / keep);
    auto conv_at = [&](int tt) {
        float acc = 0.f;
        int l0 = (tt - (K - 1) + s0 - 1) / s0;
        if (l0 < 0) l0 = 0;
        for (int l = l0; l < L && l * s0 <= tt; l++) {
            const int k = tt - l * s0;
            if (k >= K) continue;
            double v = 0;
#pragma unroll 8
            for (int sp = 0; sp < nsplit; sp++) v += P[((int64_t) sp * L + l) * N + oc * K + k];
            acc += (float) v;
        }
        return acc;
    };
    float * pv = prev + (int64_t) oc * OLf;
    const bool has2 = t < PT;               // position t + keep lies in the carried tail
    float y, y2 = 0.f;
    if (NS > 0 && K <= 2 * s0) {
        // at most two taps per output position: request every partial of both positions (2 x 2 taps x 16 splits) before the first
        // add - the partials were just written by other workgroups, so each dependent batch would cost a memory round trip
        double v[2][2][NS > 0 ? NS : 1];
        bool ok[2][2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int tt = p == 0 ? t : t + keep;
            int l0 = (tt - (K - 1) + s0 - 1) / s0;
            if (l0 < 0) l0 = 0;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int l = l0 + q, k = tt - l * s0;
                ok[p][q] = (p == 0 || has2) && l < L && k >= 0 && k < K;
                const int lc = ok[p][q] ? l : 0, kc = ok[p][q] ? k : 0;
#pragma unroll
                for (int sp = 0; sp < NS; sp++) v[p][q][sp] = P[((int64_t) sp * L + lc) * N + oc * K + kc];
            }
        }
        float r[2] = { 0.f, 0.f };
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                double sum = 0;
#pragma unroll
                for (int sp = 0; sp < NS; sp++) sum += v[p][q][sp];
                if (ok[p][q]) r[p] += (float) sum;   // taps in ascending l, splits in ascending order: as conv_at
            }
        y = r[0]; y2 = r[1];
        if (has2) y = y + pv[keep + t];
    } else {
        y = conv_at(t);
        if (has2) { y = y + pv[keep + t]; y2 = conv_at(t + keep); }
    }
    pv[t] = y;
    if (has2) pv[t + keep] = y2;
    const float yo = bias ? y + bias[oc] : y;
    *(float *) at(out, t, oc, 0, 0) = yo;
    if (sc.panel) conv_scatter_elem(sc, t, oc, yo);   // the next conv's im2col panel (hip_common.h conv_scatter)
}
void k_convtr_finish(hipStream_t s, tdesc out, float * prev, const float * bias, const void * ws, int K, int OC, int L, int s0, int nsplit, const conv_scatter * sc) {
    GGML_ASSERT(K - s0 <= L * s0 && "tail longer than the new window is not expected on this path");
    const int64_t n = (int64_t) L * s0 * OC;
    conv_scatter scv;
    memset(&scv, 0, sizeof(scv));
    if (sc) scv = *sc;
    const int nb = nblocks(n) + (scv.panel && scv.TP > 0 ? 1 : 0);
    switch (nsplit) {
        case 1:  convtr_finish_kernel<1><<<nb, BLOCK, 0, s>>>(out, prev, bias, (const double *) ws, K, OC, L, s0, nsplit, scv); break;
        case 2:  convtr_finish_kernel<2><<<nb, BLOCK, 0, s>>>(out, prev, bias, (const double *) ws, K, OC, L, s0, nsplit, scv); break;
        case 4:  convtr_finish_kernel<4><<<nb, BLOCK, 0, s>>>(out, prev, bias, (const double *) ws, K, OC, L, s0, nsplit, scv); break;
        case 8:  convtr_finish_kernel<8><<<nb, BLOCK, 0, s>>>(out, prev, bias, (const double *) ws, K, OC, L, s0, nsplit, scv); break;
        case 16: convtr_finish_kernel<16><<<nb, BLOCK, 0, s>>>(out, prev, bias, (const double *) ws, K, OC, L, s0, nsplit, scv); break;
        default: convtr_finish_kernel<0><<<nb, BLOCK, 0, s>>>(out, prev, bias, (const double *) ws, K, OC, L, s0, nsplit, scv); break;
    }
}

// depthwise variant for one input frame: y[k, c] = x[c] * w[k, c]; y[:PT] += prev[K-PT:]; prev = y; out = (y + bias)[:K-PT]
__global__ void dw_convtr_frame_kernel(float * out, float * prev, const float * bias, const char * x, int64_t x_cs, const char * w, int64_t w_cs, int K, int PT, int C, int64_t out_cs, int64_t out_ks) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float xv = *(const float *) (x + (int64_t) c * x_cs);
    const float * wc = (const float *) (w + (int64_t) c * w_cs);
    float * pv = prev + (int64_t) c * K;
    const int keep = K - PT;
    float y[16];
    for (int k = 0; k < K; k++) { y[k] = xv * wc[k]; if (k < PT) y[k] = y[k] + pv[keep + k]; }
    for (int k = 0; k < K; k++) pv[k] = y[k];
    for (int k = 0; k < keep; k++) out[(int64_t) c * out_cs + k * out_ks] = bias ? y[k] + bias[c] : y[k];
}
void k_dw_convtr_frame(hipStream_t s, float * out, float * prev, const float * bias, const char * x, int64_t x_cs, const char * w, int64_t w_cs, int K, int PT, int C, int64_t out_cs, int64_t out_ks) {
    GGML_ASSERT(K <= 16);
    dw_convtr_frame_kernel<<<(C + 63) / 64, 64, 0, s>>>(out, prev, bias, x, x_cs, w, w_cs, K, PT, C, out_cs < 0 ? K - PT : out_cs, out_ks);
}

// F16 im2col of concat(prev, act(x)) without materialising the concat: dst[ci*Kw + k, ol] = xc[ol*s0 + k, ci]
__global__ void stream_im2col_kernel(tdesc dst, const float * prev, int TP, tdesc x, int Kw, int s0, int pre_elu, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t CK = (uint32_t) dst.ne[0];
    const uint32_t ol = (uint32_t) i / CK, c = (uint32_t) i - ol * CK;
    const uint32_t ci = c / (uint32_t) Kw, k = c - ci * (uint32_t) Kw;
    const int l = (int) (ol * (uint32_t) s0 + k);
    float v;
    if (l < TP) v = prev[l + (int64_t) ci * TP];
    else { v = *(const float *) (x.data + (int64_t) (l - TP) * x.nb[0] + (int64_t) ci * x.nb[1]); if (pre_elu) v = v > 0.f ? v : expm1f(v); }
    ((uint16_t *) dst.data)[i] = f2h(v);
}
void k_stream_im2col(hipStream_t s, tdesc dst, const float * prev, int TP, tdesc x, int Kw, int s0, int pre_elu) {
    const int64_t n = dst.ne[0] * dst.ne[1];
    GGML_ASSERT(dst.type == GGML_TYPE_F16 && x.type == GGML_TYPE_F32 && dst.nb[1] == dst.ne[0] * 2);
    if (n) stream_im2col_kernel<<<nblocks(n), BLOCK, 0, s>>>(dst, prev, TP, x, Kw, s0, pre_elu, n);
}
// prev <- last TP samples of concat(prev, act(x)), one thread per channel (reads complete before its writes)
__global__ void conv_tail_kernel(float * prev, int TP, tdesc x, int pre_elu) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= (int) x.ne[1]) return;
    const int L = (int) x.ne[0];
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (j < TP) {
            const int l = L + j;   // index into the concatenation
            if (l < TP) v[j] = prev[l + (int64_t) ci * TP];
            else { float t = *(const float *) (x.data + (int64_t) (l - TP) * x.nb[0] + (int64_t) ci * x.nb[1]); v[j] = pre_elu ? (t > 0.f ? t : expm1f(t)) : t; }
        }
    }
#pragma unroll
    for (int j = 0; j < 32; j++) if (j < TP) prev[j + (int64_t) ci * TP] = v[j];
}
void k_conv_tail(hipStream_t s, float * prev, int TP, tdesc x, int pre_elu) {
    GGML_ASSERT(TP <= 32);
    if (TP > 0) conv_tail_kernel<<<(int) ((x.ne[1] + 63) / 64), 64, 0, s>>>(prev, TP, x, pre_elu);
}

// addend (optional): the timesteps are ts[i] + addend[i or 0] - the `add(arange, offset)` in front of the node (moshi_get_timestep_embedding, rope.h:8-20) folded in
__global__ void timestep_embedding_kernel(tdesc dst, tdesc ts, int dim, int max_period, const float * addend, int addend_n) {
    const int64_t i = blockIdx.x;
    const int half = dim / 2;
    float * e = (float *) at(dst, 0, i, 0, 0);
    float t = *(const float *) at(ts, i, 0, 0, 0);
    if (addend) t = t + addend[addend_n == 1 ? 0 : i];
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        // float transcendentals evaluated in double and rounded once: the 1-ulp differences between the
        // device's expf and a host libm would otherwise be amplified by t (up to the context length)
        const float lf = (float) log((double) max_period);
        const float freq = (float) exp((double) (-lf * j / half));
        const float arg = t * freq;
        e[j] = (float) cos((double) arg);
        e[j + half] = (float) sin((double) arg);
    }
    if (threadIdx.x == 0 && (dim & 1)) e[2 * half] = 0.f;
}
void k_timestep_embedding(hipStream_t s, tdesc dst, tdesc ts, int dim, int max_period, const float * addend, int addend_n) {
    if (ts.ne[0]) timestep_embedding_kernel<<<(int) ts.ne[0], 64, 0, s>>>(dst, ts, dim, max_period, addend, addend_n);
}

// ---------------------------------------------------------------------------------------------------
// generic matrix product
// ---------------------------------------------------------------------------------------------------
static int vec_dot_type(int t) {
    switch (t) {
        case GGML_TYPE_F32: return GGML_TYPE_F32;
        case GGML_TYPE_F16: return GGML_TYPE_F16;
        case GGML_TYPE_BF16: return GGML_TYPE_BF16;
        case GGML_TYPE_Q4_0: case GGML_TYPE_Q8_0: return GGML_TYPE_Q8_0;
        case GGML_TYPE_Q4_K: return GGML_TYPE_Q8_K;
        default: GGML_ABORT("mul_mat: unsupported weight type %s", ggml_type_name((enum ggml_type) t));
    }
}

size_t k_mul_mat_ws_size(const struct ggml_tensor * a, const struct ggml_tensor * b) {
    const int vt = vec_dot_type(a->type);
    return ggml_row_size((enum ggml_type) vt, b->ne[0]) * (size_t) (b->ne[1] * b->ne[2] * b->ne[3]) + 256;
}

// converts one activation row per block to the dot type (dense rows in ws)
__global__ void convert_rows_kernel(tdesc b, int vt, char * ws, int64_t row_bytes) {
    __shared__ float shf[BLOCK / 64];
    int64_t i1, i2, i3;
    row_coords(b, blockIdx.x, i1, i2, i3);
    char * out = ws + (int64_t) blockIdx.x * row_bytes;
    const int64_t K = b.ne[0];
    if (vt == GGML_TYPE_F32 || vt == GGML_TYPE_F16 || vt == GGML_TYPE_BF16) {
        const int es = elem_size(vt);
        for (int64_t i = threadIdx.x; i < K; i += blockDim.x) st_from_f32(out + i * es, vt, ld_as_f32(at(b, i, i1, i2, i3), b.type));
        return;
    }
    if (vt == GGML_TYPE_Q8_0) {
        // one wave-lane group of 32 per block: thread t handles element t of block (t/32)
        for (int64_t base = 0; base < K; base += blockDim.x) {
            const int64_t i = base + threadIdx.x;
            const float v = i < K ? *(const float *) at(b, i, i1, i2, i3) : 0.f;
            float amax = fabsf(v);
            for (int o = 16; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
            const float d = amax / 127.f;
            const float id = d ? 1.0f / d : 0.0f;
            if (i < K) {
                block_q8_0 * blk = (block_q8_0 *) out + i / 32;
                blk->qs[i % 32] = (int8_t) roundf(v * id);
                if (i % 32 == 0) blk->d = f2h(d);
            }
        }
        return;
    }
    // Q8_K: 256 threads = one super-block per iteration
    for (int64_t base = 0; base < K; base += 256) {
        const int tid = threadIdx.x;
        const float v = *(const float *) at(b, base + tid, i1, i2, i3);
        // signed value of the element with the largest magnitude (first one on ties)
        float amax = fabsf(v); int ai = tid;
        for (int o = 32; o > 0; o >>= 1) {
            const float oa = __shfl_xor(amax, o, 64); const int oi = __shfl_xor(ai, o, 64);
            if (oa > amax || (oa == amax && oi < ai)) { amax = oa; ai = oi; }
        }
        __shared__ float s_amax[4]; __shared__ int s_ai[4]; __shared__ float s_vals[256]; __shared__ int s_q[256];
        __syncthreads();
        s_vals[tid] = v;
        if ((tid & 63) == 0) { s_amax[tid >> 6] = amax; s_ai[tid >> 6] = ai; }
        __syncthreads();
        float bm = s_amax[0]; int bi = s_ai[0];
        for (int w = 1; w < 4; w++) if (s_amax[w] > bm || (s_amax[w] == bm && s_ai[w] < bi)) { bm = s_amax[w]; bi = s_ai[w]; }
        block_q8_K * blk = (block_q8_K *) out + base / 256;
        int q = 0;
        float dd = 0.f;
        if (bm != 0.f) {
            const float iscale = -127.f / s_vals[bi];
            q = nearest_int_dev(iscale * v);
            q = q < 127 ? q : 127;
            dd = 1.f / iscale;
        }
        blk->qs[tid] = (int8_t) q;
        s_q[tid] = q;
        __syncthreads();
        if (tid < 16) { int sum = 0; for (int l = 0; l < 16; l++) sum += s_q[tid * 16 + l]; blk->bsums[tid] = (int16_t) sum; }
        if (tid == 0) blk->d = dd;
        (void) shf;
    }
}

// streaming-conv tail update folded into the product launch: prev <- last TP positions of (prev | act(x)), per channel. An EXTRA workgroup at the end of the
// grid does it beside the tiles (the operand was materialised before this launch: nobody reads the old tail any more), element (channel, j) per thread
// through LDS in rounds of whole channels - loads (no branch in front of any: the per-channel loop of 32 conditional loads this replaces was a chain of up
// to TP serial round trips on workgroup 0, 8-12 us on the launch's critical path), barrier, ELU where the element is a new sample, stores.
#define MM_TAIL_LDS 4096
__device__ __forceinline__ void mm_tail_update(const mm_epilogue & e) {
    __shared__ float mm_tail_buf[MM_TAIL_LDS];
    const int TP = e.tail_TP, L = e.tail_L, C = e.tail_C, cpr = MM_TAIL_LDS / TP;
    for (int c0 = 0; c0 < C; c0 += cpr) {
        const int n = (C - c0 < cpr ? C - c0 : cpr) * TP;
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int cl = i / TP, j = i - cl * TP, l = L + j;   // index into the concatenation (prev | x)
            const int64_t ci = c0 + cl;
            const float * src = l < TP ? e.tail_prev + l + ci * TP : (const float *) (e.tail_x + (int64_t) (l - TP) * e.tail_nb0 + ci * e.tail_nb1);
            mm_tail_buf[i] = *src;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int cl = i / TP, j = i - cl * TP;
            float v = mm_tail_buf[i];
            if (e.tail_pre_elu && L + j >= TP) v = v > 0.f ? v : expm1f(v);
            e.tail_prev[(int64_t) c0 * TP + i] = v;
        }
        __syncthreads();
    }
}
// true: this workgroup is the extra one (it has done the side jobs and must leave)
__device__ __forceinline__ bool mm_tail_block(const mm_epilogue & e) {
    const bool tail = e.tail_prev != nullptr, stail = e.sc.panel != nullptr && e.sc.TP > 0;
    if (!(tail || stail) || blockIdx.x != gridDim.x - 1) return false;
    if (stail) conv_scatter_tail(e.sc);
    if (tail) mm_tail_update(e);
    return true;
}
static int mm_tail_extra(const mm_epilogue & e) { return e.tail_prev != nullptr || (e.sc.panel != nullptr && e.sc.TP > 0) ? 1 : 0; }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void mul_mat_kernel(tdesc dst, tdesc a, tdesc b, const char * ws, int64_t ws_row_bytes, int vt, int64_t total, mm_epilogue epi) {
    if (mm_tail_block(epi)) return;
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= total) return;
    int64_t m, n, i2, i3;
    unravel(dst, o, m, n, i2, i3);
    const int64_t r2 = b.ne[2] / a.ne[2], r3 = b.ne[3] / a.ne[3];
    const char * w = at(a, 0, m, i2 / r2, i3 / r3);
    const char * y = ws + (n + b.ne[1] * (i2 + b.ne[2] * i3)) * ws_row_bytes;
    const int64_t K = a.ne[0];
    float result;
    if (a.type == GGML_TYPE_F32 || a.type == GGML_TYPE_F16 || a.type == GGML_TYPE_BF16) {
        double acc = 0;
        if (a.type == GGML_TYPE_F32)       for (int64_t k = lane; k < K; k += 64) acc += (double) (((const float *) w)[k] * ((const float *) y)[k]);
        else if (a.type == GGML_TYPE_F16)  for (int64_t k = lane; k < K; k += 64) acc += (double) (h2f(((const uint16_t *) w)[k]) * h2f(((const uint16_t *) y)[k]));
        else                               for (int64_t k = lane; k < K; k += 64) acc += (double) (bf2f(((const uint16_t *) w)[k]) * bf2f(((const uint16_t *) y)[k]));
        result = (float) wave_sum_f64(acc);
    } else if (a.type == GGML_TYPE_Q8_0) {
        float acc = 0;
        for (int64_t ib = lane; ib < K / 32; ib += 64) {
            const block_q8_0 * xb = (const block_q8_0 *) w + ib; const block_q8_0 * yb = (const block_q8_0 *) y + ib;
            int sumi = 0;
            for (int j = 0; j < 32; j++) sumi += xb->qs[j] * yb->qs[j];
            acc += sumi * (h2f(xb->d) * h2f(yb->d));
        }
        result = wave_sum_f32(acc);
    } else if (a.type == GGML_TYPE_Q4_0) {
        float acc = 0;
        for (int64_t ib = lane; ib < K / 32; ib += 64) {
            const block_q4_0 * xb = (const block_q4_0 *) w + ib; const block_q8_0 * yb = (const block_q8_0 *) y + ib;
            int sumi = 0;
            for (int j = 0; j < 16; j++) {
                const int v0 = (xb->qs[j] & 0x0F) - 8, v1 = (xb->qs[j] >> 4) - 8;
                sumi += v0 * yb->qs[j] + v1 * yb->qs[j + 16];
            }
            acc += sumi * h2f(xb->d) * h2f(yb->d);
        }
        result = wave_sum_f32(acc);
    } else {   // Q4_K x Q8_K
        float acc = 0;
        for (int64_t ib = lane; ib < K / 256; ib += 64) {
            const block_q4_K * xb = (const block_q4_K *) w + ib; const block_q8_K * yb = (const block_q8_K *) y + ib;
            acc += q4k_q8k_block_dot(xb, yb->qs, yb->bsums, yb->d);
        }
        result = wave_sum_f32(acc);
    }
    if (lane == 0) {
        if (epi.bias) result = result + epi.bias[n];
        if (epi.residual) result = *(const float *) (epi.residual + m * epi.res_nb0 + n * epi.res_nb1) + result;
        *(float *) at(dst, m, n, i2, i3) = result;
        if (epi.sc.panel) conv_scatter_elem(epi.sc, (int) m, (int) n, result);
    }
}

// ---- dense f16 x f16 contraction on the matrix cores (Mimi conv stacks: dst[ow, co] = sum_k A[ow][k] * B[co][k]) ----
// Both operands are K-contiguous, which is exactly the 16x16x32 MFMA fragment order: lane l holds 8 consecutive k
// of row (l & 15), so fragments are plain 16-byte global loads - no LDS, one wave per 16x16 output tile.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Few tiles (deep conv layers at 12.5-50 Hz) would leave the chip empty and the K loop latency-bound, so a tile can be split
// over SK waves of the workgroup (contiguous K slices, partial tiles summed through LDS in slice order); the K loop requests
// four fragment pairs before the first MFMA.
__global__ void __launch_bounds__(1024) mul_mat_f16_mfma_kernel(tdesc dst, tdesc a, tdesc b, int mt, int nt, int SK, int TPW, mm_epilogue epi) {
    __shared__ f32x4 red[16][64];
    if (mm_tail_block(epi)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slice = wave % SK, tw = wave / SK;
    const int tile = blockIdx.x * TPW + tw;
    const bool live = tile < mt * nt;
    const int tm = live ? tile % mt : 0, tn = live ? tile / mt : 0;
    const int r = lane & 15, kq = lane >> 4;
    const int K = (int) a.ne[0];
    const int steps = K / 32, per = (steps + SK - 1) / SK;
    const int s_begin = slice * per, s_end = min(steps, s_begin + per);
    const char * ap = a.data + (int64_t) (tm * 16 + r) * a.nb[1] + kq * 16;
    const char * bp = b.data + (int64_t) (tn * 16 + r) * b.nb[1] + kq * 16;
    f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
    int st = s_begin;
    for (; st + 4 <= s_end; st += 4) {
        f16x8 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { av[u] = *(const f16x8 *) (ap + (st + u) * 64); bv[u] = *(const f16x8 *) (bp + (st + u) * 64); }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[u], bv[u], acc, 0, 0, 0);
    }
    for (; st < s_end; st++) {
        const f16x8 av = *(const f16x8 *) (ap + st * 64);
        const f16x8 bv = *(const f16x8 *) (bp + st * 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    }
    bool writer = live;
    if (SK > 1) {
        if (slice > 0) red[wave][lane] = acc;
        __syncthreads();
        if (slice > 0) writer = false;
        else for (int sl = 1; sl < SK; sl++) { const f32x4 o = red[wave + sl][lane]; acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; }
    }
    if (writer) {
        // C[row = 4*(lane>>4) + j][col = lane & 15]: rows are `a` rows (dst dim 0), cols are `b` rows (dst dim 1)
        float * out = (float *) (dst.data + (int64_t) (tn * 16 + r) * dst.nb[1]) + tm * 16 + kq * 4;
        if (epi.bias) { const float bv = epi.bias[tn * 16 + r]; acc[0] = acc[0] + bv; acc[1] = acc[1] + bv; acc[2] = acc[2] + bv; acc[3] = acc[3] + bv; }
        if (epi.residual) {
            const char * rp = epi.residual + (int64_t) (tn * 16 + r) * epi.res_nb1 + (int64_t) (tm * 16 + kq * 4) * epi.res_nb0;
    #pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = *(const float *) (rp + (int64_t) j * epi.res_nb0) + acc[j];
        }
        *(f32x4 *) out = acc;
        if (epi.sc.panel) {
#pragma unroll
            for (int j = 0; j < 4; j++) conv_scatter_elem(epi.sc, tm * 16 + kq * 4 + j, tn * 16 + r, acc[j]);
        }
    }
}

// few activation rows (M = a.ne1 <= 8) against many weight rows in `b`: one wave per b row, b read once. The M activation rows (M x K halves, the im2col
// matrix) are staged in LDS first, by loads that are all in flight together: read from memory
// chunk by chunk inside the dot loop, behind `if (m < M)`, each was a serial round trip to a matrix another launch had just written (48 us for the encoder's
// 512 -> 1024 stride-8 conv, r05 kernel trace). The row's first 8 weight chunks are requested before the staging. Per lane the products are added in the old
// order (chunk c = k / 512 ascending, 8 halves each, in double; chunks past K contribute nothing), then the wave sum.
#define SMALLM_NW 8
__global__ void __launch_bounds__(SMALLM_NW * 64) mul_mat_smallm_kernel(tdesc dst, tdesc a, tdesc b, int M, int N, mm_epilogue epi) {
    extern __shared__ __attribute__((aligned(16))) char smallm_lds[];
    _Float16 * As = (_Float16 *) smallm_lds;   // [M][K]
    if (mm_tail_block(epi)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * SMALLM_NW + wave, nc = n < N ? n : N - 1;
    const int K = (int) a.ne[0], nchunks = (K + 511) / 512, K8 = K / 8;
    const char * bp = b.data + (int64_t) nc * b.nb[1];
    f16x8 bv[8];
    auto request = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int kk = (c0 + u) * 512 + lane * 8; bv[u] = *(const f16x8 *) (bp + (kk < K ? kk : 0) * 2); }   // (no load behind a branch; chunks past K are skipped below and re-read the row's first, in-bounds 16 bytes)
    };
    request(0);
    auto stage = [&](auto fetch) {
        for (int i0 = threadIdx.x; i0 < M * K8; i0 += SMALLM_NW * 64 * 4) {
            f16x8 t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = i0 + u * SMALLM_NW * 64;
                const int ic = idx < M * K8 ? idx : M * K8 - 1, m = ic / K8, k8 = ic - m * K8;
                t[u] = fetch(m, k8);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int idx = i0 + u * SMALLM_NW * 64; if (idx < M * K8) *(f16x8 *) (As + (int64_t) idx * 8) = t[u]; }
        }
    };
    if (epi.af_x) {   // (rows converted from F32 here: eight unconditional loads per fragment)
        if (epi.af_elu) stage([&](int m, int k8) { f16x8 r; float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = *(const float *) (epi.af_x + (int64_t) m * epi.af_nb0 + (int64_t) (k8 * 8 + j) * epi.af_nb1);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = (_Float16) (v[j] > 0.f ? v[j] : expm1f(v[j]));
            return r; });
        else stage([&](int m, int k8) { f16x8 r; float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = *(const float *) (epi.af_x + (int64_t) m * epi.af_nb0 + (int64_t) (k8 * 8 + j) * epi.af_nb1);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = (_Float16) v[j];
            return r; });
    } else
    stage([&](int m, int k8) { return *(const f16x8 *) (a.data + (int64_t) m * a.nb[1] + k8 * 16); });
    __syncthreads();
    double acc[8];
#pragma unroll
    for (int m = 0; m < 8; m++) acc[m] = 0;
    for (int c0 = 0; c0 < nchunks; c0 += 8) {
        if (c0 > 0) request(c0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int kk = (c0 + u) * 512 + lane * 8;
            if (kk < K) {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    if (m < M) {
                        const f16x8 av = *(const f16x8 *) (As + (int64_t) m * K + kk);
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[m] += (double) ((float) av[j] * (float) bv[u][j]);
                    }
                }
            }
        }
    }
    if (n < N) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (m < M) {
                const double v = wave_sum_f64(acc[m]);
                if (lane == 0) {
                    float r = (float) v;
                    if (epi.bias) r = r + epi.bias[n];
                    if (epi.residual) r = *(const float *) (epi.residual + (int64_t) m * epi.res_nb0 + (int64_t) n * epi.res_nb1) + r;
                    *(float *) (dst.data + (int64_t) m * dst.nb[0] + (int64_t) n * dst.nb[1]) = r;
                    if (epi.sc.panel) conv_scatter_elem(epi.sc, m, n, r);
                }
            }
        }
    }
}

// short rows (K <= 64, e.g. the first SEANet conv: 1 input channel x 7 taps): one thread per output element
__global__ void mul_mat_f16_shortk_kernel(tdesc dst, tdesc a, tdesc b, int K, int M, int64_t total, mm_epilogue epi) {
    if (mm_tail_block(epi)) return;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const uint32_t n = (uint32_t) i / (uint32_t) M, m = (uint32_t) i - n * (uint32_t) M;
        const uint16_t * ap = (const uint16_t *) (a.data + (int64_t) m * a.nb[1]);
        const uint16_t * bp = (const uint16_t *) (b.data + (int64_t) n * b.nb[1]);
        double acc = 0;
        for (int k = 0; k < K; k++) acc += (double) (h2f(ap[k]) * h2f(bp[k]));
        float r = (float) acc;
        if (epi.bias) r = r + epi.bias[n];
        if (epi.residual) r = *(const float *) (epi.residual + (int64_t) m * epi.res_nb0 + (int64_t) n * epi.res_nb1) + r;
        *(float *) (dst.data + (int64_t) m * dst.nb[0] + (int64_t) n * dst.nb[1]) = r;
        if (epi.sc.panel) conv_scatter_elem(epi.sc, (int) m, (int) n, r);
    }
}

// One embedding row through a small Q8_0 projection (tts: the Depth transformer's low-rank embeddings, lm_utils.h:157-217 - get_rows of a 128-wide table,
// a 128 -> 1024 linear, a cast): get_rows_kernel + convert_rows_kernel + mul_mat_kernel (+ the F32 -> F32 copy of the cast) as ONE launch with their
// arithmetic - the row dequantised as get_rows does, re-quantised to Q8_0 blocks as convert_rows does (d = amax / 127 kept as F16, q = roundf(x / d)),
// a wave per output element with block ib on lane ib, the blocks' terms added in block order (vec_dot_q8_0_q8_0: bit-exact against the oracle, which the
// generic kernel's butterfly sum is only to 1 ulp).
__global__ void __launch_bounds__(256) lowrank_embed_kernel(lowrank_embed_args a) {
    const int lane = threadIdx.x & 63;
    const int m = (int) blockIdx.x * 4 + (int) (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int64_t r = *a.index;
    if (r < 0 || r >= a.n_rows) { if (lane == 0) a.out[m] = NAN; return; }   // (get_rows_kernel hands on a NaN row)
    const char * row = a.table + r * a.row_bytes;
    const char * w = a.w + (int64_t) m * a.w_row_bytes;
    float term = 0.f;   // (K <= 2048: at most one block per lane)
    for (int ib = lane; ib < a.K / 32; ib += 64) {
        float xv[32], amax = 0.f;
#pragma unroll
        for (int j = 0; j < 32; j++) { xv[j] = dequant_elem(row, a.type, (int64_t) ib * 32 + j); amax = fmaxf(amax, fabsf(xv[j])); }
        const float d = amax / 127.f;
        const float id = d ? 1.0f / d : 0.0f;
        const block_q8_0 * xb = (const block_q8_0 *) w + ib;
        int sumi = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) sumi += xb->qs[j] * (int) (int8_t) roundf(xv[j] * id);
        term = sumi * (h2f(xb->d) * h2f(f2h(d)));
    }
    // vec_dot_q8_0_q8_0's order: sumf = 0; sumf += term_ib for ib = 0, 1, ... (mul_mat_kernel adds the lanes' terms as a butterfly, which differs in the
    // last bit now and then; this launch keeps the reference's order)
    float sumf = 0.f;
    for (int ib = 0; ib < a.K / 32; ib++) sumf += __shfl(term, ib, 64);
    if (lane == 0) a.out[m] = sumf;
}
void k_lowrank_embed(hipStream_t s, const lowrank_embed_args & a) { lowrank_embed_kernel<<<(a.M + 3) / 4, 256, 0, s>>>(a); }

static int mfma_split_k(int tiles, int steps) {
    int SK = 1;
    static const int sk_target = getenv("MI355X_MFMA_SK_TARGET") ? atoi(getenv("MI355X_MFMA_SK_TARGET")) : 512;
    while (SK < 16 && tiles * SK < sk_target && steps / (SK * 2) >= 4) SK *= 2;
    return SK;
}
// true: k_mul_mat runs this product as mul_mat_smallm_kernel (the only form that takes mm_epilogue::af_x)
bool k_mul_mat_is_few_rows(tdesc a, tdesc b) {
    return a.type == GGML_TYPE_F16 && b.type == GGML_TYPE_F16 && a.nb[0] == 2 && b.nb[0] == 2 && a.ne[2] * a.ne[3] * b.ne[2] * b.ne[3] == 1 && !(a.ne[0] < 32 && a.ne[1] >= 256) &&
           a.ne[0] % 8 == 0 && (a.nb[1] % 16) == 0 && (b.nb[1] % 16) == 0 && ((uintptr_t) a.data % 16) == 0 && ((uintptr_t) b.data % 16) == 0 &&
           a.ne[1] <= 8 && (size_t) a.ne[1] * a.ne[0] * 2 <= 128 * 1024;
}
void k_mul_mat(hipStream_t s, tdesc dst, tdesc a, tdesc b, void * ws, const mm_epilogue * epi_) {
    const int64_t total = td_nelements(dst);
    if (total == 0) return;
    mm_epilogue epi;
    memset(&epi, 0, sizeof(epi));
    if (epi_) epi = *epi_;
    if (a.type == GGML_TYPE_F16 && b.type == GGML_TYPE_F16 && a.nb[0] == 2 && b.nb[0] == 2 && a.ne[2] * a.ne[3] * b.ne[2] * b.ne[3] == 1 &&
        a.ne[0] < 32 && a.ne[1] >= 256) {
        mul_mat_f16_shortk_kernel<<<nblocks(total) + mm_tail_extra(epi), BLOCK, 0, s>>>(dst, a, b, (int) a.ne[0], (int) a.ne[1], total, epi);
        return;
    }
    if (a.type == GGML_TYPE_F16 && b.type == GGML_TYPE_F16 && a.nb[0] == 2 && b.nb[0] == 2 && a.ne[2] * a.ne[3] * b.ne[2] * b.ne[3] == 1 &&
        a.ne[0] % 8 == 0 && (a.nb[1] % 16) == 0 && (b.nb[1] % 16) == 0 && ((uintptr_t) a.data % 16) == 0 && ((uintptr_t) b.data % 16) == 0) {
        const int M = (int) a.ne[1], N = (int) b.ne[1];
        if (dst.nb[0] != 4 && !(M <= 8 && (size_t) M * a.ne[0] * 2 <= 128 * 1024)) goto generic_product;   // (only the few-row kernel stores through any destination strides)
        if (M <= 8 && (size_t) M * a.ne[0] * 2 <= 128 * 1024) {
            const size_t lds = (size_t) M * a.ne[0] * 2;
            static bool granted = false;
            if (!granted) { HIP_CHECK(hipFuncSetAttribute((const void *) mul_mat_smallm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); granted = true; }
            mul_mat_smallm_kernel<<<(N + SMALLM_NW - 1) / SMALLM_NW + mm_tail_extra(epi), SMALLM_NW * 64, lds, s>>>(dst, a, b, M, N, epi);
            return;
        }
        if (M % 16 == 0 && N % 16 == 0 && a.ne[0] % 32 == 0 && (dst.nb[1] % 16) == 0 && ((uintptr_t) dst.data % 16) == 0) {
            const int mt = M / 16, nt = N / 16, tiles = mt * nt, steps = (int) a.ne[0] / 32;
            const int SK = mfma_split_k(tiles, steps);
            const int TPW = SK >= 4 ? 1 : 4 / SK;
            mul_mat_f16_mfma_kernel<<<(tiles + TPW - 1) / TPW + mm_tail_extra(epi), 64 * SK * TPW, 0, s>>>(dst, a, b, mt, nt, SK, TPW, epi);
            return;
        }
    }
generic_product:
    const int vt = vec_dot_type(a.type);
    GGML_ASSERT(a.nb[0] == (int64_t) ggml_type_size((enum ggml_type) a.type));
    GGML_ASSERT(b.type == GGML_TYPE_F32 || b.type == vt);
    if (vt == GGML_TYPE_Q8_0 || vt == GGML_TYPE_Q8_K) GGML_ASSERT(b.type == GGML_TYPE_F32);
    if (vt == GGML_TYPE_Q8_K) GGML_ASSERT(a.ne[0] % 256 == 0);
    const int64_t row_bytes = (int64_t) ggml_row_size((enum ggml_type) vt, b.ne[0]);
    const int64_t rows = b.ne[1] * b.ne[2] * b.ne[3];
    const char * y = (const char *) ws;
    const bool direct = b.type == vt && b.nb[0] == (int64_t) ggml_type_size((enum ggml_type) vt) && b.nb[1] == row_bytes &&
                        b.nb[2] == row_bytes * b.ne[1] && b.nb[3] == b.nb[2] * b.ne[2];
    if (direct) y = b.data;
    else convert_rows_kernel<<<(int) rows, BLOCK, 0, s>>>(b, vt, (char *) ws, row_bytes);
    mul_mat_kernel<<<nblocks(total, 4) + mm_tail_extra(epi), 256, 0, s>>>(dst, a, b, y, row_bytes, vt, total, epi);
}

// ---------------------------------------------------------------------------------------------------
// batched small uploads
// ---------------------------------------------------------------------------------------------------
__global__ void scatter_uploads_kernel(const upload_desc * descs, const char * blob) {
    const upload_desc d = descs[blockIdx.x];
    const char * src = blob + d.offset;   // 16-byte aligned inside the blob (queue_upload pads)
    if (((uintptr_t) d.dst & 15) == 0 && (d.size & 15) == 0) {   // e.g. the 7.7 KB PCM frame: two 16-byte rounds of 256 threads
        for (uint32_t i = threadIdx.x; i < d.size / 16; i += blockDim.x) ((uint4 *) d.dst)[i] = ((const uint4 *) src)[i];
    } else if (((uintptr_t) d.dst & 3) == 0 && (d.size & 3) == 0) {
        for (uint32_t i = threadIdx.x; i < d.size / 4; i += blockDim.x) ((uint32_t *) d.dst)[i] = ((const uint32_t *) src)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < d.size; i += blockDim.x) d.dst[i] = src[i];
    }
}
void k_scatter_uploads(hipStream_t s, const upload_desc * descs, const char * blob, int n) {
    if (n > 0) scatter_uploads_kernel<<<n, 256, 0, s>>>(descs, blob);
}
