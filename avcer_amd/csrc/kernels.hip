// Element-wise, normalisation, pooling, attention and fusion kernels of the AVCER hot path (gfx950).
// All of these are HBM/LDS-bound: 16-byte vector accesses, one 64-lane wave per row for reductions,
// f32 statistics regardless of the activation storage type.
#include "common.h"
#include "split_dev.h"

#include <cstdlib>
#include <type_traits>

namespace {

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf from Abramowitz & Stegun 7.1.26: see gemm_dev.h gelu_fast (same code; |gelu_fast - gelu| <= 4.7e-7 on [-8, 8])
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f, a = __builtin_fabsf(z);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, a, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * a);
    const float erfz = __builtin_copysignf(__builtin_fmaf(-p, e, 1.0f), z);
    return 0.5f * x * (1.0f + erfz);
}
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }  // keeps NaN like torch; two instructions (gemm_dev.h)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <typename T> __device__ __forceinline__ float ldf(const T* p, long i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }
// `ovf`: the context's range-contract counter (split_dev.h sp_commit); only the sp32 forms look at it
template <typename T> __device__ __forceinline__ void stf(T* p, long i, float v, unsigned* ovf = nullptr);
template <> __device__ __forceinline__ void stf<float>(float* p, long i, float v, unsigned*) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, long i, float v, unsigned*) { p[i] = f2bf(v); }

// load / store 4 consecutive elements
template <typename T> __device__ __forceinline__ void ld4(const T* p, long i, float* v);
template <> __device__ __forceinline__ void ld4<float>(const float* p, long i, float* v) {
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, long i, float* v) {
    const uint2 t = *reinterpret_cast<const uint2*>(p + i);
    v[0] = bf2f((bf16_t)(t.x & 0xffff)); v[1] = bf2f((bf16_t)(t.x >> 16));
    v[2] = bf2f((bf16_t)(t.y & 0xffff)); v[3] = bf2f((bf16_t)(t.y >> 16));
}
template <typename T> __device__ __forceinline__ void st4(T* p, long i, const float* v, unsigned* ovf = nullptr);
template <> __device__ __forceinline__ void st4<float>(float* p, long i, const float* v, unsigned*) {
    *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, long i, const float* v, unsigned*) {
    uint2 t;
    t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p + i) = t;
}

// sp32 storage (AVCER_MODE_F16X3 activations): per aligned group of 32 channels, 32 fp16 hi then 32 fp16 lo (split_dev.h),
// x = hi + lo.  4 bytes per element; element index e lives at byte ((e & ~31) << 2) + ((e & 31) << 1) (+64 for lo).
struct sp32_t { uint32_t raw; };
__device__ __forceinline__ long sp32_byte(long e) { return ((e & ~31L) << 2) + ((e & 31L) << 1); }
template <> __device__ __forceinline__ float ldf<sp32_t>(const sp32_t* p, long i) {
    const char* b = reinterpret_cast<const char*>(p) + sp32_byte(i);
    return sp2f(*reinterpret_cast<const uint16_t*>(b)) + sp2f(*reinterpret_cast<const uint16_t*>(b + 64));
}
template <> __device__ __forceinline__ void stf<sp32_t>(sp32_t* p, long i, float v, unsigned* ovf) {
    char* b = reinterpret_cast<char*>(p) + sp32_byte(i);
    v = sp_value(v);  // one f32 number for both halves (split_dev.h)
    sp_count_now(ovf, __builtin_fabsf(v));  // range contract: a finite |v| >= 65520 is counted (these kernels are HBM-bound)
    const uint16_t h = f2sp(v);
    *reinterpret_cast<uint16_t*>(b) = h;
    *reinterpret_cast<uint16_t*>(b + 64) = f2sp(v - sp2f(h));
}
template <> __device__ __forceinline__ void ld4<sp32_t>(const sp32_t* p, long i, float* v) {
    const char* b = reinterpret_cast<const char*>(p) + sp32_byte(i);
    const uint2 h = *reinterpret_cast<const uint2*>(b);
    const uint2 l = *reinterpret_cast<const uint2*>(b + 64);
    v[0] = sp2f((uint16_t)(h.x & 0xffff)) + sp2f((uint16_t)(l.x & 0xffff));
    v[1] = sp2f((uint16_t)(h.x >> 16)) + sp2f((uint16_t)(l.x >> 16));
    v[2] = sp2f((uint16_t)(h.y & 0xffff)) + sp2f((uint16_t)(l.y & 0xffff));
    v[3] = sp2f((uint16_t)(h.y >> 16)) + sp2f((uint16_t)(l.y >> 16));
}
template <> __device__ __forceinline__ void st4<sp32_t>(sp32_t* p, long i, const float* vin, unsigned* ovf) {
    char* b = reinterpret_cast<char*>(p) + sp32_byte(i);
    uint16_t h[4];
    uint2 hh, ll;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sp_value(vin[j]);  // one f32 number for both halves (split_dev.h)
    {
        float amax = 0.f;  // range contract (split_dev.h sp_commit): these kernels are HBM-bound, the test rides along
        amax = sp_max2(sp_max2(amax, v[0], v[1]), v[2], v[3]);
        sp_count_now(ovf, amax);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = f2sp(v[j]);
    hh.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    hh.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    ll.x = (uint32_t)f2sp(v[0] - sp2f(h[0])) | ((uint32_t)f2sp(v[1] - sp2f(h[1])) << 16);
    ll.y = (uint32_t)f2sp(v[2] - sp2f(h[2])) | ((uint32_t)f2sp(v[3] - sp2f(h[3])) << 16);
    *reinterpret_cast<uint2*>(b) = hh;
    *reinterpret_cast<uint2*>(b + 64) = ll;
}

// 8 consecutive elements (i a multiple of 8).  The sp32 forms move ONE 16-byte piece per half: 8-byte accesses run at
// 0.54-0.70 x the 16-byte rate on this part (MI355X_MICROARCH.md), and conv0 / LayerNorm / the average pool were written
// with the 4-element helpers above (conv0 at 2 x its write floor, the pool at 2 x its read floor: round-4 review, item 7b).
template <typename T> __device__ __forceinline__ void ld8(const T* p, long i, float* v) {
    ld4<T>(p, i, v);
    ld4<T>(p, i + 4, v + 4);
}
template <> __device__ __forceinline__ void ld8<sp32_t>(const sp32_t* p, long i, float* v) {
    const char* b = reinterpret_cast<const char*>(p) + sp32_byte(i);
    const uint4 h = *reinterpret_cast<const uint4*>(b);
    const uint4 l = *reinterpret_cast<const uint4*>(b + 64);
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = sp2f((uint16_t)(hw[j] & 0xffff)) + sp2f((uint16_t)(lw[j] & 0xffff));
        v[2 * j + 1] = sp2f((uint16_t)(hw[j] >> 16)) + sp2f((uint16_t)(lw[j] >> 16));
    }
}
template <typename T> __device__ __forceinline__ void st8(T* p, long i, const float* v, unsigned* ovf = nullptr) {
    st4<T>(p, i, v, ovf);
    st4<T>(p, i + 4, v + 4, ovf);
}
template <> __device__ __forceinline__ void st8<sp32_t>(sp32_t* p, long i, const float* vin, unsigned* ovf) {
    char* b = reinterpret_cast<char*>(p) + sp32_byte(i);
    float v[8], amax = 0.f;
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = sp_value(vin[j]);  // one f32 number for both halves (split_dev.h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        amax = sp_max2(amax, v[2 * j], v[2 * j + 1]);
        const uint16_t h0 = f2sp(v[2 * j]), h1 = f2sp(v[2 * j + 1]);
        hw[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        lw[j] = (uint32_t)f2sp(v[2 * j] - sp2f(h0)) | ((uint32_t)f2sp(v[2 * j + 1] - sp2f(h1)) << 16);
    }
    sp_count_now(ovf, amax);  // range contract (split_dev.h)
    *reinterpret_cast<uint4*>(b) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *reinterpret_cast<uint4*>(b + 64) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

// ------------------------------------------------------------------------------------------------ preprocess
// data/utils.py:19-39.  u8 [n,in_h,in_w,3] RGB -> zero-bordered [n,230,230,4] (BGR - mean, 4th channel 0).
// The border materialises Conv2dSame's asymmetric padding (video.py:68-80: 2 before, 3 after) plus one extra
// row/column so that the stem runs as an un-padded 8x8-tap convolution with 32 contiguous values per tap row.
constexpr int PP = 230;
template <typename T>
__global__ void preprocess_kernel(const uint8_t* __restrict__ in, T* __restrict__ out, int n, int in_h, int in_w) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * PP * PP;
    if (idx >= total) return;
    const int x = idx % PP;
    const int y = (idx / PP) % PP;
    const int b = idx / ((long)PP * PP);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y >= 2 && y < 226 && x >= 2 && x < 226) {
        int sy = y - 2, sx = x - 2;
        if (in_h != 224 || in_w != 224) {  // PIL NEAREST: src = floor((dst + 0.5) * in / out)
            sy = min((int)(((double)sy + 0.5) * ((double)in_h / 224.0)), in_h - 1);
            sx = min((int)(((double)sx + 0.5) * ((double)in_w / 224.0)), in_w - 1);
        }
        const uint8_t* px = in + (((long)b * in_h + sy) * in_w + sx) * 3;
        v[0] = (float)px[2] - 91.4953f;
        v[1] = (float)px[1] - 103.8827f;
        v[2] = (float)px[0] - 131.0912f;
    }
    st4<T>(out, idx * 4, v);
}

// Planar split-fp16 variant (input of stem_pool_kernel): the same zero-bordered image as two fp16 planes
// [n,230,230,4], hi = bf16(v) and lo = bf16(v - hi), so that one 8-pixel tap row is 64 contiguous bytes per plane.
__device__ __forceinline__ void st4_planar(bf16_t* hi, bf16_t* lo, long idx, const float* vin, unsigned* ovf = nullptr) {
    uint16_t h[4];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sp_value(vin[j]);  // one f32 number for both halves (split_dev.h)
    {
        float amax = 0.f;  // range contract (split_dev.h sp_commit)
        amax = sp_max2(sp_max2(amax, v[0], v[1]), v[2], v[3]);
        sp_count_now(ovf, amax);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = f2sp(v[j]);
    uint2 hh, ll;
    hh.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    hh.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    ll.x = (uint32_t)f2sp(v[0] - sp2f(h[0])) | ((uint32_t)f2sp(v[1] - sp2f(h[1])) << 16);
    ll.y = (uint32_t)f2sp(v[2] - sp2f(h[2])) | ((uint32_t)f2sp(v[3] - sp2f(h[3])) << 16);
    *reinterpret_cast<uint2*>(hi + idx * 4) = hh;
    *reinterpret_cast<uint2*>(lo + idx * 4) = ll;
}

__global__ void preprocess_planar_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int n,
                                         int in_h, int in_w) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * PP * PP;
    if (idx >= total) return;
    const int x = idx % PP;
    const int y = (idx / PP) % PP;
    const int b = idx / ((long)PP * PP);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y >= 2 && y < 226 && x >= 2 && x < 226) {
        int sy = y - 2, sx = x - 2;
        if (in_h != 224 || in_w != 224) {  // PIL NEAREST: src = floor((dst + 0.5) * in / out)
            sy = min((int)(((double)sy + 0.5) * ((double)in_h / 224.0)), in_h - 1);
            sx = min((int)(((double)sx + 0.5) * ((double)in_w / 224.0)), in_w - 1);
        }
        const uint8_t* px = in + (((long)b * in_h + sy) * in_w + sx) * 3;
        v[0] = (float)px[2] - 91.4953f;
        v[1] = (float)px[1] - 103.8827f;
        v[2] = (float)px[0] - 131.0912f;
    }
    st4_planar(hi, lo, idx, v);
}

__global__ void pack_nchw_planar_kernel(const float* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int n,
                                        unsigned* ovf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * PP * PP;
    if (idx >= total) return;
    const int x = idx % PP;
    const int y = (idx / PP) % PP;
    const int b = idx / ((long)PP * PP);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y >= 2 && y < 226 && x >= 2 && x < 226) {
        const long o = (long)b * 3 * 224 * 224 + (long)(y - 2) * 224 + (x - 2);
        v[0] = in[o]; v[1] = in[o + 224 * 224]; v[2] = in[o + 2 * 224 * 224];
    }
    st4_planar(hi, lo, idx, v, ovf);  // arbitrary floats: the one input that can break the fp16 range by itself
}

// Same zero-bordered image from an ALREADY preprocessed float tensor [n,3,224,224] (the tensor the reference's
// pth_model_static is called with, get_prob_video.py:103-109).
template <typename T>
__global__ void pack_nchw_kernel(const float* __restrict__ in, T* __restrict__ out, int n) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * PP * PP;
    if (idx >= total) return;
    const int x = idx % PP;
    const int y = (idx / PP) % PP;
    const int b = idx / ((long)PP * PP);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y >= 2 && y < 226 && x >= 2 && x < 226) {
        const long o = (long)b * 3 * 224 * 224 + (long)(y - 2) * 224 + (x - 2);
        v[0] = in[o]; v[1] = in[o + 224 * 224]; v[2] = in[o + 2 * 224 * 224];
    }
    st4<T>(out, idx * 4, v);
}

// ------------------------------------------------------------------------------------------------ face stage (row f4)
// retina_face_predictor.py:70-82 with box_utils.py:210-249: every prior's box, score and five landmarks in pixels.
// The arithmetic follows torch's evaluation order in f32 with contraction off, so only expf may differ (<= 2 ulp).
__global__ void face_decode_kernel(const float* __restrict__ loc, const float* __restrict__ conf,
                                   const float* __restrict__ landms, const float* __restrict__ priors, int P, float im_w,
                                   float im_h, float var0, float var1, float* __restrict__ dets) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    {   // grid.y = frame of a batch: every frame has its own P rows of loc / conf / landms / dets, the priors are shared
        const long f = blockIdx.y;
        loc += f * P * 4; conf += f * P * 2; landms += f * P * 10; dets += f * P * 15;
    }
    const float4 pr = *reinterpret_cast<const float4*>(priors + 4L * i);
    const float4 l = *reinterpret_cast<const float4*>(loc + 4L * i);
    const float cx = pr.x + (l.x * var0) * pr.z;
    const float cy = pr.y + (l.y * var0) * pr.w;
    const float w = pr.z * expf(l.z * var1);
    const float h = pr.w * expf(l.w * var1);
    const float x0 = cx - w / 2.0f, y0 = cy - h / 2.0f;
    float* o = dets + 15L * i;
    o[0] = x0 * im_w;
    o[1] = y0 * im_h;
    o[2] = (w + x0) * im_w;
    o[3] = (h + y0) * im_h;
    o[4] = conf[2L * i + 1];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        o[5 + 2 * k] = (pr.x + (landms[10L * i + 2 * k] * var0) * pr.z) * im_w;
        o[6 + 2 * k] = (pr.y + (landms[10L * i + 2 * k + 1] * var0) * pr.w) * im_h;
    }
}

// retina_face_predictor.py:86-108 + py_cpu_nms.py:11-39 on the GPU: confidence floor, descending-score order, greedy
// NMS with the "+1 pixel" areas, top-k, final threshold.  Two kernels per batch of frames:
//   face_rank_kernel   position of every candidate (score > conf_thresh) in the descending order of py_cpu_nms
//                      (`scores.argsort()[: -top_k - 1 : -1]`: an ascending sort read backwards, so wherever that sort
//                      keeps tied scores in index order -- numpy's default does on short arrays, `kind="stable"` always
//                      -- ties are visited HIGHER prior index first; that is the rule here) by counting, written as
//                      order[rank] = prior index
//   face_nms_kernel    one workgroup per frame: the top nms_top_k boxes sit in LDS, boxes are visited in order, a kept
//                      box clears every later box whose IoU with it exceeds the threshold (one barrier per KEPT box)
// f32 arithmetic in numpy's evaluation order, contraction off, so the keep decisions are the reference's.
__global__ void face_rank_kernel(const float* __restrict__ dets, int P, float conf_thresh, int nms_top_k,
                                 int32_t* __restrict__ order, int32_t* __restrict__ count) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* d = dets + (long)f * P * 15;
    __shared__ float sc[256];
    const float si = i < P ? d[15L * i + 4] : 0.f;
    const bool cand = i < P && si > conf_thresh;
    int rank = 0;
    for (int j0 = 0; j0 < P; j0 += 256) {
        __syncthreads();
        const int jj = j0 + threadIdx.x;
        sc[threadIdx.x] = jj < P ? d[15L * jj + 4] : -1.f;
        __syncthreads();
        if (cand) {
            const int lim = min(256, P - j0);
            for (int k = 0; k < lim; ++k) {
                const float sj = sc[k];
                rank += (sj > conf_thresh) && (sj > si || (sj == si && j0 + k > i));
            }
        }
    }
    if (cand) {
        atomicAdd(count + f, 1);
        if (rank < nms_top_k) order[(long)f * nms_top_k + rank] = i;
    }
}

// The same order by SORTING (round 6): one workgroup per frame, the (score, prior index) keys of all P priors in LDS, a bitonic
// sort of the next power of two.  The counting kernel above does P comparisons per candidate -- 90 M per 640 x 360 frame when
// every prior is a candidate (synthetic detector weights; 14.7 ms per 750 frames, profiles/r06_face_kernel_stats_before.csv) --
// the sort 105 compare-exchange passes over 16 K keys.  Key = the score's bits made monotone (sign flip), prior index + 1 in the
// low word: a descending sort visits equal scores HIGHER index first, the rule above; non-candidates are key 0 and sink.
constexpr int SORT_THREADS = 1024;
constexpr int SORT_MAX = 16384;  // keys held in LDS (128 KiB); frames with more priors keep the counting kernel

// NPT compare-exchanges per thread and pass (N = 2048 NPT keys): fully unrolled, branch-free, so that a pass is NPT pairs of
// LDS reads in flight at once instead of a chain of dependent round trips
template <int NPT>
__device__ __forceinline__ void bitonic_desc(unsigned long long* key, int tid) {
    constexpr int N = 2 * SORT_THREADS * NPT;
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long ka[NPT], kb[NPT];
            int ia[NPT];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int t = tid + u * SORT_THREADS;
                ia[u] = 2 * t - (t & (j - 1));
                ka[u] = key[ia[u]];
                kb[u] = key[ia[u] + j];
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                // descending runs where bit k of the position is clear: the whole array at k = N
                const bool sw = ((ia[u] & k) == 0) ? ka[u] < kb[u] : ka[u] > kb[u];
                key[ia[u]] = sw ? kb[u] : ka[u];
                key[ia[u] + j] = sw ? ka[u] : kb[u];
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(SORT_THREADS) face_sort_kernel(const float* __restrict__ dets, int P, float conf_thresh,
                                                                  int nms_top_k, int32_t* __restrict__ order, int32_t* __restrict__ count) {
    extern __shared__ __align__(16) unsigned long long sort_keys[];
    unsigned long long* key = sort_keys;
    __shared__ int cnt;
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* d = dets + (long)f * P * 15;
    if (tid == 0) cnt = 0;
    __syncthreads();
    // candidates only, compacted in arrival order (the sort fixes the order): a real detector leaves a few dozen of 9520 priors
    // above the confidence floor, and the sort below then runs on 2048 keys instead of 16384
    for (int i0 = 0; i0 < P; i0 += 4 * SORT_THREADS) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SORT_THREADS + tid;
            sc[u] = i < P ? d[15L * i + 4] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SORT_THREADS + tid;
            if (i < P && sc[u] > conf_thresh) {
                unsigned v = __float_as_uint(sc[u]);
                v = (v & 0x80000000u) ? ~v : (v | 0x80000000u);
                key[atomicAdd(&cnt, 1)] = ((unsigned long long)v << 32) | (unsigned)(i + 1);
            }
        }
    }
    __syncthreads();
    const int c = cnt;
    int N = 2 * SORT_THREADS;
    while (N < c) N <<= 1;
    for (int i = c + tid; i < N; i += SORT_THREADS) key[i] = 0ull;  // padding sinks to the end of a descending sort
    __syncthreads();
    switch (N / (2 * SORT_THREADS)) {
        case 1: bitonic_desc<1>(key, tid); break;
        case 2: bitonic_desc<2>(key, tid); break;
        case 4: bitonic_desc<4>(key, tid); break;
        default: bitonic_desc<8>(key, tid); break;
    }
    const int n = min(c, nms_top_k);
    for (int r = tid; r < n; r += SORT_THREADS) order[(long)f * nms_top_k + r] = (int)(unsigned)(key[r] & 0xffffffffu) - 1;
    if (tid == 0) count[f] = c;
}

constexpr int NMS_THREADS = 1024;
constexpr int NMS_MAX = 6144;  // API bound on nms_top_k (the kernel itself no longer holds all boxes at once)

// Greedy NMS of one frame per workgroup, in CHUNKS of 1024 boxes of the descending-score order (round 6).  The reference keeps
// every survivor and then takes keep[:top_k] (retina_face_predictor.py:96-100); boxes are visited in score order, so the first top_k
// kept ones ARE that prefix and nothing behind the chunk that completes them is ever looked at.  Per chunk: every thread owns one
// box; it is first tested against the boxes kept from earlier chunks (their corners sit in LDS), then the chunk is walked in
// order with one barrier per kept box and at most ONE IoU per thread and kept box.  The first form of this kernel held all
// nms_top_k boxes in 126 KiB of LDS (one block per CU) and tested every later box against every kept one: with every prior a
// candidate (synthetic detector weights) 4.4-6.7 ms per 750 frames, here the walk ends inside the first chunk and three blocks
// share a CU.  f32 arithmetic in numpy's evaluation order, contraction off: the keep decisions are the reference's.
__global__ void __launch_bounds__(NMS_THREADS) face_nms_kernel(const float* __restrict__ dets, int P, const int32_t* __restrict__ order,
                                                                const int32_t* __restrict__ count, int nms_top_k, float nms_thresh,
                                                                int top_k, float threshold, float* __restrict__ out,
                                                                int32_t* __restrict__ out_n) {
#pragma clang fp contract(off)
    __shared__ float cx1[NMS_THREADS], cy1[NMS_THREADS], cx2[NMS_THREADS], cy2[NMS_THREADS], car[NMS_THREADS];  // the chunk
    __shared__ float kx1[1024], ky1[1024], kx2[1024], ky2[1024], kar[1024];                                       // kept so far
    __shared__ unsigned char cdead[NMS_THREADS];
    __shared__ int kept[1024];  // position in the score order of every kept box (beyond top_k they are never reported)
    __shared__ int out_rows;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(count[f], nms_top_k);
    const float* d = dets + (long)f * P * 15;
    const int32_t* ord = order + (long)f * nms_top_k;
    const int cap = min(top_k, 1024);
    int nkept = 0;  // uniform across the block
    for (int c0 = 0; c0 < n && nkept < cap; c0 += NMS_THREADS) {
        const int cn = min(NMS_THREADS, n - c0);
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, ar = 0.f;
        bool dead = tid >= cn;
        if (!dead) {
            const float* r = d + 15L * ord[c0 + tid];
            x1 = r[0]; y1 = r[1]; x2 = r[2]; y2 = r[3];
            ar = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);
            for (int k = 0; k < nkept; ++k) {  // boxes kept from earlier chunks all precede this one in the order
                const float w = fmaxf(0.0f, fminf(kx2[k], x2) - fmaxf(kx1[k], x1) + 1.0f);
                const float h = fmaxf(0.0f, fminf(ky2[k], y2) - fmaxf(ky1[k], y1) + 1.0f);
                const float inter = w * h;
                const float ovr = inter / (kar[k] + ar - inter);
                if (!(ovr <= nms_thresh)) { dead = true; break; }
            }
        }
        __syncthreads();  // the previous chunk's arrays are free (every thread has left its walk)
        cx1[tid] = x1; cy1[tid] = y1; cx2[tid] = x2; cy2[tid] = y2; car[tid] = ar;
        cdead[tid] = dead ? 1 : 0;
        __syncthreads();
        for (int a = 0; a < cn; ++a) {
            if (cdead[a]) continue;  // uniform: every thread reads the same flag behind the previous barrier
            const float ax1 = cx1[a], ay1 = cy1[a], ax2 = cx2[a], ay2 = cy2[a], aar = car[a];
            if (tid == 0) {
                kept[nkept] = c0 + a;
                kx1[nkept] = ax1; ky1[nkept] = ay1; kx2[nkept] = ax2; ky2[nkept] = ay2; kar[nkept] = aar;
            }
            if (++nkept >= cap) break;
            if (tid > a && !dead) {
                const float w = fmaxf(0.0f, fminf(ax2, x2) - fmaxf(ax1, x1) + 1.0f);
                const float h = fmaxf(0.0f, fminf(ay2, y2) - fmaxf(ay1, y1) + 1.0f);
                const float inter = w * h;
                const float ovr = inter / (aar + ar - inter);
                if (!(ovr <= nms_thresh)) { dead = true; cdead[tid] = 1; }
            }
            __syncthreads();
        }
        __syncthreads();  // tid 0's last kept entry is visible before the next chunk tests against it
    }
    // dets[keep][:top_k], then the rows with score >= threshold (retina_face_predictor.py:96-108)
    __syncthreads();
    const int nk = min(nkept, cap);
    if (tid == 0) {
        int m = 0;
        for (int k = 0; k < nk; ++k) {
            const int idx = ord[kept[k]];
            if (d[15L * idx + 4] >= threshold) kept[m++] = idx;  // compaction in place (m <= k)
        }
        out_rows = m;
        out_n[f] = m;
    }
    __syncthreads();
    for (int e = tid; e < out_rows * 15; e += NMS_THREADS) out[((long)f * top_k + e / 15) * 15 + e % 15] = d[15L * kept[e / 15] + e % 15];
}

// get_face_images.py:52-56 (crop of the decoded frame) + data/utils.py:34 (PIL NEAREST resize to 224x224) in one pass:
// tile pixel (y, x) = frame[f][y0 + floor((y + .5) * ch / 224)][x0 + floor((x + .5) * cw / 224)], channels swapped
// when the frames are BGR (cv2) so that the tile is RGB like the image PIL reads back.  One thread per 4 tile pixels.
__global__ void crop_tiles_kernel(const uint8_t* __restrict__ frames, int T, int H, int W, const int32_t* __restrict__ rects,
                                  int n, int swap_rb, uint8_t* __restrict__ tiles) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * 224 * 56;
    if (idx >= total) return;
    const int xq = idx % 56;
    const int y = (idx / 56) % 224;
    const int t = idx / (56L * 224);
    const int32_t* r = rects + 5L * t;
    const int f = r[0], x0 = r[1], y0 = r[2], cw = r[3] - r[1], ch = r[4] - r[2];
    uint8_t px[12];
    const bool ok = f >= 0 && f < T && x0 >= 0 && y0 >= 0 && cw > 0 && ch > 0 && x0 + cw <= W && y0 + ch <= H;
    if (ok) {
        const int sy = y0 + min((int)(((double)y + 0.5) * ((double)ch / 224.0)), ch - 1);
        const uint8_t* row = frames + ((long)f * H + sy) * W * 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = xq * 4 + j;
            const int sx = x0 + min((int)(((double)x + 0.5) * ((double)cw / 224.0)), cw - 1);
            const uint8_t* s = row + 3L * sx;
            px[3 * j + 0] = swap_rb ? s[2] : s[0];
            px[3 * j + 1] = s[1];
            px[3 * j + 2] = swap_rb ? s[0] : s[2];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) px[j] = 0;
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(tiles + ((long)t * 224 + y) * 224 * 3 + xq * 12);  // 12-byte aligned to 4
    o[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24);
    o[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24);
    o[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((uint32_t)px[11] << 24);
}

// ------------------------------------------------------------------------------------------------ RetinaFace network (row f4)
// retina_face_predictor.py:59-65: BGR pixels minus (104, 117, 123) (an RGB frame is flipped first), written as a
// zero-bordered NHWC4 image [n, ph, pw, 4] with the picture at offset (3, 3): the border is torchvision's conv1 padding
// of 3 plus the extra rows/columns that let the 7x7/2 stem run as an un-padded 8x(8 pixels x 4 channels) contraction.
template <typename T>
__global__ void face_pre_kernel(const uint8_t* __restrict__ in, T* __restrict__ out, int n, int h, int w, int ph, int pw,
                                int rgb) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * ph * pw;
    if (idx >= total) return;
    const int x = idx % pw;
    const int y = (idx / pw) % ph;
    const int b = idx / ((long)ph * pw);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int sy = y - 3, sx = x - 3;
    if (sy >= 0 && sy < h && sx >= 0 && sx < w) {
        const uint8_t* px = in + (((long)b * h + sy) * w + sx) * 3;
        v[0] = (float)((int)px[rgb ? 2 : 0] - 104);
        v[1] = (float)((int)px[1] - 117);
        v[2] = (float)((int)px[rgb ? 0 : 2] - 123);
    }
    st4<T>(out, idx * 4, v);
}

// torchvision ResNet max-pool: 3x3, stride 2, padding 1 (padded taps never win); NHWC.
template <typename T>
__global__ void maxpool3s2p1_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int oh, int ow) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = c / 4;
    const long total = (long)n * oh * ow * c4;
    if (idx >= total) return;
    const int cc = (idx % c4) * 4;
    long t = idx / c4;
    const int ox = t % ow; t /= ow;
    const int oy = t % oh;
    const int b = t / oh;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bool nan[4] = {false, false, false, false};
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= h) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if (ix < 0 || ix >= w) continue;
            float v[4];
            ld4<T>(x, (((long)b * h + iy) * w + ix) * c + cc, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) { m[j] = fmaxf(m[j], v[j]); nan[j] |= v[j] != v[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (nan[j]) m[j] = NAN;
    st4<T>(y, (((long)b * oh + oy) * ow + ox) * c + cc, m);
}

// retina_face_net.py:92-98: y += nearest-upsampled coarser level (F.interpolate(mode="nearest"): src = floor(dst*in/out))
template <typename T>
__global__ void upsample_add_kernel(T* __restrict__ y, const T* __restrict__ coarse, int n, int h, int w, int ch, int cw, int c,
                                    unsigned* ovf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = c / 4;
    const long total = (long)n * h * w * c4;
    if (idx >= total) return;
    const int cc = (idx % c4) * 4;
    long t = idx / c4;
    const int x = t % w; t /= w;
    const int yy = t % h;
    const int b = t / h;
    const int sy = min((int)floorf((float)yy * ((float)ch / (float)h)), ch - 1);
    const int sx = min((int)floorf((float)x * ((float)cw / (float)w)), cw - 1);
    float a[4], u[4];
    const long o = (((long)b * h + yy) * w + x) * c + cc;
    ld4<T>(y, o, a);
    ld4<T>(coarse, (((long)b * ch + sy) * cw + sx) * c + cc, u);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += u[j];
    st4<T>(y, o, a, ovf);
}

// retina_face.py:9-43,104-113: per position the fused head GEMM produced 32 values = class [2 anchors x 2], bbox
// [2 x 4], landmarks [2 x 10] (row stride ld); scatter them anchor-major into loc / conf (softmax over the 2 classes,
// F.softmax(dim=-1)) / landms at `row0` of an image with P rows in all.
__global__ void face_head_kernel(const float* __restrict__ hd, int ld, int n, int hw, int row0, int P, float* __restrict__ loc,
                                 float* __restrict__ conf, float* __restrict__ landms) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * hw * 2) return;
    const int a = idx & 1;
    const long pos = idx >> 1;            // b * hw + cell
    const int b = pos / hw;
    const long cell = pos - (long)b * hw;
    const float* v = hd + pos * ld;
    const long r = (long)b * P + row0 + cell * 2 + a;
    const float c0 = v[2 * a], c1 = v[2 * a + 1];
    const float mx = fmaxf(c0, c1);
    const float e0 = expf(c0 - mx), e1 = expf(c1 - mx);
    conf[2 * r] = e0 / (e0 + e1);
    conf[2 * r + 1] = e1 / (e0 + e1);
#pragma unroll
    for (int j = 0; j < 4; ++j) loc[4 * r + j] = v[4 + 4 * a + j];
#pragma unroll
    for (int j = 0; j < 10; ++j) landms[10 * r + j] = v[12 + 10 * a + j];
}

// get_prob_video.py:115-123: window rows = relu(features) gathered by index into [nwin, 10, 512]
__global__ void gather_windows_kernel(const float* __restrict__ feats, const int32_t* __restrict__ idx, float* __restrict__ out,
                                      long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long row = i / 128;  // 512 / 4 vectors per feature row
    const int c = (i % 128) * 4;
    float v[4];
    ld4<float>(feats, (long)idx[row] * 512 + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = relu_nan(v[j]);
    st4<float>(out, row * 512 + c, v);
}

// get_prob_audio_8_cl.py:79-86 + data/utils.py:63-89: chunk = wav[start:end] padded to `window` samples with the
// chunk mean (mode 0; NaN for an empty chunk, like torch.mean of an empty tensor), zeros (1) or by tiling (2).
__global__ void audio_chunks_kernel(const float* __restrict__ wav, const int32_t* __restrict__ starts,
                                    const int32_t* __restrict__ ends, int window, int mode, float* __restrict__ out) {
    __shared__ float red[8];
    __shared__ float bc;
    const int c = blockIdx.x;
    const int s = starts[c], len = ends[c] - starts[c];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    float fill = 0.f;
    if (mode == 0) {
        float a = 0.f;
        for (int i = tid; i < len; i += blockDim.x) a += wav[s + i];
        a = wave_sum(a);
        if (lane == 0) red[wv] = a;
        __syncthreads();
        if (tid == 0) { float t = 0.f; for (int i = 0; i < nw; ++i) t += red[i]; bc = t / (float)len; }
        __syncthreads();
        fill = bc;
    }
    float* o = out + (long)c * window;
    for (int i = tid; i < window; i += blockDim.x) {
        float v;
        if (i < len) v = wav[s + i];
        else if (mode == 2) v = len > 0 ? wav[s + i % len] : NAN;  // empty chunk: the reference raises (i % 0); never index
        else v = fill;
        o[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ pooling
// video.py:103,117: MaxPool2d(3, stride 2), no padding; NHWC.
template <typename T>
__global__ void maxpool3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int oh, int ow) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = c / 4;
    const long total = (long)n * oh * ow * c4;
    if (idx >= total) return;
    const int cc = (idx % c4) * 4;
    long t = idx / c4;
    const int ox = t % ow; t /= ow;
    const int oy = t % oh;
    const int b = t / oh;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bool nan[4] = {false, false, false, false};
    for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
            float v[4];
            ld4<T>(x, (((long)b * h + oy * 2 + dy) * w + ox * 2 + dx) * c + cc, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) { m[j] = fmaxf(m[j], v[j]); nan[j] |= v[j] != v[j]; }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (nan[j]) m[j] = NAN;
    st4<T>(y, (((long)b * oh + oy) * ow + ox) * c + cc, m);
}

// video.py:110,124: AdaptiveAvgPool2d((1,1)) over hw positions of an NHWC tensor -> f32 [n,c]
// One thread per (frame, 8 consecutive channels): 16-byte loads (the one-channel form read an sp32 tensor two bytes at a
// time: 3.4 TB/s); every channel still adds its positions in order 0 .. hw-1, so the sums are the same bits.
// Positions are requested seven at a time before any is added (one frame per call is a single block: its 49 dependent
// round trips were 19 us), and added in the same order.  ysp: the same values once more as sp32 pairs (fc1's operand in x3 mode).
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, float* __restrict__ y, sp32_t* __restrict__ ysp, int n, int hw, int c,
                               unsigned* ovf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = c / 8;
    if (idx >= (long)n * c8) return;
    const int cc = (idx % c8) * 8;
    const long b = idx / c8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int U = 7;
    for (int i0 = 0; i0 < hw; i0 += U) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) ld8<T>(x, (b * hw + min(i0 + u, hw - 1)) * c + cc, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u < hw) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[u][j];
            }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = s[j] / (float)hw;
    st8<float>(y, b * c + cc, s);
    if (ysp) st8<sp32_t>(ysp, b * c + cc, s, ovf);
}

// ------------------------------------------------------------------------------------------------ tiny heads
// out[m, j] = sum_k f(x[m,k]) * w[j,k] + b[j], n <= 16 outputs; optional ReLU on the input, optional softmax.
// video.py:131-132 (relu1 -> fc2) + get_prob_video.py:107-109 (softmax dim=1); LSTM fc (video.py:184);
// audio feature_downsample (audio_8_cl.py:189).  One wave per row.
__global__ void small_linear_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                    float* __restrict__ logits, float* __restrict__ probs, int m, int k, int n,
                                    int relu_in) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m) return;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    // (unrolled: the loads of eight K positions are in flight together -- one row per call is one wave, and 8 dependent
    // round trips per output were the kernel; the adds keep their order)
    // and no load sits behind a branch of its own: `if (j < n) acc[j] += xv * w[..]` made n dependent round trips per K step
    // (3 us each: 59 us for the audio head's 1024 x 8); rows past n re-read row n - 1 into accumulators nobody uses)
#pragma unroll 4
    for (int kk = lane; kk < k; kk += 64) {
        float xv = x[(long)row * k + kk];
        if (relu_in) xv = relu_nan(xv);
        float wv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) wv[j] = w[(long)min(j, n - 1) * k + kk];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += xv * wv[j];
    }
    float mx = -INFINITY;
    float bj[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bj[j] = b[min(j, n - 1)];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < n) {
            acc[j] = wave_sum(acc[j]) + bj[j];
            mx = fmaxf(mx, acc[j]);
        }
    if (lane == 0) {
        float den = 0.f;
        float e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < n) {
                if (logits) logits[(long)row * n + j] = acc[j];
                e[j] = expf(acc[j] - mx);
                den += e[j];
            }
        if (probs)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < n) probs[(long)row * n + j] = e[j] / den;
    }
}

// ------------------------------------------------------------------------------------------------ LSTM cell
// torch.nn.LSTM cell (video.py:173-178), gate order i,f,g,o.  xproj already holds x W_ih^T + b_ih + b_hh.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// h_sp: h once more as sp32 pairs (same row stride), the operand of the next step's recurrent contraction in the x3 mode
__global__ void lstm_cell_kernel(const float* __restrict__ xproj, long xproj_ld, const float* __restrict__ hproj,
                                 float* __restrict__ c, float* __restrict__ h_out, sp32_t* __restrict__ h_sp, long h_ld, int n,
                                 int hid, int first, unsigned* ovf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * hid) return;
    const int j = idx % hid;
    const int r = idx / hid;
    const float* xp = xproj + (long)r * xproj_ld;
    float gi = xp[j], gf = xp[hid + j], gg = xp[2 * hid + j], go = xp[3 * hid + j];
    float cp = 0.f;
    if (!first) {
        const float* hp = hproj + (long)r * 4 * hid;
        gi += hp[j]; gf += hp[hid + j]; gg += hp[2 * hid + j]; go += hp[3 * hid + j];
        cp = c[idx];
    }
    const float cn = sigmoidf_(gf) * cp + sigmoidf_(gi) * tanhf(gg);
    c[idx] = cn;
    const float hn = sigmoidf_(go) * tanhf(cn);
    h_out[(long)r * h_ld + j] = hn;
    if (h_sp) stf<sp32_t>(h_sp, (long)r * h_ld + j, hn, ovf);
}

// ------------------------------------------------------------------------------------------------ audio front end
// HF Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm (call site get_prob_audio_8_cl.py:88-89):
// (x - mean) / sqrt(var + 1e-7), population variance, per row.  One block per row.
__global__ void wav_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int t) {
    __shared__ float red[8];
    __shared__ float bc;
    const float* xr = x + (long)blockIdx.x * t;
    float* yr = y + (long)blockIdx.x * t;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    // (every pass requests eight of a thread's samples before it uses one -- a 4 s window is one block, and 125 dependent
    // round trips per pass were 90 us; each thread still adds its samples in the same order)
    constexpr int U = 8;
    const int bd = blockDim.x;
    float s = 0.f;
    for (int i = tid; i < t; i += U * bd) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xr[min(i + u * bd, t - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * bd < t) s += v[u];
    }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (tid == 0) { float a = 0.f; for (int i = 0; i < nw; ++i) a += red[i]; bc = a / (float)t; }
    __syncthreads();
    const float mean = bc;
    float q = 0.f;
    for (int i = tid; i < t; i += U * bd) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xr[min(i + u * bd, t - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * bd < t) { const float d = v[u] - mean; q += d * d; }
    }
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[wv] = q;
    __syncthreads();
    if (tid == 0) { float a = 0.f; for (int i = 0; i < nw; ++i) a += red[i]; bc = sqrtf(a / (float)t + 1e-7f); }
    __syncthreads();
    const float sd = bc;
    for (int i = tid; i < t; i += U * bd) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xr[min(i + u * bd, t - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * bd < t) yr[i + u * bd] = (v[u] - mean) / sd;
    }
}

// wav2vec2 feature-extractor layer 0: Conv1d(1 -> 512, k=10, stride 5, bias) -> LayerNorm(512) -> GELU, fused.
// Output [n, t_out, 512] time-major.  One wave per time step, each lane owns 8 consecutive channels whose
// 10-tap filters stay in registers for the whole block; the write (2 KiB contiguous per wave) is the cost.
template <typename T>
__global__ void __launch_bounds__(256) conv0_ln_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ g,
                                                         const float* __restrict__ beta, T* __restrict__ y, int t_in,
                                                         int t_out, int steps_per_block, unsigned* ovf) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = lane * 8;
    float wr[8][10], br[8], gr[8], ber[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 10; ++k) wr[j][k] = w[(c0 + j) * 10 + k];
        br[j] = b[c0 + j]; gr[j] = g[c0 + j]; ber[j] = beta[c0 + j];
    }
    const int row = blockIdx.y;
    const float* xr = x + (long)row * t_in;
    const int t_begin = blockIdx.x * steps_per_block;
    const int t_end = min(t_begin + steps_per_block, t_out);
    for (int t = t_begin + wv; t < t_end; t += 4) {
        float xv[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) xv[k] = xr[t * 5 + k];
        float v[8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) a += xv[k] * wr[j][k];
            v[j] = a + br[j];
            s += v[j];
        }
        const float mean = wave_sum(s) * (1.f / 512.f);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * (1.f / 512.f) + 1e-5f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float u = (v[j] - mean) * rstd * gr[j] + ber[j];
            if constexpr (std::is_same<T, float>::value) v[j] = gelu_erf(u);  // f32 mode: the library erff
            else v[j] = gelu_fast(u);
        }
        const long o = ((long)row * t_out + t) * 512 + c0;
        st8<T>(y, o, v, ovf);
    }
}

// LayerNorm over the last dimension (c in {512, 1024}), optional residual add in front, optional GELU behind,
// dual output (f32 residual-stream copy and/or bf16 GEMM-operand copy).  One wave per row.
template <typename TI, typename OB, int C>
__global__ void __launch_bounds__(256) layernorm_kernel(const TI* __restrict__ x, const TI* __restrict__ res,
                                                      const float* __restrict__ g, const float* __restrict__ b,
                                                      float* __restrict__ yf, OB* __restrict__ yb, long rows,
                                                      float eps, int act, unsigned* ovf) {
    constexpr int PER = C / 64;  // 8 or 16 consecutive elements per lane
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const long base = row * C + lane * PER;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; i += 8) ld8<TI>(x, base + i, v + i);
    if (res) {
        float r[PER];
#pragma unroll
        for (int i = 0; i < PER; i += 8) ld8<TI>(res, base + i, r + i);
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] += r[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / C) + eps);
#pragma unroll
    for (int i = 0; i < PER; i += 4) {
        const float4 gg = *reinterpret_cast<const float4*>(g + lane * PER + i);
        const float4 bb = *reinterpret_cast<const float4*>(b + lane * PER + i);
        v[i + 0] = (v[i + 0] - mean) * rstd * gg.x + bb.x;
        v[i + 1] = (v[i + 1] - mean) * rstd * gg.y + bb.y;
        v[i + 2] = (v[i + 2] - mean) * rstd * gg.z + bb.z;
        v[i + 3] = (v[i + 3] - mean) * rstd * gg.w + bb.w;
    }
    if (act == 2) {
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = gelu_erf(v[i]);
    } else if (act == 3) {
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = gelu_fast(v[i]);
    }
#pragma unroll
    for (int i = 0; i < PER; i += 8) {
        if (yf) st8<float>(yf, base + i, v + i);
        if (yb) st8<OB>(yb, base + i, v + i, ovf);
    }
}

// attention_layers.py:206-211,249-254: x + pe[:, :S]; writes f32 (residual) and/or bf16 (GEMM operand).
template <typename OB>
__global__ void add_pe_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ yf,
                              OB* __restrict__ yb, long total4, int s, int c, unsigned* ovf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const long e = idx * 4;
    const int cc = e % c;
    const int ss = (e / c) % s;
    float v[4], p[4];
    ld4<float>(x, e, v);
    ld4<float>(pe, (long)ss * c + cc, p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += p[j];
    if (yf) st4<float>(yf, e, v);
    if (yb) st4<OB>(yb, e, v, ovf);
}

// ------------------------------------------------------------------------------------------------ attention
// softmax(q k^T * scale) v for one (batch, head) per workgroup; S <= 256, d in {32, 64}.
// wav2vec2 encoder self-attention (16 x 64) and attention_layers.py:10-38 (32 x 32, 16 x 64).
// K (rows padded by 4 floats: conflict-free b128 row reads) and V live in LDS as f32; each wave owns query rows.
// 8 waves share one head's K/V image (~105 KiB f32 at S=199): two waves per SIMD hide the LDS latency of the score loop
constexpr int ATT_WAVES = 8;
constexpr int ATT_THREADS = ATT_WAVES * 64;
template <typename T, typename TO, int D>
__global__ void __launch_bounds__(ATT_THREADS) attention_kernel(const T* __restrict__ qkv, TO* __restrict__ out, int s, int heads,
                                                      float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KP = D + 4;
    float* ks = reinterpret_cast<float*>(smem_raw);           // [s][KP]
    float* vs = ks + (long)s * KP;                            // [s][D]
    float* ps = vs + (long)s * D;                             // [ATT_WAVES][256]
    float* qs = ps + ATT_WAVES * 256;                         // [ATT_WAVES][D]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int e = heads * D;
    const long rowstride = 3L * e;
    const T* base = qkv + (long)b * s * rowstride + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < s * (D / 4); i += ATT_THREADS) {
        const int r = i / (D / 4), c4 = (i % (D / 4)) * 4;
        float kv[4], vv[4];
        ld4<T>(base, (long)r * rowstride + e + c4, kv);
        ld4<T>(base, (long)r * rowstride + 2 * e + c4, vv);
        *reinterpret_cast<float4*>(ks + r * KP + c4) = make_float4(kv[0], kv[1], kv[2], kv[3]);
        *reinterpret_cast<float4*>(vs + r * D + c4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    __syncthreads();
    float* pw = ps + wv * 256;
    float* qw = qs + wv * D;
    for (int qi = wv; qi < s; qi += ATT_WAVES) {
        if (lane < D) qw[lane] = ldf<T>(base, (long)qi * rowstride + lane) * scale;
        __builtin_amdgcn_wave_barrier();
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            float a = -INFINITY;
            if (j < s) {
                a = 0.f;
#pragma unroll
                for (int c = 0; c < D; c += 4) {
                    const float4 kk = *reinterpret_cast<const float4*>(ks + j * KP + c);
                    const float4 qq = *reinterpret_cast<const float4*>(qw + c);
                    a += qq.x * kk.x + qq.y * kk.y + qq.z * kk.z + qq.w * kk.w;
                }
            }
            sc[jj] = a;
            mx = fmaxf(mx, a);
        }
        mx = wave_max(mx);
        float den = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            const float pv = j < s ? expf(sc[jj] - mx) : 0.f;
            pw[j] = pv;
            den += pv;
        }
        den = wave_sum(den);
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        if constexpr (D == 64) {
            for (int j = 0; j < s; ++j) o += pw[j] * vs[j * D + lane];
        } else {
            const int c = lane & 31, half = lane >> 5;
            for (int j = half; j < s; j += 2) o += pw[j] * vs[j * D + c];
            o += __shfl_xor(o, 32, 64);
        }
        if (lane < D) stf<TO>(out, ((long)b * s + qi) * e + h * D + lane, o / den);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- MFMA attention for 64-wide heads (wav2vec2 encoder layers, TransformerLayer 2) ---------------------------------
// One workgroup (8 waves) per (window, head).  K is kept in LDS as 16-bit rows (GEMM swizzle; fp16 in the x3 form, bf16 else), V transposed and key-
// permuted, both as hi (+ lo in the split mode) planes.  Each wave takes 16-query tiles:
//   S^T tile = K . Q^T   (swapped operands: a lane then holds, for ONE query lane&15, the keys 16t + 4(lane>>4) + r)
//   softmax over keys     in-lane over its registers + 2 shuffles across the four lane groups
//   O^T tile = V^T . P^T  the exponentiated accumulators of key tiles (2b, 2b+1) ARE the B operand of the PV MFMA for
//                         key block b once V^T is stored with k-index 8g+e <-> key 32b + 16(e>>2) + 4g + (e&3)
// X3 = 1: every product as hi.hi + hi.lo + lo.hi of fp16 pairs (f32-grade, as conv_gemm MODE 2/3; split_dev.h: scores and
// exponentials are O(1), nothing here needs a scale); X3 = 0: bf16 operands.
typedef __attribute__((ext_vector_type(8))) __bf16 att_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float att_f32x4_t;
// operand element by arithmetic: the split type (fp16) in the x3 form, bf16 in the plain form
template <int X3> struct AttOp {
    typedef spe_t elem_t;
    typedef spx8_t frag_t;
    static __device__ __forceinline__ uint16_t bits(float f) { return f2sp(f); }
    static __device__ __forceinline__ float val(uint16_t b) { return sp2f(b); }
    static __device__ __forceinline__ att_f32x4_t mfma(const frag_t a, const frag_t b, const att_f32x4_t c) { return mfma_sp(a, b, c); }
};
template <> struct AttOp<0> {
    typedef __bf16 elem_t;
    typedef att_bf16x8_t frag_t;
    static __device__ __forceinline__ uint16_t bits(float f) { return f2bf(f); }
    static __device__ __forceinline__ float val(uint16_t b) { return bf2f(b); }
    static __device__ __forceinline__ att_f32x4_t mfma(const frag_t a, const frag_t b, const att_f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ int att_swz(int row, int chunk) {
    return row * 128 + ((chunk ^ (int)((0x32765410u >> (((row >> 1) & 7) * 4)) & 7u)) << 4);
}

constexpr int ATTM_WAVES = 8;  // one query tile per wave at 99 tokens (7 tiles): the four-wave form ran two rounds of 2 / 2 / 2 / 1
template <typename T, typename TO, int NKT, int X3, int D>
__global__ void __launch_bounds__(64 * ATTM_WAVES) attention_mfma_kernel(const T* __restrict__ qkv, TO* __restrict__ out, int s, int heads,
                                                           float scale, unsigned* ovf) {
    static_assert(D == 64 || D == 32, "head dimension 64 (wav2vec2 layers, tl2) or 32 (tl1)");
    using Op = AttOp<X3>;
    using frag_t = typename Op::frag_t;
    using elem_t = typename Op::elem_t;
    constexpr int KS = D / 32;               // 32-wide K-steps of Q.K^T
    constexpr int TV = D / 16;               // 16-row tiles of V^T / O^T
    constexpr int SP = NKT * 16;             // padded key count
    constexpr int VROW = SP * 2 + 16;        // bytes per V^T row (16-byte pad against bank conflicts)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* khi = smem_raw;                    // [SP][128 B]
    char* klo = khi + SP * 128;
    char* vhi = klo + (X3 ? SP * 128 : 0);   // [D][VROW]
    char* vlo = vhi + D * VROW;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int e = heads * D;
    const long rowstride = 3L * e;
    const T* base = qkv + (long)b * s * rowstride + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, q16 = lane & 15;

    // Every global load of the block is issued before the first one is needed: a block is a chain of HBM round trips
    // otherwise (four staging passes, then one per query tile: ~6 x 2.5 us against ~4 us of arithmetic -- the launch ran at
    // a third of its present speed).  First the query rows of this wave's tiles (tile wv, wv + 8, ...), raw; then K / V in
    // batches of up to four staging passes.
    const int nqt = (s + 15) >> 4;
    constexpr int NTHR = 64 * ATTM_WAVES;
    constexpr int QI = NKT / ATTM_WAVES;     // query tiles per wave
    float qraw[QI][KS][8];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const int qrow = (wv + ATTM_WAVES * qi) * 16 + q16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (qrow < s) {
                ld4<T>(base, (long)qrow * rowstride + 32 * ks + 8 * g, qraw[qi][ks]);
                ld4<T>(base, (long)qrow * rowstride + 32 * ks + 8 * g + 4, qraw[qi][ks] + 4);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qraw[qi][ks][j] = 0.f;
            }
        }
    }
    // ---- stage K (row-major) and V (transposed + permuted) as 16-bit planes
    constexpr int ITEMS = SP * (D / 8), PASSES = ITEMS / NTHR, UB = PASSES < 4 ? PASSES : 4;
    static_assert(ITEMS % NTHR == 0 && PASSES % UB == 0 && NKT % ATTM_WAVES == 0, "whole staging passes, whole batches");
    float amax = 0.f;  // largest finite magnitude this thread split into an fp16 pair
    for (int p0 = 0; p0 < PASSES; p0 += UB) {
    float kvb[UB][8], vvb[UB][8];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int it = (p0 + u) * NTHR + tid;
        const int r = it / (D / 8), c = it % (D / 8);
        if (r < s) {
            ld4<T>(base, (long)r * rowstride + e + 8 * c, kvb[u]);
            ld4<T>(base, (long)r * rowstride + e + 8 * c + 4, kvb[u] + 4);
            ld4<T>(base, (long)r * rowstride + 2 * e + 8 * c, vvb[u]);
            ld4<T>(base, (long)r * rowstride + 2 * e + 8 * c + 4, vvb[u] + 4);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { kvb[u][j] = 0.f; vvb[u][j] = 0.f; }
        }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int it = (p0 + u) * NTHR + tid;
        const int r = it / (D / 8), c = it % (D / 8);   // key row, chunk of 8 head-dim elements (K rows keep a 128-byte pitch)
        float kv[8], vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // one f32 number for both halves of every pair (split_dev.h sp_value)
            kv[j] = sp_value(kvb[u][j]);
            vv[j] = sp_value(vvb[u][j]);
        }
        if (X3) {  // range contract of the fp16 pairs (split_dev.h sp_commit)
#pragma unroll
            for (int j = 0; j < 8; j += 2) amax = sp_max2(sp_max2(amax, kv[j], kv[j + 1]), vv[j], vv[j + 1]);
        }
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint16_t h0 = Op::bits(kv[2 * j]), h1 = Op::bits(kv[2 * j + 1]);
            hw[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            lw[j] = (uint32_t)Op::bits(kv[2 * j] - Op::val(h0)) | ((uint32_t)Op::bits(kv[2 * j + 1] - Op::val(h1)) << 16);
        }
        *reinterpret_cast<uint4*>(khi + att_swz(r, c)) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        if (X3) *reinterpret_cast<uint4*>(klo + att_swz(r, c)) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        // V^T: key r sits at k-position (r>>5)*32 + ((r&15)>>2)*8 + ((r>>4)&1)*4 + (r&3) of every head-dim row
        const int kpos = (r >> 5) * 32 + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint16_t hv = Op::bits(vv[j]);
            *reinterpret_cast<uint16_t*>(vhi + (8 * c + j) * VROW + kpos * 2) = hv;
            if (X3) *reinterpret_cast<uint16_t*>(vlo + (8 * c + j) * VROW + kpos * 2) = Op::bits(vv[j] - Op::val(hv));
        }
    }
    }
    __syncthreads();

#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const int tq = wv + ATTM_WAVES * qi;
        if (tq >= nqt) break;
        const int qrow = tq * 16 + q16;
        // ---- Q fragments (B operand): this lane's query row, head-dim 32ks + 8g .. +7, pre-scaled
        frag_t qh[KS], ql[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float (&qv)[8] = qraw[qi][ks];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = sp_value(qv[j] * scale);
                if (X3) amax = __builtin_fmaxf(amax, __builtin_fabsf(x));
                const elem_t hh = (elem_t)x;
                qh[ks][j] = hh;
                ql[ks][j] = (elem_t)(x - (float)hh);
            }
        }
        // ---- scores^T: key tiles x this query tile
        att_f32x4_t sc[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            sc[t] = att_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = att_swz(t * 16 + q16, ks * 4 + g);
                const frag_t kh = *reinterpret_cast<const frag_t*>(khi + off);
                if (X3) {
                    const frag_t kl = *reinterpret_cast<const frag_t*>(klo + off);
                    sc[t] = Op::mfma(kl, qh[ks], sc[t]);
                    sc[t] = Op::mfma(kh, ql[ks], sc[t]);
                }
                sc[t] = Op::mfma(kh, qh[ks], sc[t]);
            }
        }
        // ---- softmax over keys (register r of tile t is key 16t + 4g + r); padded keys contribute nothing
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * t + 4 * g + r >= s) sc[t][r] = -INFINITY;
                mx = fmaxf(mx, sc[t][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (16 * t + 4 * g + r < s) ? expf(sc[t][r] - mx) : 0.f;
                sc[t][r] = pv;
                den += pv;
            }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        // ---- O^T = V^T . P^T over key blocks of 32
        att_f32x4_t oc[TV];
#pragma unroll
        for (int tv = 0; tv < TV; ++tv) oc[tv] = att_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKT / 2; ++kb) {
            frag_t ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = sp_value(sc[2 * kb + (j >> 2)][j & 3]);
                const elem_t hh = (elem_t)x;
                ph[j] = hh;
                pl[j] = (elem_t)(x - (float)hh);
            }
#pragma unroll
            for (int tv = 0; tv < TV; ++tv) {
                const int off = (tv * 16 + q16) * VROW + (kb * 4 + g) * 16;
                const frag_t vh = *reinterpret_cast<const frag_t*>(vhi + off);
                if (X3) {
                    const frag_t vl = *reinterpret_cast<const frag_t*>(vlo + off);
                    oc[tv] = Op::mfma(vl, ph, oc[tv]);
                    oc[tv] = Op::mfma(vh, pl, oc[tv]);
                }
                oc[tv] = Op::mfma(vh, ph, oc[tv]);
            }
        }
        // ---- normalise and store: registers of tile tv are head-dim 16tv + 4g + r of query lane&15
        if (qrow < s) {
            const float inv = 1.f / den;
#pragma unroll
            for (int tv = 0; tv < TV; ++tv) {
                float o4[4] = {oc[tv][0] * inv, oc[tv][1] * inv, oc[tv][2] * inv, oc[tv][3] * inv};
                st4<TO>(out, ((long)b * s + qrow) * e + h * D + 16 * tv + 4 * g, o4, ovf);
            }
        }
    }
    if (X3) sp_count_now(ovf, amax);
}

// ------------------------------------------------------------------------------------------------ audio head
// audio_8_cl.py:151-152: MaxPool1d(5) (stride 5, floor) then ReLU over time of [n, t_in, c].
// ysp: the same values once more as sp32 pairs (the operand of the head's second convolution in the x3 mode)
__global__ void maxpool1d_relu_kernel(const float* __restrict__ x, float* __restrict__ y, sp32_t* __restrict__ ysp, int n, int t_in,
                                      int t_out, int c, int k, unsigned* ovf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * t_out * c) return;
    const int cc = idx % c;
    const int t = (idx / c) % t_out;
    const int b = idx / ((long)c * t_out);
    float m = -INFINITY;
    bool nan = false;
    for (int i = 0; i < k; ++i) {
        const float v = x[((long)b * t_in + t * k + i) * c + cc];
        m = fmaxf(m, v);
        nan |= v != v;
    }
    const float r = nan ? NAN : relu_nan(m);
    y[idx] = r;
    if (ysp) stf<sp32_t>(ysp, idx, r, ovf);
}

// audio_8_cl.py:155-156: AdaptiveAvgPool1d(1) then ReLU: [n, t, c] -> [n, c]
__global__ void mean_time_relu_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int t, int c) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * c) return;
    const int cc = idx % c;
    const int b = idx / c;
    float s = 0.f;
    for (int i = 0; i < t; ++i) s += x[((long)b * t + i) * c + cc];
    y[idx] = relu_nan(s / (float)t);
}

// The sp32 split of an activation-shaped tensor (avcer_split_weights; also the LSTM's hidden state): per group of 32
// elements, 32 hi then 32 lo values of the split type (x = hi + lo + O(2^-22 |x|), split_dev.h); same 4 bytes per element
// as f32, so the DMA addressing is unchanged.  No scaling: activations are stored as they are.
__global__ void split_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, size_t n, unsigned* ovf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = sp_value(w[i]);
    sp_count_now(ovf, __builtin_fabsf(v));  // range contract of an unscaled split (split_dev.h)
    const uint16_t h = f2sp(v);
    const size_t g = i >> 5, j = i & 31;
    out[g * 64 + j] = h;
    out[g * 64 + 32 + j] = f2sp(v - sp2f(h));
}

// max |w| of a tensor as float bits (non-negative floats order like their bit patterns; NaN / inf end up largest)
__global__ void absmax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ slot) {
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(slot, m);
}

// The power of two a weight matrix is multiplied by before its fp16 split, from its largest magnitude (float bits): the one
// that puts that magnitude into [2^14, 2^15) -- hi halves stay under fp16's 65504, and a weight's lo half stays a NORMAL
// fp16 number down to 2^-18 of the largest weight (split_dev.h).  Returns (scale, 1 / scale) as float bits; (1, 1) for an
// all-zero, non-finite or absurdly scaled tensor, and always in the bf16 lab build (bf16 has f32's range).
__device__ __forceinline__ uint2 split_scale_bits(unsigned maxbits) {
    const unsigned e = (maxbits >> 23) & 0xffu;  // biased exponent of max |w|
#if defined(AVCER_SPLIT_BF16)
    return make_uint2(0x3f800000u, 0x3f800000u);
#else
    if (e < 15u || e > 254u) return make_uint2(0x3f800000u, 0x3f800000u);
    return make_uint2((268u - e) << 23, (e - 14u) << 23);  // 2^(141 - e), 2^(e - 141): max |w| * scale in [2^14, 2^15)
#endif
}

// The split of a WEIGHT matrix [n][k] for the x3 contractions: the layout above along K, every value multiplied by the
// tensor's power-of-two scale first, and the rows of every group of 32 output channels re-ordered: stored row 16t + 4g + r
// holds channel 8g + 4t + r (t = 0,1; g = 0..3; r = 0..3).  With weights as the MFMA A operand a lane's accumulators are 4
// consecutive ROWS of a 16-row tile, so this order leaves lane group g of two adjacent tiles with the 8 consecutive
// channels 8g..8g+7: a 16-byte piece of the output row (direct whole-line stores, no LDS staging) and, in fused.hip, the
// B fragment of the next contraction in natural K order.  Trailer behind the n * k * 4 bytes (AVCER_SPLIT_TRAILER_BYTES):
// word 1 = max |w| as float bits (written by absmax_kernel before this launch), word 0 = 1 / scale as a float, written
// here: the multiplier every consumer applies to its accumulators.
__global__ void split_weight_rows_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int n, int k) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned* trailer = reinterpret_cast<unsigned*>(out + (size_t)n * k * 2);
    const uint2 sc = split_scale_bits(trailer[1]);
    if (i == 0) trailer[0] = sc.y;
    if (i >= (size_t)n * k) return;
    const int row = i / k, col = i - (size_t)row * k;      // destination row / K index
    const int j = row & 31, t = j >> 4, g = (j >> 2) & 3, r = j & 3;
    const int src = (row & ~31) + 8 * g + 4 * t + r;
    const float v = sp_value(w[(size_t)src * k + col] * __uint_as_float(sc.x));
    const uint16_t h = f2sp(v);
    const size_t o = (size_t)row * k * 2 + (size_t)(col >> 5) * 64 + (col & 31);
    out[o] = h;
    out[o + 32] = f2sp(v - sp2f(h));
}

// The row-split weights once more, in MFMA fragment order for conv_gemm_wd_kernel (gemm.hip): [N/16][K/32][hi, lo][64 lanes]
// [8 x 16 bit], lane l of a fragment = stored row 16 nt + (l & 15), K elements 8 (l >> 4) .. + 8 of the K-step -- the 16 bytes
// that lane feeds the MFMA as its A operand, so a wave fetches a whole fragment with one coalesced 1 KiB load.  `rows` is
// the output of split_weight_rows_kernel (rows already permuted, hi / lo per 32-element K group); one thread per 16 bytes.
// The trailer (scale words) is copied behind the fragments.
__global__ void weight_frags_kernel(const uint4* __restrict__ rows, uint4* __restrict__ out, int n, int k) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // destination 16-byte piece
    const int nk = k >> 5;
    const size_t pieces = (size_t)n * nk * 8;
    if (i < AVCER_SPLIT_TRAILER_BYTES / 16) out[pieces + i] = rows[pieces + i];
    if (i >= pieces) return;
    const int lane = i & 63, hl = (i >> 6) & 1;
    const size_t f = i >> 7;  // (n tile, K-step)
    const int ks = (int)(f % nk), nt = (int)(f / nk);
    const int row = nt * 16 + (lane & 15), chunk = lane >> 4;
    // source: row-major sp32, 128 bytes per (row, K-step): 64 hi then 64 lo, 16-byte chunks
    out[i] = rows[((size_t)row * nk + ks) * 8 + hl * 4 + chunk];
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f2bf(x[i]);
}

// ------------------------------------------------------------------------------------------------ fusion
// get_prob_audio_8_cl.py:94-101 + run.py:90: frame f's audio logits = mean over the windows whose frame span
// [lo, hi) contains f (pandas float32 group mean).
__global__ void frame_mean_kernel(const float* __restrict__ win, const int32_t* __restrict__ lo,
                                  const int32_t* __restrict__ hi, int n_win, int c, int n_frames, float* __restrict__ out,
                                  int32_t* __restrict__ count) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * c) return;
    const int f = idx / c, j = idx % c;
    double s = 0.0;
    int cnt = 0;
    for (int w = 0; w < n_win; ++w)
        if (lo[w] <= f && f < hi[w]) { s += (double)win[(long)w * c + j]; ++cnt; }
    out[idx] = cnt ? (float)(s / cnt) : 0.f;
    if (j == 0 && count) count[f] = cnt;
}

struct FuseParams {
    double w[21];  // weights_1[m][k] * weights_2[m]
    double pair_w[14];
    int has_w1, cmask;
};

__device__ __forceinline__ void softmax7(const float* in, float* out) {
    float mx = in[0];
#pragma unroll
    for (int k = 1; k < 7; ++k) mx = fmaxf(mx, in[k]);  // NaN handling below
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 7; ++k) nan |= in[k] != in[k];
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) { out[k] = expf(in[k] - mx); den += out[k]; }
#pragma unroll
    for (int k = 0; k < 7; ++k) out[k] = nan ? NAN : out[k] / den;
}

// run.py:85-165 + data/utils.py:222-241.  One thread per frame; float32 tables promoted to float64 at the
// weight multiply exactly like numpy does for `float32 ndarray * python list`.
__global__ void fuse_kernel(const float* __restrict__ stat, const float* __restrict__ dyn, const float* __restrict__ aud,
                            int n, int n_aud, int aud_c, FuseParams fp, double* __restrict__ comp_prob,
                            int32_t* __restrict__ comp_argmax) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const int order[7] = {0, 6, 5, 4, 1, 2, 3};  // video column -> audio order (get_prob_video.py:56-64, run.py:56-65)
    const int p1[7] = {3, 4, 5, 2, 1, 3, 1}, p2[7] = {6, 6, 6, 6, 6, 5, 5};  // run.py:66-74
    float s[7], dl[7], d[7], al[7], a[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { s[k] = stat[(long)f * 7 + order[k]]; dl[k] = dyn[(long)f * 7 + order[k]]; }
    softmax7(dl, d);
    const int fa = f < n_aud ? f : n_aud - 1;  // run.py:99-103 tail padding with the last audio row
#pragma unroll
    for (int k = 0; k < 7; ++k) al[k] = aud[(long)fa * aud_c + k];
    softmax7(al, a);
    if (fp.has_w1) {
        // float32 table * python list -> float64 (run.py:108-111)
        double pm[4][7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            pm[1][k] = (double)s[k] * fp.w[k];
            pm[2][k] = (double)d[k] * fp.w[7 + k];
            pm[3][k] = (double)a[k] * fp.w[14 + k];
            pm[0][k] = pm[1][k] + pm[2][k] + pm[3][k];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            double q[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) q[k] = fp.cmask ? (pm[m][k] > 1.0 / 7.0 ? pm[m][k] : 0.0) : pm[m][k];
            double best = 0.0;
            int bi = 0;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const double v = q[p1[c]] * fp.pair_w[2 * c] + q[p2[c]] * fp.pair_w[2 * c + 1];
                comp_prob[((long)m * n + f) * 7 + c] = v;
                if (c == 0) { best = v; bi = 0; }
                else if (!(best != best) && (v > best || v != v)) { best = v; bi = c; }  // numpy argmax: first NaN wins
            }
            comp_argmax[(long)m * n + f] = bi;
        }
    } else {
        // run.py:113-114: np.sum of float32 tables / 3 stays float32, and so does get_compound_expression's
        // arithmetic (python scalars are weak); only the store into the float64 `prob` array widens.
        float pm[4][7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            pm[1][k] = s[k]; pm[2][k] = d[k]; pm[3][k] = a[k];
            pm[0][k] = ((s[k] + d[k]) + a[k]) / 3.0f;
        }
        const float thr = (float)(1.0 / 7.0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float q[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) q[k] = fp.cmask ? (pm[m][k] > thr ? pm[m][k] : 0.0f) : pm[m][k];
            float best = 0.0f;
            int bi = 0;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const float v = q[p1[c]] * (float)fp.pair_w[2 * c] + q[p2[c]] * (float)fp.pair_w[2 * c + 1];
                comp_prob[((long)m * n + f) * 7 + c] = (double)v;
                if (c == 0) { best = v; bi = 0; }
                else if (!(best != best) && (v > best || v != v)) { best = v; bi = c; }
            }
            comp_argmax[(long)m * n + f] = bi;
        }
    }
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace

#define CHECK_LAUNCH(ctx, name)                                                                      \
    do {                                                                                             \
        hipError_t _e = hipGetLastError();                                                           \
        if (_e != hipSuccess) return set_err((ctx), AVCER_EHIP, name " launch: %s", hipGetErrorString(_e)); \
    } while (0)

int k_preprocess(avcer_ctx* ctx, const uint8_t* frames, int n, int in_h, int in_w, void* out, int kind, hipStream_t st) {
    const long total = (long)n * PP * PP;
    if (kind == 3)
        preprocess_planar_kernel<<<cdiv(total, 256), 256, 0, st>>>(frames, (bf16_t*)out, (bf16_t*)out + total * 4, n, in_h, in_w);
    else if (kind == 1) preprocess_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>(frames, (bf16_t*)out, n, in_h, in_w);
    else preprocess_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(frames, (float*)out, n, in_h, in_w);
    CHECK_LAUNCH(ctx, "preprocess");
    return AVCER_OK;
}

// `kind` of activation storage: 0 = f32, 1 = bf16, 2 = sp32
int k_maxpool3s2(avcer_ctx* ctx, const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int kind, hipStream_t st) {
    const long total = (long)n * oh * ow * (c / 4);
    const int grid = cdiv(total, 256);
    if (kind == 1) maxpool3s2_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, n, h, w, c, oh, ow);
    else if (kind == 2) maxpool3s2_kernel<sp32_t><<<grid, 256, 0, st>>>((const sp32_t*)x, (sp32_t*)y, n, h, w, c, oh, ow);
    else maxpool3s2_kernel<float><<<grid, 256, 0, st>>>((const float*)x, (float*)y, n, h, w, c, oh, ow);
    CHECK_LAUNCH(ctx, "maxpool3s2");
    return AVCER_OK;
}

int k_avgpool_hw(avcer_ctx* ctx, const void* x, float* y, void* y_sp32, int n, int hw, int c, int kind, hipStream_t st) {
    if (c % 32) return set_err(ctx, AVCER_EINVAL, "avgpool: c=%d must be a multiple of 32", c);
    const long total = (long)n * (c / 8);
    // 64 threads per block: one frame per call is 256 threads, and four CUs run them in the time of one
    const int grid = cdiv(total, 64);
    sp32_t* ysp = (sp32_t*)y_sp32;
    if (kind == 1) avgpool_kernel<bf16_t><<<grid, 64, 0, st>>>((const bf16_t*)x, y, ysp, n, hw, c, ctx->ovf);
    else if (kind == 2) avgpool_kernel<sp32_t><<<grid, 64, 0, st>>>((const sp32_t*)x, y, ysp, n, hw, c, ctx->ovf);
    else avgpool_kernel<float><<<grid, 64, 0, st>>>((const float*)x, y, ysp, n, hw, c, ctx->ovf);
    CHECK_LAUNCH(ctx, "avgpool");
    return AVCER_OK;
}

int k_small_linear(avcer_ctx* ctx, const float* x, const float* w, const float* b, float* logits, float* probs, int m,
                   int k, int n, int relu_in, hipStream_t st) {
    if (n > 16) return set_err(ctx, AVCER_EINVAL, "small_linear: n=%d > 16", n);
    small_linear_kernel<<<cdiv(m, 4), 256, 0, st>>>(x, w, b, logits, probs, m, k, n, relu_in);
    CHECK_LAUNCH(ctx, "small_linear");
    return AVCER_OK;
}

int k_lstm_cell(avcer_ctx* ctx, const float* xproj, int64_t xproj_ld, const float* hproj, float* c, float* h_out, void* h_sp,
                int64_t h_ld, int n, int hid, int first, hipStream_t st) {
    if (h_sp && h_ld % 32) return set_err(ctx, AVCER_EINVAL, "lstm_cell: sp32 rows need a stride in whole groups of 32");
    lstm_cell_kernel<<<cdiv((long)n * hid, 256), 256, 0, st>>>(xproj, xproj_ld, hproj, c, h_out, (sp32_t*)h_sp, h_ld, n, hid, first,
                                                               ctx->ovf);
    CHECK_LAUNCH(ctx, "lstm_cell");
    return AVCER_OK;
}

int k_wav_normalize(avcer_ctx* ctx, const float* x, float* y, int n, int t, hipStream_t st) {
    wav_normalize_kernel<<<n, 512, 0, st>>>(x, y, t);
    CHECK_LAUNCH(ctx, "wav_normalize");
    return AVCER_OK;
}

int k_conv0_ln_gelu(avcer_ctx* ctx, const float* x, const float* w, const float* b, const float* g, const float* beta,
                    void* y, int n, int t_in, int t_out, int kind, hipStream_t st) {
    const int spb = 64;
    dim3 grid(cdiv(t_out, spb), n);
    if (kind == 1) conv0_ln_gelu_kernel<bf16_t><<<grid, 256, 0, st>>>(x, w, b, g, beta, (bf16_t*)y, t_in, t_out, spb, nullptr);
    else if (kind == 2) conv0_ln_gelu_kernel<sp32_t><<<grid, 256, 0, st>>>(x, w, b, g, beta, (sp32_t*)y, t_in, t_out, spb, ctx->ovf);
    else conv0_ln_gelu_kernel<float><<<grid, 256, 0, st>>>(x, w, b, g, beta, (float*)y, t_in, t_out, spb, nullptr);
    CHECK_LAUNCH(ctx, "conv0_ln_gelu");
    return AVCER_OK;
}

namespace {
template <typename TI, typename OB>
int ln_launch(avcer_ctx* ctx, const void* x, const void* res, const float* g, const float* b, void* yf, void* yb, int64_t rows,
              int c, float eps, int act, hipStream_t st) {
    const int grid = cdiv(rows, 4);
    if (c == 512)
        layernorm_kernel<TI, OB, 512><<<grid, 256, 0, st>>>((const TI*)x, (const TI*)res, g, b, (float*)yf, (OB*)yb, rows, eps, act, ctx->ovf);
    else if (c == 1024)
        layernorm_kernel<TI, OB, 1024><<<grid, 256, 0, st>>>((const TI*)x, (const TI*)res, g, b, (float*)yf, (OB*)yb, rows, eps, act, ctx->ovf);
    else
        return set_err(ctx, AVCER_EINVAL, "layernorm: c=%d unsupported", c);
    return AVCER_OK;
}
}  // namespace

// in_kind: storage of x / res (0 f32, 1 bf16, 2 sp32); yb_kind: storage of the operand copy yb (1 bf16, 2 sp32)
int k_layernorm(avcer_ctx* ctx, const void* x, const void* res, const float* g, const float* b, void* yf, void* yb,
                int64_t rows, int c, float eps, int act, int in_kind, int yb_kind, hipStream_t st) {
    int r;
    if (yb_kind == 2) {
        if (in_kind == 2) r = ln_launch<sp32_t, sp32_t>(ctx, x, res, g, b, yf, yb, rows, c, eps, act, st);
        else if (in_kind == 1) r = ln_launch<bf16_t, sp32_t>(ctx, x, res, g, b, yf, yb, rows, c, eps, act, st);
        else r = ln_launch<float, sp32_t>(ctx, x, res, g, b, yf, yb, rows, c, eps, act, st);
    } else {
        if (in_kind == 2) r = ln_launch<sp32_t, bf16_t>(ctx, x, res, g, b, yf, yb, rows, c, eps, act, st);
        else if (in_kind == 1) r = ln_launch<bf16_t, bf16_t>(ctx, x, res, g, b, yf, yb, rows, c, eps, act, st);
        else r = ln_launch<float, bf16_t>(ctx, x, res, g, b, yf, yb, rows, c, eps, act, st);
    }
    if (r != AVCER_OK) return r;
    CHECK_LAUNCH(ctx, "layernorm");
    return AVCER_OK;
}

int k_add_pe(avcer_ctx* ctx, const float* x, const float* pe, float* yf, void* yb, int n, int s, int c, int yb_kind,
             hipStream_t st) {
    const long total4 = (long)n * s * c / 4;
    if (yb_kind == 2) add_pe_kernel<sp32_t><<<cdiv(total4, 256), 256, 0, st>>>(x, pe, yf, (sp32_t*)yb, total4, s, c, ctx->ovf);
    else add_pe_kernel<bf16_t><<<cdiv(total4, 256), 256, 0, st>>>(x, pe, yf, (bf16_t*)yb, total4, s, c, nullptr);
    CHECK_LAUNCH(ctx, "add_pe");
    return AVCER_OK;
}

// in_kind: storage of qkv (0 f32, 1 bf16); out_kind: storage of the context vectors (0 f32, 1 bf16, 2 sp32)
int k_attention(avcer_ctx* ctx, const void* qkv, void* out, int n, int s, int heads, int d, float scale, int in_kind,
                int out_kind, hipStream_t st) {
    if (s > 256 || s < 1) return set_err(ctx, AVCER_EINVAL, "attention: S=%d outside [1,256]", s);
    if (d != 32 && d != 64) return set_err(ctx, AVCER_EINVAL, "attention: head dim %d", d);
    if (in_kind == 2 || (in_kind == 1) != (out_kind == 1))
        return set_err(ctx, AVCER_EINVAL, "attention: unsupported storage combination %d -> %d", in_kind, out_kind);
    const int grid = n * heads;
    // bf16 / split-fp16 modes: QK^T and PV on the MFMA, 64- and 32-wide heads (the f32 mode keeps exact f32 arithmetic)
    if (in_kind == 1 || out_kind == 2) {
        const int nkt = s <= 128 ? 8 : 16;
        const int sp = nkt * 16, x3 = out_kind == 2;
        const size_t lds_m = (size_t)sp * 128 * (x3 ? 2 : 1) + (size_t)d * (sp * 2 + 16) * (x3 ? 2 : 1);
#define ATTM(T, TO, NKT, X3, D)                                                                                       \
    do {                                                                                                              \
        static uint64_t attr_dev = 0; /* the attribute is per device: one bit per device index */                    \
        if (!((attr_dev >> (ctx->device & 63)) & 1)) {                                                                \
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)attention_mfma_kernel<T, TO, NKT, X3, D>,                   \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
            attr_dev |= 1ull << (ctx->device & 63);                                                                   \
        }                                                                                                             \
        attention_mfma_kernel<T, TO, NKT, X3, D><<<grid, 64 * ATTM_WAVES, lds_m, st>>>((const T*)qkv, (TO*)out, s, heads, scale, ctx->ovf); \
    } while (0)
#define ATTM_D(T, TO, NKT, X3) do { if (d == 64) ATTM(T, TO, NKT, X3, 64); else ATTM(T, TO, NKT, X3, 32); } while (0)
        if (x3) { if (nkt == 8) ATTM_D(float, sp32_t, 8, 1); else ATTM_D(float, sp32_t, 16, 1); }
        else { if (nkt == 8) ATTM_D(bf16_t, bf16_t, 8, 0); else ATTM_D(bf16_t, bf16_t, 16, 0); }
#undef ATTM_D
#undef ATTM
        CHECK_LAUNCH(ctx, "attention_mfma");
        return AVCER_OK;
    }
    const size_t lds = ((size_t)s * (d + 4) + (size_t)s * d + ATT_WAVES * 256 + ATT_WAVES * d) * sizeof(float);
#define ATT(T, TO, D)                                                                                                \
    do {                                                                                                             \
        static uint64_t attr_dev = 0;                                                                                \
        if (!((attr_dev >> (ctx->device & 63)) & 1)) {                                                               \
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)attention_kernel<T, TO, D>,                                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
            attr_dev |= 1ull << (ctx->device & 63);                                                                  \
        }                                                                                                            \
        attention_kernel<T, TO, D><<<grid, ATT_THREADS, lds, st>>>((const T*)qkv, (TO*)out, s, heads, scale);        \
    } while (0)
    if (in_kind == 1) { if (d == 64) ATT(bf16_t, bf16_t, 64); else ATT(bf16_t, bf16_t, 32); }
    else if (out_kind == 2) { if (d == 64) ATT(float, sp32_t, 64); else ATT(float, sp32_t, 32); }
    else { if (d == 64) ATT(float, float, 64); else ATT(float, float, 32); }
#undef ATT
    CHECK_LAUNCH(ctx, "attention");
    return AVCER_OK;
}

int k_maxpool1d_relu(avcer_ctx* ctx, const float* x, float* y, void* y_sp32, int n, int t_in, int t_out, int c, int k, hipStream_t st) {
    if (y_sp32 && c % 32) return set_err(ctx, AVCER_EINVAL, "maxpool1d_relu: sp32 rows need c in whole groups of 32");
    maxpool1d_relu_kernel<<<cdiv((long)n * t_out * c, 256), 256, 0, st>>>(x, y, (sp32_t*)y_sp32, n, t_in, t_out, c, k, ctx->ovf);
    CHECK_LAUNCH(ctx, "maxpool1d_relu");
    return AVCER_OK;
}

int k_mean_time_relu(avcer_ctx* ctx, const float* x, float* y, int n, int t, int c, hipStream_t st) {
    mean_time_relu_kernel<<<cdiv((long)n * c, 256), 256, 0, st>>>(x, y, n, t, c);
    CHECK_LAUNCH(ctx, "mean_time_relu");
    return AVCER_OK;
}

int k_f32_to_bf16(avcer_ctx* ctx, const float* x, bf16_t* y, size_t n, hipStream_t st) {
    f32_to_bf16_kernel<<<cdiv((long)n, 256), 256, 0, st>>>(x, y, n);
    CHECK_LAUNCH(ctx, "f32_to_bf16");
    return AVCER_OK;
}

int k_frame_mean(avcer_ctx* ctx, const float* win_logits, const int32_t* lo, const int32_t* hi, int n_win, int c,
                 int n_frames, float* out, int32_t* count, hipStream_t st) {
    frame_mean_kernel<<<cdiv((long)n_frames * c, 128), 128, 0, st>>>(win_logits, lo, hi, n_win, c, n_frames, out, count);
    CHECK_LAUNCH(ctx, "frame_mean");
    return AVCER_OK;
}

int k_fuse(avcer_ctx* ctx, const float* stat, const float* dyn, const float* aud, int n, int n_aud, int aud_c,
           const double* w, int has_w1, int cwt, int cmask, double* comp_prob, int32_t* comp_argmax, hipStream_t st) {
    FuseParams fp;
    for (int i = 0; i < 21; ++i) fp.w[i] = w ? w[i] : 1.0;
    // data/utils.py:228-236 with dict_weights of run.py:116-123 (Rule 2) or 1,1
    static const int dictw[7] = {0, 5, 6, 5, 6, 4, 2};
    static const int p1[7] = {3, 4, 5, 2, 1, 3, 1}, p2[7] = {6, 6, 6, 6, 6, 5, 5};
    for (int c = 0; c < 7; ++c) {
        if (cwt) {
            const double sw = (double)(dictw[p1[c]] + dictw[p2[c]]);
            fp.pair_w[2 * c] = dictw[p1[c]] / sw;
            fp.pair_w[2 * c + 1] = dictw[p2[c]] / sw;
        } else {
            fp.pair_w[2 * c] = 1.0;
            fp.pair_w[2 * c + 1] = 1.0;
        }
    }
    fp.has_w1 = has_w1;
    fp.cmask = cmask;
    fuse_kernel<<<cdiv(n, 64), 64, 0, st>>>(stat, dyn, aud, n, n_aud, aud_c, fp, comp_prob, comp_argmax);
    CHECK_LAUNCH(ctx, "fuse");
    return AVCER_OK;
}

int k_face_decode(avcer_ctx* ctx, const float* loc, const float* conf, const float* landms, const float* priors, int T, int P,
                  int im_h, int im_w, float var0, float var1, float* dets, hipStream_t st) {
    face_decode_kernel<<<dim3(cdiv(P, 256), T), 256, 0, st>>>(loc, conf, landms, priors, P, (float)im_w, (float)im_h, var0, var1, dets);
    CHECK_LAUNCH(ctx, "face_decode");
    return AVCER_OK;
}

int k_crop_tiles(avcer_ctx* ctx, const uint8_t* frames, int T, int H, int W, const int32_t* rects, int n, int swap_rb,
                 uint8_t* tiles, hipStream_t st) {
    crop_tiles_kernel<<<cdiv((long)n * 224 * 56, 256), 256, 0, st>>>(frames, T, H, W, rects, n, swap_rb, tiles);
    CHECK_LAUNCH(ctx, "crop_tiles");
    return AVCER_OK;
}

int k_face_nms(avcer_ctx* ctx, const float* dets, int T, int P, float conf_thresh, float nms_thresh, int nms_top_k, int top_k,
               float threshold, int32_t* order, int32_t* count, float* out, int32_t* out_n, hipStream_t st) {
    if (nms_top_k > NMS_MAX) return set_err(ctx, AVCER_EINVAL, "face_nms: nms_top_k %d exceeds %d", nms_top_k, NMS_MAX);
    static uint64_t attr_dev = 0;
    if (!((attr_dev >> (ctx->device & 63)) & 1)) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)face_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_MAX * 8));
        attr_dev |= 1ull << (ctx->device & 63);
    }
    if (P <= SORT_MAX) {
        int N = 2 * SORT_THREADS;  // LDS for the worst case (every prior a candidate); the kernel sorts the next power of two of ITS count
        while (N < P) N <<= 1;
        face_sort_kernel<<<T, SORT_THREADS, (size_t)N * 8, st>>>(dets, P, conf_thresh, nms_top_k, order, count);
        CHECK_LAUNCH(ctx, "face_sort");
    } else {
        if (hipMemsetAsync(count, 0, (size_t)T * 4, st) != hipSuccess) return set_err(ctx, AVCER_EHIP, "face_nms: memset failed");
        face_rank_kernel<<<dim3(cdiv(P, 256), T), 256, 0, st>>>(dets, P, conf_thresh, nms_top_k, order, count);
        CHECK_LAUNCH(ctx, "face_rank");
    }
    face_nms_kernel<<<T, NMS_THREADS, 0, st>>>(dets, P, order, count, nms_top_k, nms_thresh, top_k, threshold, out, out_n);
    CHECK_LAUNCH(ctx, "face_nms");
    return AVCER_OK;
}

int k_face_pre(avcer_ctx* ctx, const uint8_t* frames, int n, int h, int w, int ph, int pw, int rgb, void* out, int bf16,
               hipStream_t st) {
    const long total = (long)n * ph * pw;
    if (bf16) face_pre_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>(frames, (bf16_t*)out, n, h, w, ph, pw, rgb);
    else face_pre_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(frames, (float*)out, n, h, w, ph, pw, rgb);
    CHECK_LAUNCH(ctx, "face_pre");
    return AVCER_OK;
}

int k_maxpool3s2p1(avcer_ctx* ctx, const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int kind, hipStream_t st) {
    const long total = (long)n * oh * ow * (c / 4);
    const int grid = cdiv(total, 256);
    if (kind == 1) maxpool3s2p1_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, n, h, w, c, oh, ow);
    else if (kind == 2) maxpool3s2p1_kernel<sp32_t><<<grid, 256, 0, st>>>((const sp32_t*)x, (sp32_t*)y, n, h, w, c, oh, ow);
    else maxpool3s2p1_kernel<float><<<grid, 256, 0, st>>>((const float*)x, (float*)y, n, h, w, c, oh, ow);
    CHECK_LAUNCH(ctx, "maxpool3s2p1");
    return AVCER_OK;
}

int k_upsample_add(avcer_ctx* ctx, void* y, const void* coarse, int n, int h, int w, int ch, int cw, int c, int kind,
                   hipStream_t st) {
    const long total = (long)n * h * w * (c / 4);
    const int grid = cdiv(total, 256);
    if (kind == 1) upsample_add_kernel<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)y, (const bf16_t*)coarse, n, h, w, ch, cw, c, nullptr);
    else if (kind == 2) upsample_add_kernel<sp32_t><<<grid, 256, 0, st>>>((sp32_t*)y, (const sp32_t*)coarse, n, h, w, ch, cw, c, ctx->ovf);
    else upsample_add_kernel<float><<<grid, 256, 0, st>>>((float*)y, (const float*)coarse, n, h, w, ch, cw, c, nullptr);
    CHECK_LAUNCH(ctx, "upsample_add");
    return AVCER_OK;
}

int k_face_head(avcer_ctx* ctx, const float* hd, int ld, int n, int hw, int row0, int P, float* loc, float* conf,
                float* landms, hipStream_t st) {
    face_head_kernel<<<cdiv((long)n * hw * 2, 256), 256, 0, st>>>(hd, ld, n, hw, row0, P, loc, conf, landms);
    CHECK_LAUNCH(ctx, "face_head");
    return AVCER_OK;
}

int k_pack_nchw(avcer_ctx* ctx, const float* x, int n, void* out, int kind, hipStream_t st) {
    const long total = (long)n * PP * PP;
    if (kind == 3) pack_nchw_planar_kernel<<<cdiv(total, 256), 256, 0, st>>>(x, (bf16_t*)out, (bf16_t*)out + total * 4, n, ctx->ovf);
    else if (kind == 1) pack_nchw_kernel<bf16_t><<<cdiv(total, 256), 256, 0, st>>>(x, (bf16_t*)out, n);
    else pack_nchw_kernel<float><<<cdiv(total, 256), 256, 0, st>>>(x, (float*)out, n);
    CHECK_LAUNCH(ctx, "pack_nchw");
    return AVCER_OK;
}

int k_gather_windows(avcer_ctx* ctx, const float* feats, const int32_t* idx, int nwin, float* out, hipStream_t st) {
    const long total4 = (long)nwin * 10 * 128;
    gather_windows_kernel<<<cdiv(total4, 256), 256, 0, st>>>(feats, idx, out, total4);
    CHECK_LAUNCH(ctx, "gather_windows");
    return AVCER_OK;
}

int k_audio_chunks(avcer_ctx* ctx, const float* wav, const int32_t* starts, const int32_t* ends, int n, int window,
                   int mode, float* out, hipStream_t st) {
    audio_chunks_kernel<<<n, 512, 0, st>>>(wav, starts, ends, window, mode, out);
    CHECK_LAUNCH(ctx, "audio_chunks");
    return AVCER_OK;
}

int k_split_weight_rows(avcer_ctx* ctx, const float* w, bf16_t* out, int n, int k, hipStream_t st) {
    if (n % 32 || k % 32) return set_err(ctx, AVCER_EINVAL, "split_weight_rows: n=%d and k=%d must be multiples of 32", n, k);
    // trailer behind the split data: max |w| (word 1) -> the tensor's power-of-two scale; its inverse is left in word 0
    unsigned* trailer = reinterpret_cast<unsigned*>(out + (size_t)n * k * 2);
    HIP_TRY(ctx, hipMemsetAsync(trailer, 0, AVCER_SPLIT_TRAILER_BYTES, st));
    absmax_kernel<<<(int)std::min<long>(cdiv((long)n * k, 256 * 16), 2048), 256, 0, st>>>(w, (size_t)n * k, trailer + 1);
    split_weight_rows_kernel<<<cdiv((long)n * k, 256), 256, 0, st>>>(w, out, n, k);
    CHECK_LAUNCH(ctx, "split_weight_rows");
    return AVCER_OK;
}

int k_weight_frags(avcer_ctx* ctx, const bf16_t* rows, bf16_t* out, int n, int k, hipStream_t st) {
    if (n % 16 || k % 32) return set_err(ctx, AVCER_EINVAL, "weight_frags: n=%d must be a multiple of 16, k=%d of 32", n, k);
    weight_frags_kernel<<<cdiv((long)n * (k / 32) * 8, 256), 256, 0, st>>>((const uint4*)rows, (uint4*)out, n, k);
    CHECK_LAUNCH(ctx, "weight_frags");
    return AVCER_OK;
}

int k_split_weights(avcer_ctx* ctx, const float* w, bf16_t* out, size_t n, hipStream_t st) {
    split_weights_kernel<<<cdiv((long)n, 256), 256, 0, st>>>(w, out, n, ctx->ovf);
    CHECK_LAUNCH(ctx, "split_weights");
    return AVCER_OK;
}
