// Implicit-GEMM convolution / linear layer for gfx950 with a fused epilogue.
//
//   Y[m, n] = act(scale[n] * sum_k A[m, k] * W[n, k] + bias[n] (+ R[m, n]))
//
// m enumerates output positions (b, oy, ox) of an NHWC tensor, k enumerates (ky, kx, c) with c contiguous,
// W is [N][K] with K contiguous.  Every contraction of the AVCER hot path maps onto it:
//   ResNet-50 1x1 / 3x3 / 7x7 convolutions      architectures/video.py:7-60,93-166
//   wav2vec2 Conv1d layers 1-6, grouped pos-conv, all Linear layers   (transformers Wav2Vec2Model)
//   TransformerLayer projections / FFN           architectures/attention_layers.py:80-144,41-57
//   time_downsample Conv1d (k5 s3 dil2, k3)      architectures/audio_8_cl.py:146-159
//   LSTM input / recurrent projections           architectures/video.py:169-185
//
// Design (CDNA4):
//   * 256 threads = 4 waves (2x2, wave tile 64 x BN/2); block tile 128(m) x BN(n), BN = 128 or 64; one K-step =
//     128 bytes per row for every element type (32 f32 / sp32 or 64 bf16), so the DMA pattern is type-independent.
//     64 / 48 KiB of LDS per block: two / three blocks per CU cover each other's barrier stalls.
//   * A and W tiles go global -> LDS by DMA (buffer_load_dwordx4 ... lds, no VGPR staging, no ds_write); the
//     hardware bounds check of the buffer descriptor supplies the zeros of image borders and of rows past M.
//     The double-buffered LDS image has 128-byte rows whose 16-byte chunks are XOR-swizzled with a searched key of
//     (row>>1)&7, applied on the SOURCE address (the DMA destination is lane-linear): conflict-free ds_read_b128
//     fragment reads for every MFMA shape used (MI355X LDS: 64 banks x 4 B, b128 reads served in 16-lane groups).
//   * MFMA operands are swapped (weights = A operand, activations = B operand) so that each lane's accumulator
//     registers hold 4 CONSECUTIVE output channels of one position.
//   * Arithmetic: MODE 0 f32 (v_mfma_f32_32x32x2_f32, exact FMA chain); MODE 1 bf16 (v_mfma_f32_16x16x32_bf16);
//     MODE 2/3 split-fp16 "x3": hi/lo fp16 pairs on v_mfma_f32_16x16x32_f16, 3 MFMAs per product, f32-grade results
//     (see conv_gemm_kernel and split_dev.h).
//   * Epilogue through LDS: scale/bias, residual (prefetched into registers at kernel start), activation, whole-line
//     16-byte stores in f32, bf16 or sp32.  Optional second A source (two fused 1x1 convolutions), grouped
//     convolution via grid.y, bijective XCD-aware block remap + grouped tile order.
#include "common.h"
#include "gemm_dev.h"

#include <type_traits>

namespace {

struct GemmParams {
    const char* X;
    const char* W;
    const float* scale;
    const float* bias;
    const char* R;
    char* Y;
    unsigned x_bytes, w_bytes;  // buffer extents for the hardware bounds check (reads past them return 0)
    // optional second A source (fused 1x1 convolutions: K elements [K1, K) come from X2, a 1x1 conv with its own stride)
    const char* X2;
    unsigned x2_bytes;
    int K1;
    long sB2, sH2, sW2;
    int coff2, st2;
    int M, N, K;
    int OH, OW, H, Wd, Cin, KW;
    int sh, sw, ph, pw, dh, dw;
    long sB, sH, sW;
    int coff;
    long ldY;
    int yoff;
    long ldR;
    int roff;
    int act, res_after;
    int ntn, ntm, nwg;
    int gm;  // m-tiles per group of the grouped block order (0 = plain n-fastest order)
    int groups;  // grid.y: group g shifts coff / yoff / roff by g*Cin / g*N and the weight/scale/bias rows by g*N
    int KH;
    int fast;  // pad-free gather with a scalar K / tap advance (see AVCER_ISSUE_TILES)
    int tile_n;  // 0 = choose, 64, 128
    int tile_m;  // dtype 7 / 8: 0 = choose, 112, 128
    int tap_inner;  // K-steps walk (channel chunk, ky, kx) instead of (ky, kx, channel chunk): see launch_conv_gemm
    const char* WF;  // dtype 7 / 8: the weights in MFMA fragment order (kernels.hip weight_frags_kernel), else null
    int tapH4, tapW4;  // byte steps of one filter tap down / right: dil_h * x_stride_h * 4, dil_w * x_stride_w * 4
    int slots;         // host side only: block slots of the device (2 per CU), for the tile choices of the launchers
    int rsub, rH, rW;  // residual sub-sampling (avcer_conv_desc.r_sub): output (b, oy, ox) adds residual row (b, oy*rsub, ox*rsub)
    unsigned* ovf;     // the context's range-contract counter (split_dev.h sp_commit)
};

// Epilogue, staged through LDS so that HBM sees whole 128-byte lines: every wave first parks its scaled/biased
// accumulators in an f32 [128][BN] image (16-byte chunks XOR-swizzled with row&7 against write conflicts), then the
// 256 threads walk the image row-major, 8 consecutive channels per thread: residual add, activation, one 16-byte
// (bf16) or two 16-byte (f32) stores.  ACT: 0 none, 1 relu, 2 gelu (compile-time: erff only in the GELU variant).
// OUT: 0 = f32, 1 = bf16, 2 = split-fp16 pairs ("sp32": per aligned group of 32 channels, 32 hi fp16 then 32 lo fp16,
// value = hi + lo; 4 bytes per element like f32, directly consumable as MODE 3 A operand).  The residual has the
// same storage type as the output.
// Raw 16-byte words of the residual belonging to 8 consecutive channels of one position: {f32 x4, f32 x4},
// {bf16 x8, -} or {sp32 hi x8, sp32 lo x8}.  They are fetched at kernel start (res_prefetch) so that the read overlaps
// the DMA / MFMA phase instead of sitting in the epilogue.
template <int OUT>
__device__ __forceinline__ void res_load(const GemmParams& p, long m, int n0, uint4& r0, uint4& r1) {
    if (p.rsub > 1) {  // the residual's own position grid is rsub x finer than the output's
        const int ohw = p.OH * p.OW;
        const int b = (int)(m / ohw), rem = (int)m - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        m = ((long)b * p.rH + (long)oy * p.rsub) * p.rW + (long)ox * p.rsub;
    }
    const long e = m * p.ldR + p.roff + n0;
    if constexpr (OUT == 0) {
        const char* rp = p.R + e * 4;
        r0 = *reinterpret_cast<const uint4*>(rp);
        r1 = *reinterpret_cast<const uint4*>(rp + 16);
    } else if constexpr (OUT == 1) {
        r0 = *reinterpret_cast<const uint4*>(p.R + e * 2);
        r1 = make_uint4(0u, 0u, 0u, 0u);
    } else {
        const char* rp = p.R + sp32_byte(e);
        r0 = *reinterpret_cast<const uint4*>(rp);
        r1 = *reinterpret_cast<const uint4*>(rp + 64);
    }
}

template <int OUT, int ACT>
__device__ __forceinline__ void finish8(const GemmParams& p, long m, int n0, const float4 a, const float4 b, const uint4 r0,
                                        const uint4 r1, sp_flags_t& ovm) {
#pragma clang fp contract(off)
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float r[8];
    if constexpr (OUT == 0) {
        r[0] = __builtin_bit_cast(float, r0.x); r[1] = __builtin_bit_cast(float, r0.y);
        r[2] = __builtin_bit_cast(float, r0.z); r[3] = __builtin_bit_cast(float, r0.w);
        r[4] = __builtin_bit_cast(float, r1.x); r[5] = __builtin_bit_cast(float, r1.y);
        r[6] = __builtin_bit_cast(float, r1.z); r[7] = __builtin_bit_cast(float, r1.w);
    } else if constexpr (OUT == 1) {
        const uint32_t w[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[2 * j] = bf2f((bf16_t)(w[j] & 0xffff)); r[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16)); }
    } else {
        const uint32_t wh[4] = {r0.x, r0.y, r0.z, r0.w}, wl[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[2 * j] = sp2f((uint16_t)(wh[j] & 0xffff)) + sp2f((uint16_t)(wl[j] & 0xffff));
            r[2 * j + 1] = sp2f((uint16_t)(wh[j] >> 16)) + sp2f((uint16_t)(wl[j] >> 16));
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = p.res_after ? v[j] : v[j] + r[j];
        if constexpr (ACT == 1) x = relu_nan(x);
        if constexpr (ACT == 2) x = gelu_erf(x);
        if constexpr (ACT == 3) x = gelu_fast(x);
        v[j] = p.res_after ? x + r[j] : x;
    }
    const long e = m * p.ldY + p.yoff + n0;
    if constexpr (OUT == 0) {
        char* yp = p.Y + e * 4;
        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(yp + 16) = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (OUT == 1) {
        uint4 t;
        t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        t.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
        t.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
        *reinterpret_cast<uint4*>(p.Y + e * 2) = t;
    } else {
        uint32_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sp_value(v[j]);  // one f32 number for both halves of the pair (split_dev.h)
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            amax = sp_max2(amax, v[2 * j], v[2 * j + 1]);
            const uint16_t h0 = f2sp(v[2 * j]), h1 = f2sp(v[2 * j + 1]);
            h[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            l[j] = (uint32_t)f2sp(v[2 * j] - sp2f(h0)) | ((uint32_t)f2sp(v[2 * j + 1] - sp2f(h1)) << 16);
        }
        sp_flag(ovm, amax);  // range contract: |x| < 65504 (split_dev.h sp_commit)
        char* yp = p.Y + sp32_byte(e);
        *reinterpret_cast<uint4*>(yp) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(yp + 64) = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

template <int BN>
__device__ __forceinline__ int stage_off(int row, int chunk) { return row * (BN * 4) + ((chunk ^ (row & 7)) << 4); }

// row0 = first row of this wave's accumulators inside the staging image
template <int MODE, int BN, typename AccT, int NFN, int NFM>
__device__ __forceinline__ void stage_acc(const GemmParams& p, char* smem, AccT (&acc)[NFN][NFM], int n_base, int row0, int wn,
                                          int lane) {
    constexpr int WN = BN / 2;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int fn = 0; fn < NFN; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * WN + fn * 32 + 8 * g + 4 * (lane >> 5);
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n_base + nl);
                if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + n_base + nl);
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm) {
                    const int ml = row0 + fm * 32 + (lane & 31);
                    *reinterpret_cast<float4*>(smem + stage_off<BN>(ml, nl >> 2)) =
                        make_float4(acc[fn][fm][4 * g + 0] * sc.x + bi.x, acc[fn][fm][4 * g + 1] * sc.y + bi.y,
                                    acc[fn][fm][4 * g + 2] * sc.z + bi.z, acc[fn][fm][4 * g + 3] * sc.w + bi.w);
                }
            }
    } else {
#pragma unroll
        for (int fn = 0; fn < NFN; ++fn) {
            const int nl = wn * WN + fn * 16 + 4 * (lane >> 4);
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n_base + nl);
            if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + n_base + nl);
#pragma unroll
            for (int fm = 0; fm < NFM; ++fm) {
                const int ml = row0 + fm * 16 + (lane & 15);
                *reinterpret_cast<float4*>(smem + stage_off<BN>(ml, nl >> 2)) =
                    make_float4(acc[fn][fm][0] * sc.x + bi.x, acc[fn][fm][1] * sc.y + bi.y, acc[fn][fm][2] * sc.z + bi.z,
                                acc[fn][fm][3] * sc.w + bi.w);
            }
        }
    }
}

template <int BMT, int BN> struct DrainMap {
    static constexpr int EP_ROWS = BMT;
    static constexpr int TPR = BN / 8;          // threads per row (8 channels each)
    static constexpr int RPP = 256 / TPR;       // rows per pass
    static constexpr int NP = EP_ROWS / RPP;    // passes per round
};

template <int OUT, int BMT, int BN>
__device__ __forceinline__ void res_prefetch(const GemmParams& p, int m_base, int n_base, int tid,
                                             uint4 (&rr)[DrainMap<BMT, BN>::NP][2]) {
    using D = DrainMap<BMT, BN>;
    const int c8 = tid % D::TPR, r0 = tid / D::TPR;
#pragma unroll
    for (int pass = 0; pass < D::NP; ++pass) {
        const long m = (long)m_base + pass * D::RPP + r0;
        rr[pass][0] = make_uint4(0u, 0u, 0u, 0u);  // all-zero bits decode to 0.0 in every storage type
        rr[pass][1] = make_uint4(0u, 0u, 0u, 0u);
        if (p.R && m < p.M) res_load<OUT>(p, m, n_base + c8 * 8, rr[pass][0], rr[pass][1]);
    }
}

template <int OUT, int BMT, int BN, int ACT>
__device__ __forceinline__ void drain_stage(const GemmParams& p, const char* smem, int m_base, int n_base, int tid,
                                            const uint4 (&rr)[DrainMap<BMT, BN>::NP][2]) {
    using D = DrainMap<BMT, BN>;
    const int c8 = tid % D::TPR, r0 = tid / D::TPR;
#pragma unroll
    for (int pass = 0; pass < D::NP; ++pass) {
        const int row = pass * D::RPP + r0;
        const long m = (long)m_base + row;
        if (m < p.M) {
            const float4 a = *reinterpret_cast<const float4*>(smem + stage_off<BN>(row, 2 * c8));
            const float4 b = *reinterpret_cast<const float4*>(smem + stage_off<BN>(row, 2 * c8 + 1));
            sp_flags_t unused = 0;  // the staged epilogue never writes sp32 (OUT == 2 belongs to the split-fp16 modes)
            finish8<OUT, ACT>(p, m, n_base + c8 * 8, a, b, rr[pass][0], rr[pass][1], unused);
        }
    }
}

// ---- direct epilogue of the split-fp16 modes (MODE 2 / 3).  Their weight rows are stored permuted inside every group of
// 32 output channels (kernels.hip split_weight_rows_kernel: stored row 16t + 4g + r = channel 8g + 4t + r), so the two
// 16-row accumulator tiles of a group leave lane group g = lane >> 4 with the 8 CONSECUTIVE channels 8g..8g+7 of one
// position: 16 contiguous bytes of an sp32 / bf16 row or 32 of an f32 row.  No LDS staging, no second barrier: residual
// words are requested at kernel start in exactly this layout and the stores cover whole 64-byte half-lines.
template <int OUT, int NFN, int NFM>
__device__ __forceinline__ void res_prefetch_direct(const GemmParams& p, int m0, int c0, int lane, uint4 (&rr)[NFN / 2][NFM][2]) {
#pragma unroll
    for (int j = 0; j < NFN / 2; ++j)
#pragma unroll
        for (int fm = 0; fm < NFM; ++fm) {
            rr[j][fm][0] = make_uint4(0u, 0u, 0u, 0u);
            rr[j][fm][1] = make_uint4(0u, 0u, 0u, 0u);
            const long m = (long)m0 + fm * 16 + (lane & 15);
            if (p.R && m < p.M) res_load<OUT>(p, m, c0 + 32 * j + 8 * (lane >> 4), rr[j][fm][0], rr[j][fm][1]);
        }
}

__device__ __forceinline__ float4 scale_bias4(const f32x4_t a, const float4 s, const float4 b) {
    return make_float4(__builtin_fmaf(a[0], s.x, b.x), __builtin_fmaf(a[1], s.y, b.y), __builtin_fmaf(a[2], s.z, b.z),
                       __builtin_fmaf(a[3], s.w, b.w));
}

template <int OUT, int ACT, int NFN, int NFM>
__device__ __forceinline__ void epilogue_direct(const GemmParams& p, f32x4_t (&acc)[NFN][NFM], int m0, int c0, int lane,
                                                const uint4 (&rr)[NFN / 2][NFM][2], const float wmul, sp_flags_t& ovm) {
#pragma unroll
    for (int j = 0; j < NFN / 2; ++j) {
        const int ch = c0 + 32 * j + 8 * (lane >> 4);  // first of this lane's 8 channels (index into scale / bias too)
        float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (p.scale) { s0 = *reinterpret_cast<const float4*>(p.scale + ch); s1 = *reinterpret_cast<const float4*>(p.scale + ch + 4); }
        if (p.bias) { b0 = *reinterpret_cast<const float4*>(p.bias + ch); b1 = *reinterpret_cast<const float4*>(p.bias + ch + 4); }
        // the weights were split as w * 2^e (split_dev.h): wmul = 2^-e folds back into the channel scale, exactly
        s0 = make_float4(s0.x * wmul, s0.y * wmul, s0.z * wmul, s0.w * wmul);
        s1 = make_float4(s1.x * wmul, s1.y * wmul, s1.z * wmul, s1.w * wmul);
#pragma unroll
        for (int fm = 0; fm < NFM; ++fm) {
            const long m = (long)m0 + fm * 16 + (lane & 15);
            if (m >= p.M) continue;
            const f32x4_t lo = acc[2 * j][fm], hi = acc[2 * j + 1][fm];
            // explicit fma: the two forms of the kernel (this one and conv_gemm_wd_kernel) must round alike whatever hipcc contracts
            finish8<OUT, ACT>(p, m, ch, scale_bias4(lo, s0, b0), scale_bias4(hi, s1, b1), rr[j][fm][0], rr[j][fm][1], ovm);
        }
    }
}

// f32 -> split pair used by MODE 2: x = hi + lo + O(2^-22 |x|) with hi, lo fp16 (round to nearest even; split_dev.h)
__device__ __forceinline__ void split8(const float4 x, const float4 y, spx8_t& hi, spx8_t& lo, sp_flags_t& ovm) {
    const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) amax = sp_max2(amax, v[j], v[j + 1]);
    sp_flag(ovm, amax);  // range contract: |x| < 65504 (split_dev.h sp_commit)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float w = sp_value(v[j]);  // one f32 number for both halves (split_dev.h)
        const spe_t h = (spe_t)w;
        hi[j] = h;
        lo[j] = (spe_t)(w - (float)h);
    }
}

// MODE 0: f32 operands, v_mfma_f32_32x32x2_f32.  MODE 1: bf16 operands, v_mfma_f32_16x16x32_bf16.
// MODE 2 ("x3"): f32 activations split on the fly into fp16 hi+lo, weights pre-split and pre-scaled (per 32-element K group:
// 32 hi then 32 lo fp16), a.w ~= ah.wh + ah.wl + al.wh on the f16 MFMA with f32 accumulation -- f32-grade
// results (relative error ~2^-22 per product) at a third of the 16-bit MFMA rate instead of a sixteenth.
// MODE 3: as MODE 2 but the activations are ALREADY stored as sp32 pairs (written by a producer's epilogue), so the
// A fragments are read like the weights and the main loop has no conversion arithmetic at all.
template <int MODE, int BMT, int BN, int TILE_BYTES, typename AccT, int NFN, int NFM>
__device__ __forceinline__ void mfma_step(const GemmParams& p, const char* smem, int cur, AccT (&acc)[NFN][NFM], int wm, int wn,
                                      int lane, sp_flags_t& ovm) {
    constexpr bool IS_F32 = MODE == 0;
    constexpr int WN = BN / 2;
    constexpr int WM = BMT / 2;
    const char* sa = smem + cur * TILE_BYTES;
    const char* sb = sa + BMT * ROWB;
    if constexpr (IS_F32) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int ch = kq * 2 + (lane >> 5);
            float4 af[NFM], wf[NFN];
#pragma unroll
            for (int fm = 0; fm < NFM; ++fm)
                af[fm] = *reinterpret_cast<const float4*>(sa + swz(wm * WM + fm * 32 + (lane & 31), ch));
#pragma unroll
            for (int fn = 0; fn < NFN; ++fn)
                wf[fn] = *reinterpret_cast<const float4*>(sb + swz(wn * WN + fn * 32 + (lane & 31), ch));
#pragma unroll
            for (int fn = 0; fn < NFN; ++fn)
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm) {
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[fn].x, af[fm].x, acc[fn][fm], 0, 0, 0);
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[fn].y, af[fm].y, acc[fn][fm], 0, 0, 0);
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[fn].z, af[fm].z, acc[fn][fm], 0, 0, 0);
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[fn].w, af[fm].w, acc[fn][fm], 0, 0, 0);
                }
        }
    } else if constexpr (MODE == 2 || MODE == 3) {
        const int g = lane >> 4;
        spx8_t ahi[NFM], alo[NFM];
#pragma unroll
        for (int fm = 0; fm < NFM; ++fm) {
            const int row = wm * WM + fm * 16 + (lane & 15);
            if constexpr (MODE == 3) {
                ahi[fm] = *reinterpret_cast<const spx8_t*>(sa + swz(row, g));
                alo[fm] = *reinterpret_cast<const spx8_t*>(sa + swz(row, 4 + g));
                continue;
            }
            const float4 x = *reinterpret_cast<const float4*>(sa + swz(row, 2 * g));
            const float4 y = *reinterpret_cast<const float4*>(sa + swz(row, 2 * g + 1));
            split8(x, y, ahi[fm], alo[fm], ovm);
        }
#pragma unroll
        for (int fn = 0; fn < NFN; ++fn) {
            const int row = wn * WN + fn * 16 + (lane & 15);
            const spx8_t whi = *reinterpret_cast<const spx8_t*>(sb + swz(row, g));
            const spx8_t wlo = *reinterpret_cast<const spx8_t*>(sb + swz(row, 4 + g));
#pragma unroll
            for (int fm = 0; fm < NFM; ++fm) {
                acc[fn][fm] = mfma_sp(wlo, ahi[fm], acc[fn][fm]);
                acc[fn][fm] = mfma_sp(whi, alo[fm], acc[fn][fm]);
                acc[fn][fm] = mfma_sp(whi, ahi[fm], acc[fn][fm]);
            }
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ks * 4 + (lane >> 4);
            bf16x8_t af[NFM], wf[NFN];
#pragma unroll
            for (int fm = 0; fm < NFM; ++fm)
                af[fm] = *reinterpret_cast<const bf16x8_t*>(sa + swz(wm * WM + fm * 16 + (lane & 15), ch));
#pragma unroll
            for (int fn = 0; fn < NFN; ++fn)
                wf[fn] = *reinterpret_cast<const bf16x8_t*>(sb + swz(wn * WN + fn * 16 + (lane & 15), ch));
#pragma unroll
            for (int fn = 0; fn < NFN; ++fn)
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm)
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], af[fm], acc[fn][fm], 0, 0, 0);
        }
    }
}

template <int MODE, int OUT, int BN>
__global__ void __launch_bounds__(256, 2) conv_gemm_kernel(const GemmParams p) {
    constexpr int BMT = 128;
    constexpr bool IS_F32 = MODE == 0;
    constexpr int ES = MODE == 1 ? 2 : 4;
    constexpr int VEC = 16 / ES;
    constexpr int BK = ROWB / ES;
    constexpr int TILE_BYTES = (BMT + BN) * ROWB;
    constexpr int WN = BN / 2;
    constexpr int WM = BMT / 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int dw = wave;  // every wave issues its share of the DMA

    // XCD-aware bijective remap: blocks b and b+8 share an XCD (speed only, never correctness)
    int bid = blockIdx.x;
    {
        const int q = p.nwg >> 3, r = p.nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Grouped order: gm m-tiles x all n-tiles form a group that is swept n-major, so the ~64 blocks an XCD runs at
    // a time share BOTH their activation slabs (across n) and their weight slabs (across m) in its 4 MiB L2.
    int tile_n, tile_m;
    if (p.gm > 1) {
        const int per_group = p.gm * p.ntn;
        const int group = bid / per_group, within = bid - group * per_group;
        const int first_m = group * p.gm;
        const int gsize = min(p.gm, p.ntm - first_m);
        tile_m = first_m + within % gsize;
        tile_n = within / gsize;
    } else {
        tile_n = bid % p.ntn;
        tile_m = bid / p.ntn;
    }
    const int m_base = tile_m * BMT;
    const int grp = blockIdx.y;
    const int n_base = grp * p.N + tile_n * BN;  // row of W / entry of scale, bias; output channel = yoff + n_base
    const int x_coff = p.coff + grp * p.Cin;

    // hardware-bounds-checked buffer descriptors: an out-of-range voffset returns zeros without a branch
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X), (short)0, (int)p.x_bytes, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W), (short)0, (int)p.w_bytes, 0x00020000);
    const auto x2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X2 ? p.X2 : p.X), (short)0,
                                                        (int)(p.X2 ? p.x2_bytes : p.x_bytes), 0x00020000);

    // Global -> LDS by DMA (buffer_load_dwordx4 ... lds): one wave-instruction fills 1 KiB = 8 tile rows x 128 B,
    // lane i landing at (row i>>3, 16-byte slot i&7).  The XOR swizzle of the LDS image is therefore applied to the
    // SOURCE: lane i fetches data chunk c = slot ^ ((row>>1)&7) of its row.  Padding taps / rows past M use an
    // out-of-range offset, for which the buffer load writes zeros.
    constexpr int A_ISS = BMT / 8 / 4;  // DMA instructions per wave per K-step for the A tile
    constexpr int B_ISS = BN / 8 / 4;
    const int lrow8 = lane >> 3;
    const int slot = lane & 7;
#define AVCER_DMA_SETUP()                                                                                           \
    unsigned a_off[A_ISS];                                                                                          \
    unsigned a_off2[A_ISS];                                                                                         \
    unsigned a_offk[A_ISS];                                                                                         \
    unsigned a_off2k[A_ISS];                                                                                        \
    int a_iy[A_ISS], a_ix[A_ISS], a_kc[A_ISS];                                                                      \
    _Pragma("unroll") for (int j = 0; j < A_ISS; ++j) {                                                             \
        const int lrow = dw * (A_ISS * 8) + j * 8 + lrow8;                                                          \
        const int m = m_base + lrow;                                                                                \
        const bool ok = m < p.M;                                                                                    \
        const int mm = ok ? m : 0;                                                                                  \
        const int ox = mm % p.OW;                                                                                   \
        const int t = mm / p.OW;                                                                                    \
        const int oy = t % p.OH;                                                                                    \
        const int b = t / p.OH;                                                                                     \
        const int iy = oy * p.sh - p.ph;                                                                            \
        const int ix = ox * p.sw - p.pw;                                                                            \
        a_iy[j] = ok ? iy : -(1 << 28);                                                                             \
        a_ix[j] = ix;                                                                                               \
        a_off[j] = (unsigned)(((long)b * p.sB + (long)iy * p.sH + (long)ix * p.sW + x_coff) * ES);                  \
        a_kc[j] = (slot ^ swz_key(lrow)) * VEC;                                                                     \
        a_offk[j] = a_off[j] + (unsigned)(a_kc[j] * ES);                                                            \
        a_off2[j] = ok ? (unsigned)(((long)b * p.sB2 + (long)oy * p.st2 * p.sH2 + (long)ox * p.st2 * p.sW2 + p.coff2) * ES) : OOB;\
        a_off2k[j] = ok ? a_off2[j] + (unsigned)(a_kc[j] * ES) : OOB;                                               \
        if (!ok) a_offk[j] = OOB;                                                                                   \
    }                                                                                                               \
    unsigned w_off[B_ISS];                                                                                          \
    _Pragma("unroll") for (int j = 0; j < B_ISS; ++j) {                                                             \
        const int lrow = dw * (B_ISS * 8) + j * 8 + lrow8;                                                          \
        w_off[j] = (unsigned)(((long)(n_base + lrow) * p.K + (slot ^ swz_key(lrow)) * VEC) * ES);                   \
    }                                                                                                               \
    int kc = 0, kx = 0, ky = 0;                                                                                     \
    int kdone = 0;                                                                                                  \
    unsigned wk = 0;

#define AVCER_ISSUE_TILES(buf)                                                                                      \
    do {                                                                                                            \
        char* sa_ = smem + (buf) * TILE_BYTES + dw * (A_ISS * 1024);                                                \
        char* sb_ = smem + (buf) * TILE_BYTES + BMT * ROWB + dw * (B_ISS * 1024);                                   \
        if (p.fast) {                                                                                               \
            /* no padding anywhere and every tap of every valid row in range: the per-lane offsets never change, */ \
            /* the K / tap advance is a scalar operand of the DMA instruction -> no vector arithmetic at all    */ \
            if (kdone >= p.K1) {                                                                                    \
                const unsigned so = (unsigned)((kdone - p.K1) * ES);                                                \
                _Pragma("unroll") for (int j = 0; j < A_ISS; ++j) dma16(x2rs, sa_ + j * 1024, a_off2k[j], so);      \
            } else {                                                                                                \
                const unsigned so = (unsigned)(((long)ky * p.dh * p.sH + (long)kx * p.dw * p.sW + kc) * ES);        \
                _Pragma("unroll") for (int j = 0; j < A_ISS; ++j) dma16(xrs, sa_ + j * 1024, a_offk[j], so);        \
            }                                                                                                       \
            _Pragma("unroll") for (int j = 0; j < B_ISS; ++j) dma16(wrs, sb_ + j * 1024, w_off[j], wk);             \
        } else if (kdone >= p.K1) {                                                                                 \
            _Pragma("unroll") for (int j = 0; j < A_ISS; ++j) {                                                     \
                const unsigned vo = a_off2[j] + (unsigned)((kdone - p.K1 + a_kc[j]) * ES);                          \
                dma16(x2rs, sa_ + j * 1024, a_off2[j] != OOB ? vo : OOB);                         \
            }                                                                                                       \
        } else if (tap_uniform) {                                                                                   \
            const int dy = ky * p.dh, dx = kx * p.dw;                                                               \
            const unsigned tap = (unsigned)(((long)dy * p.sH + (long)dx * p.sW + kc) * ES);                         \
            _Pragma("unroll") for (int j = 0; j < A_ISS; ++j) {                                                     \
                const int iy = a_iy[j] + dy, ix = a_ix[j] + dx;                                                     \
                const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.Wd);                   \
                dma16(xrs, sa_ + j * 1024, ok ? a_offk[j] + tap : OOB);                                             \
            }                                                                                                       \
        } else {                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < A_ISS; ++j) {                                                     \
                int kk = kc + a_kc[j], kxx = kx, kyy = ky;                                                          \
                if (kk >= p.Cin) { kk -= p.Cin; if (++kxx == p.KW) { kxx = 0; ++kyy; } }                            \
                const int dy = kyy * p.dh, dx = kxx * p.dw;                                                         \
                const int iy = a_iy[j] + dy, ix = a_ix[j] + dx;                                                     \
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;                      \
                const unsigned vo = a_off[j] + (unsigned)(((long)dy * p.sH + (long)dx * p.sW + kk) * ES);           \
                dma16(xrs, sa_ + j * 1024, ok ? vo : OOB);                                                          \
            }                                                                                                       \
        }                                                                                                           \
        kdone += BK;                                                                                                \
        if (!p.fast) {                                                                                              \
            _Pragma("unroll") for (int j = 0; j < B_ISS; ++j) dma16(wrs, sb_ + j * 1024, w_off[j] + wk);            \
        }                                                                                                           \
        if (p.tap_inner) {                                                                                          \
            if (++kx == p.KW) { kx = 0; if (++ky == p.KH) { ky = 0; kc += BK; } }                                   \
            wk = (unsigned)(((ky * p.KW + kx) * p.Cin + kc) * ES);                                                  \
        } else {                                                                                                    \
            wk += ROWB;                                                                                             \
            kc += BK;                                                                                               \
            while (kc >= p.Cin) {                                                                                   \
                kc -= p.Cin;                                                                                        \
                if (++kx == p.KW) { kx = 0; ++ky; }                                                                 \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)

    const int nk = p.K / BK;
    // Cin a multiple of the K-step: a K-step never straddles two filter taps, so the tap offset is one scalar per step
    // and the per-lane gather address is two adds and a bounds test instead of a per-lane (ky, kx, c) decomposition
    const bool tap_uniform = p.Cin % BK == 0;
    int cur = 0;

    // residual tile of this thread's epilogue rows: requested now, consumed after the last MFMA
    constexpr bool DIRECT = MODE >= 2;  // split-fp16 modes: permuted weight rows, epilogue straight from the accumulators
    constexpr int NFN = IS_F32 ? WN / 32 : WN / 16;
    constexpr int NFM = IS_F32 ? WM / 32 : WM / 16;
    using D = DrainMap<BMT, BN>;
    uint4 rres[DIRECT ? 1 : D::NP][2];
    uint4 rdir[DIRECT ? NFN / 2 : 1][DIRECT ? NFM : 1][2];
    // channel index of this wave's first output column: n_base counts rows of W / entries of scale and bias, the
    // epilogue helpers add p.yoff / p.roff themselves
    float wmul = 1.f;  // accumulator multiplier of the scaled split weights (trailer behind them: split_dev.h)
    if constexpr (DIRECT) wmul = split_wmul(p.W, p.w_bytes);
    if constexpr (DIRECT) res_prefetch_direct<OUT, NFN, NFM>(p, m_base + wm * WM, n_base + wn * WN, lane, rdir);
    else res_prefetch<OUT, BMT, BN>(p, m_base, n_base, tid, rres);

    static_assert(MODE < 2 || OUT != 1, "split-fp16 modes write f32 or sp32");
    using acc_t = typename std::conditional<IS_F32, f32x16_t, f32x4_t>::type;
    acc_t acc[NFN][NFM];
#pragma unroll
    for (int a = 0; a < NFN; ++a)
#pragma unroll
        for (int b = 0; b < NFM; ++b) acc[a][b] = acc_t{0};

    sp_flags_t ovm = 0;  // lanes that split a finite |x| >= 65520 into an fp16 pair (split_dev.h sp_commit)
    AVCER_DMA_SETUP();
    AVCER_ISSUE_TILES(0);
    __syncthreads();  // hipcc puts the s_waitcnt vmcnt(0) of the in-flight DMA in front of the barrier
    for (int step = 0; step < nk; ++step) {
        if (step + 1 < nk) AVCER_ISSUE_TILES(cur ^ 1);
        mfma_step<MODE, BMT, BN, TILE_BYTES>(p, smem, cur, acc, wm, wn, lane, ovm);
        // Keep every MFMA of this K-step in front of the barrier.  Left alone, hipcc hoists the s_waitcnt vmcnt(0) +
        // s_barrier above the second half of them, so the DMA issued at the top of the step gets half a step to land.
#pragma unroll
        for (int a = 0; a < NFN; ++a)
#pragma unroll
            for (int b = 0; b < NFM; ++b) asm volatile("" : "+v"(acc[a][b]));
        __syncthreads();
        cur ^= 1;
    }
#undef AVCER_ISSUE_TILES
#undef AVCER_DMA_SETUP

    if constexpr (DIRECT) {
        if (p.act == 3) epilogue_direct<OUT, 3, NFN, NFM>(p, acc, m_base + wm * WM, n_base + wn * WN, lane, rdir, wmul, ovm);
        else if (p.act == 2) epilogue_direct<OUT, 2, NFN, NFM>(p, acc, m_base + wm * WM, n_base + wn * WN, lane, rdir, wmul, ovm);
        else if (p.act == 1) epilogue_direct<OUT, 1, NFN, NFM>(p, acc, m_base + wm * WM, n_base + wn * WN, lane, rdir, wmul, ovm);
        else epilogue_direct<OUT, 0, NFN, NFM>(p, acc, m_base + wm * WM, n_base + wn * WN, lane, rdir, wmul, ovm);
        sp_commit(p.ovf, ovm);
    } else {
        // epilogue through LDS (the tile buffers are free: the loop ended on a barrier)
        stage_acc<MODE, BN>(p, smem, acc, n_base, wm * WM, wn, lane);
        __syncthreads();
        if (p.act == 3) drain_stage<OUT, BMT, BN, 3>(p, smem, m_base, n_base, tid, rres);
        else if (p.act == 2) drain_stage<OUT, BMT, BN, 2>(p, smem, m_base, n_base, tid, rres);
        else if (p.act == 1) drain_stage<OUT, BMT, BN, 1>(p, smem, m_base, n_base, tid, rres);
        else drain_stage<OUT, BMT, BN, 0>(p, smem, m_base, n_base, tid, rres);
    }
}

// ------------------------------------------------------------------------------------------------ weights direct (dtype 7 / 8)
// The split-fp16 contraction of sp32 activations in a second structure (round 3; tools/gemm_lab.hip is its test bench):
//   * the WEIGHT fragments never touch LDS.  The weights are stored once more in MFMA fragment order -- [N/16][K/32][hi, lo]
//     [64 lanes][16 B], rows permuted like every split weight -- so a wave gets a whole fragment with ONE coalesced 1 KiB
//     load straight into the registers the MFMA reads (inline asm: hipcc would otherwise drain the LDS-DMA queue for it);
//   * only the activation tile goes through LDS (16 KiB per K-step instead of 32): a FOUR-slot ring fits twice per CU
//     (64 KiB per block); the tile of step s+2 is requested in the second half of step s and stays in flight across the
//     step boundary;
//   * 128 x 256 tiles, the four waves side by side along n, each owning all 128 positions x 64 channels: nobody loads a
//     weight fragment twice and one LDS fragment read feeds 12 MFMAs (the 2 x 2 form: 6); the fragments of 16 positions are
//     read while the 12 MFMAs of the previous 16 run (two register pairs, ping-pong) -- across K-steps too: the step's one
//     barrier sits in the MIDDLE of its MFMA work, so the step boundary is seamless (no wait, no exposed LDS latency).
// Same product order per output element as conv_gemm_kernel<3, *, *>: bit-identical results (tests/test_gpu_gemm_wd.py).
// Vector-memory operations of a wave, in issue order: prologue A(0) W(0) A(1); step s: W(s+1) [8 loads, two in front of
// each of the first four tiles' MFMAs], A(s+2) [4 DMA pieces, in front of the last four tiles'].  Two counted waits per step:
// vmcnt(8) in front of the mid-step barrier (A(s+1), issued in the second half of the step before, has landed; this step's 8
// weight loads may still fly) and vmcnt(4) at the end (W(s+1) is in its registers; the 4 pieces may fly).  Past the end of
// K the same operations are issued on dummy targets (out-of-range DMA = zeros into a free slot, a repeated weight load into
// dead registers), so the count is exact.  The K position is a handful of scalars advanced by additions; GATHER selects the
// activation addressing at compile time: 0 = plain matrix (every Linear and 1x1 convolution: one tap, no padding), 1 = the
// same with a second source for the tail of K (conv3 + downsample), 2 = several taps without padding (Conv1d of the audio
// model), 3 = zero padding by per-row bounds tests (3x3 convolutions).
#define AVCER_WREGS4(H, L) "+v"(H[0]), "+v"(L[0]), "+v"(H[1]), "+v"(L[1]), "+v"(H[2]), "+v"(L[2]), "+v"(H[3]), "+v"(L[3])

template <int OUT, int ACT, int NFN, int NFM>
__device__ __forceinline__ void wd_epilogue(const GemmParams& p, f32x4_t (&acc)[NFN][NFM], int m_base, int c0, int lane, const float wmul) {
    sp_flags_t ovm = 0;  // range contract of the sp32 output (split_dev.h sp_commit)
    // straight from the accumulators (weight rows are permuted: lane group g holds channels 8g..8g+7 of every group of 32);
    // the residual is read here, four positions at a time, into the registers the weight fragments left free
#pragma unroll
    for (int j = 0; j < NFN / 2; ++j) {
        const int ch = c0 + 32 * j + 8 * (lane >> 4);
        float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (p.scale) { s0 = *reinterpret_cast<const float4*>(p.scale + ch); s1 = *reinterpret_cast<const float4*>(p.scale + ch + 4); }
        if (p.bias) { b0 = *reinterpret_cast<const float4*>(p.bias + ch); b1 = *reinterpret_cast<const float4*>(p.bias + ch + 4); }
        // the weights were split as w * 2^e (split_dev.h): wmul = 2^-e folds back into the channel scale, exactly
        s0 = make_float4(s0.x * wmul, s0.y * wmul, s0.z * wmul, s0.w * wmul);
        s1 = make_float4(s1.x * wmul, s1.y * wmul, s1.z * wmul, s1.w * wmul);
#pragma unroll
        for (int h = 0; h < NFM; h += 4) {
            uint4 rr[4][2];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                rr[f][0] = make_uint4(0u, 0u, 0u, 0u);
                rr[f][1] = make_uint4(0u, 0u, 0u, 0u);
                const long m = (long)m_base + (h + f) * 16 + (lane & 15);
                if (h + f < NFM && p.R && m < p.M) res_load<OUT>(p, m, ch, rr[f][0], rr[f][1]);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const long m = (long)m_base + (h + f) * 16 + (lane & 15);
                if (h + f >= NFM || m >= p.M) continue;
                finish8<OUT, ACT>(p, m, ch, scale_bias4(acc[2 * j][h + f], s0, b0), scale_bias4(acc[2 * j + 1][h + f], s1, b1), rr[f][0],
                                  rr[f][1], ovm);
            }
        }
    }
    if constexpr (OUT == 2) sp_commit(p.ovf, ovm);
}

template <int OUT, int GATHER, int NFM>
__global__ void __launch_bounds__(256, 2) conv_gemm_wd_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the asm statements below only parse for the device target
    // NFM = 8: 128 positions per tile.  NFM = 7: 112 -- the ring slots keep 128 rows, the DMA pieces of rows 112..127 are
    // out-of-range (zeros, no fetch) and the eighth fragment row is never read.  Which of the two leaves the smaller last
    // round on the 512 block slots is the launcher's choice (launch_wd); an element's K order does not depend on it.
    constexpr int BMT = 16 * NFM, BN = 256, NFN = 4, STAGES = 4, ABYTES = 128 * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[STAGES * ABYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int q = p.nwg >> 3, r = p.nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_n, tile_m;
    {
        const int per_group = p.gm * p.ntn;
        const int group = bid / per_group, within = bid - group * per_group;
        const int first_m = group * p.gm;
        const int gsize = min(p.gm, p.ntm - first_m);
        tile_m = first_m + within % gsize;
        tile_n = within / gsize;
    }
    const int m_base = tile_m * BMT, n_base = tile_n * BN;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X), (short)0, (int)p.x_bytes, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    i32x4_t wfrs;
    {
        const uint64_t a = (uint64_t)p.WF;
        wfrs[0] = (int)(a & 0xffffffffu);
        wfrs[1] = (int)((a >> 32) & 0xffffu);
        wfrs[2] = (int)p.w_bytes;
        wfrs[3] = 0x00020000;
    }
    const float wmul = split_wmul(p.WF, p.w_bytes);  // accumulator multiplier of the scaled split weights (split_dev.h)
    const int lrow8 = lane >> 3, slot = lane & 7, g = lane >> 4;
    const int nk = p.K >> 5;
    const int n1 = p.K1 >> 5;  // K-steps served by the first source
    unsigned a_offk[4];
    unsigned a_off2k[GATHER == 1 ? 4 : 1];
    int a_iy[GATHER == 3 ? 4 : 1], a_ix[GATHER == 3 ? 4 : 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = wave * 32 + j * 8 + lrow8;
        const int m = m_base + lrow;
        const bool ok = m < p.M && lrow < BMT;
        const int mm = ok ? m : 0;
        const int ox = mm % p.OW;
        const int t = mm / p.OW;
        const int oy = t % p.OH;
        const int b = t / p.OH;
        const int iy = oy * p.sh - p.ph, ix = ox * p.sw - p.pw;
        const unsigned kcb = (unsigned)((slot ^ swz_key(lrow)) << 4);  // this lane's 16-byte chunk of the 128-byte K-step
        a_offk[j] = ok ? (unsigned)(((long)b * p.sB + (long)iy * p.sH + (long)ix * p.sW + p.coff) * 4) + kcb : OOB;
        if constexpr (GATHER == 1)
            a_off2k[j] = ok ? (unsigned)(((long)b * p.sB2 + (long)oy * p.st2 * p.sH2 + (long)ox * p.st2 * p.sW2 + p.coff2) * 4) + kcb : OOB;
        if constexpr (GATHER == 3) {
            a_iy[j] = ok ? iy : -(1 << 28);
            a_ix[j] = ix;
        }
    }
    // byte offset of this lane's 16 bytes of the wave's first fragment (n tile, K-step 0, hi); the other three n tiles of the
    // wave lie whole multiples of wstride further on, which travels in the scalar offset of the load
    const unsigned wv0 = (unsigned)(((long)(n_base / 16 + wave * NFN) * nk) * 2048 + lane * 16);
    const unsigned wstride = (unsigned)nk * 2048u;
    f32x4_t acc[NFN][NFM];
#pragma unroll
    for (int a = 0; a < NFN; ++a)
#pragma unroll
        for (int b = 0; b < NFM; ++b) acc[a][b] = f32x4_t{0};
    u32x4_t wh0[NFN], wl0[NFN], wh1[NFN], wl1[NFN];
    // K position of the next A issue, all scalars: tap (ky, kx), channel byte offset kc4, byte offset a_so of that (tap,
    // chunk) relative to a row's first tap, byte offset wk of the K-step inside a weight row; w_saved lags one issue behind
    const int cin4 = p.Cin * 4;
    int kc4 = 0, kx = 0, ky = 0;
    unsigned a_so = 0, x2_so = 0, wk = 0, w_saved = 0;
    int islot = 0;  // ring slot of the next A issue

// the two loads (hi, lo) of n tile FN of the NEXT K-step's weight fragments
#define AVCER_WD_LOAD_W1(WH, WL, FN)                                                                                    \
    do {                                                                                                                \
        const unsigned so_ = w_saved * 16u + (unsigned)(FN) * wstride; /* K-step index * 2048 + n tile */               \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(WH[FN]) : "v"(wv0), "s"(wfrs), "s"(so_) : "memory"); \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(WL[FN]) : "v"(wv0), "s"(wfrs), "s"(so_) : "memory"); \
    } while (0)
#define AVCER_WD_LOAD_W(WH, WL)                                                                                         \
    do {                                                                                                                \
        _Pragma("unroll") for (int fn = 0; fn < NFN; ++fn) AVCER_WD_LOAD_W1(WH, WL, fn);                                 \
    } while (0)
// One 1 KiB piece (8 rows) J of the activation tile of K-step T, into ring slot islot.  ONE DMA instruction whatever T is
// (no branch: tools/audit_asm_loads.py counts the vector-memory operations of every path through the loop): past the end of
// K (T >= nk) and for rows outside the image the per-lane offset is out of range -- the hardware writes zeros without
// fetching --, and the second source of GATHER 1 is a wave-uniform choice of descriptor and scalar offset.
#define AVCER_WD_ISSUE_A1(T, J)                                                                                         \
    do {                                                                                                                \
        char* sa_ = smem + islot * ABYTES + wave * 4096 + (J) * 1024;                                                   \
        const bool live_ = (T) < nk;                                                                                    \
        if constexpr (GATHER == 1) {                                                                                    \
            const bool second_ = (T) >= n1; /* tail of K: the second source */                                          \
            const auto rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(second_ ? p.X2 : p.X), (short)0,       \
                                                               (int)(second_ ? p.x2_bytes : p.x_bytes), 0x00020000);    \
            const unsigned vo_ = second_ ? a_off2k[GATHER == 1 ? (J) : 0] : a_offk[J];                                  \
            dma16(rs_, sa_, live_ ? vo_ : OOB, second_ ? x2_so : a_so);                                                 \
        } else if constexpr (GATHER == 3) {                                                                             \
            const int iy = a_iy[GATHER == 3 ? (J) : 0] + ky * p.dh, ix = a_ix[GATHER == 3 ? (J) : 0] + kx * p.dw;       \
            const bool ok = live_ & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.Wd);                   \
            /* the tap offset goes into the per-lane offset: the hardware bounds test ignores the scalar offset, and */ \
            /* a border row's base offset is negative (wrapped) until its tap is added                              */ \
            dma16(xrs, sa_, ok ? a_offk[J] + a_so : OOB);                                                               \
        } else {                                                                                                        \
            dma16(xrs, sa_, live_ ? a_offk[J] : OOB, a_so);                                                             \
        }                                                                                                               \
    } while (0)
// behind the last piece of K-step T: the ring slot and the K position (a handful of scalars) move on
#define AVCER_WD_ADVANCE_A(T)                                                                                           \
    do {                                                                                                                \
        islot = islot == STAGES - 1 ? 0 : islot + 1;                                                                    \
        if ((T) < nk) {                                                                                                 \
            w_saved = wk;                                                                                               \
            if (GATHER == 1 && (T) >= n1) {                                                                             \
                x2_so += ROWB;                                                                                          \
                wk += ROWB;                                                                                             \
            } else if constexpr (GATHER <= 1) { /* plain matrix: one tap, channels in order */                          \
                a_so += ROWB;                                                                                           \
                wk += ROWB;                                                                                             \
            } else if (p.tap_inner) { /* (channel chunk, ky, kx) order: the next tap of the same chunk */               \
                a_so += p.tapW4;                                                                                        \
                wk += cin4;                                                                                             \
                if (++kx == p.KW) {                                                                                     \
                    kx = 0;                                                                                             \
                    a_so += p.tapH4 - p.KW * p.tapW4;                                                                   \
                    if (++ky == p.KH) { ky = 0; a_so += ROWB - p.KH * p.tapH4; kc4 += ROWB; wk = (unsigned)kc4; }       \
                }                                                                                                       \
            } else { /* (ky, kx, channel chunk) order */                                                                \
                a_so += ROWB;                                                                                           \
                wk += ROWB;                                                                                             \
                kc4 += ROWB;                                                                                            \
                if (kc4 == cin4) {                                                                                      \
                    kc4 = 0;                                                                                            \
                    a_so += p.tapW4 - cin4;                                                                             \
                    if (++kx == p.KW) { kx = 0; a_so += p.tapH4 - p.KW * p.tapW4; ++ky; }                               \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#define AVCER_WD_ISSUE_A(T)                                                                                             \
    do {                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) AVCER_WD_ISSUE_A1(T, j);                                           \
        AVCER_WD_ADVANCE_A(T);                                                                                          \
    } while (0)
#define AVCER_WD_WAIT(N, H, L) asm volatile("s_waitcnt vmcnt(" #N ")" : AVCER_WREGS4(H, L)::"memory")
#define AVCER_WD_READ(BASE, R, AH, AL)                                                                                  \
    do {                                                                                                                \
        const int row = (R) * 16 + (lane & 15);                                                                         \
        AH = *reinterpret_cast<const spx8_t*>((BASE) + swz(row, g));                                                  \
        AL = *reinterpret_cast<const spx8_t*>((BASE) + swz(row, 4 + g));                                              \
    } while (0)
#define AVCER_WD_MFMA(R, AH, AL, WH, WL)                                                                                \
    _Pragma("unroll") for (int fn = 0; fn < NFN; ++fn) {                                                                 \
        const spx8_t whi = __builtin_bit_cast(spx8_t, WH[fn]), wlo = __builtin_bit_cast(spx8_t, WL[fn]);          \
        f32x4_t& c_ = acc[fn][R];                                                                                       \
        c_ = mfma_sp(wlo, AH, c_);                                             \
        c_ = mfma_sp(whi, AL, c_);                                             \
        c_ = mfma_sp(whi, AH, c_);                                             \
    }
// One K-step.  Its twelve vector-memory operations are SPREAD over the step instead of leading it (round 4): the two
// loads of one n tile of W(S+1) in front of the MFMAs of each of the tiles 0-3, one piece of A(S+2) in front of those of
// each of the last four tiles.  As a burst at the top of the step (96 operations per CU within a few hundred cycles) they
// backed up the CU's one vector-memory pipe and the waves sat in ISSUE behind it with their MFMAs unissued: -2 ... -10 % time
// per layer against the burst on the same box (profiles/experiments/r04_wd_ablate_*.txt).  Counted waits: vmcnt(8) in front
// of the mid-step barrier (A(S+1), issued in the second half of step S-1, has landed; the 8 weight loads of this step may
// fly), vmcnt(4) at the end (W(S+1) is in its registers, this step's four pieces may fly).
#define AVCER_WD_STEP(S, PH, WH, WL, WHN, WLN)                                                                          \
    do {                                                                                                                \
        const char* sa = smem + rslot * ABYTES;                                                                         \
        rslot = rslot == STAGES - 1 ? 0 : rslot + 1;                                                                    \
        const char* sa_next = smem + rslot * ABYTES;                                                                    \
        /* one 16-position tile at a time: the two fragment reads of tile t+1 are issued in front of the 12 MFMAs of */ \
        /* tile t (two register pairs, ping-pong), and the last tile's partner is tile 0 of the NEXT step: the loop   */ \
        /* has no seam (PH = which pair holds tile 0 of this step; with an odd tile count it alternates by step).     */ \
        /* The scheduling fences pin that order: left alone, hipcc either hoists all sixteen reads (no registers left */ \
        /* for them) or sinks each pair behind the MFMAs it should cover.                                             */ \
        _Pragma("unroll") for (int t = 0; t < NFM; ++t) {                                                                \
            const int cur_ = ((PH) + t) & 1, nxt_ = cur_ ^ 1;                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            if (t + 1 < NFM) AVCER_WD_READ(sa, t + 1, ah[nxt_], al[nxt_]);                                              \
            else AVCER_WD_READ(sa_next, 0, ah[nxt_], al[nxt_]);                                                         \
            asm volatile("" ::: "memory");                                                                              \
            if (t < NFN) {                                                                                              \
                AVCER_WD_LOAD_W1(WHN, WLN, t);                                                                          \
            } else { /* pieces 0..3 of A(S+2) at tiles 4 .. 7 (NFM = 7: tiles 4, 5 and, two pieces, 6) */               \
                if (NFM == 8 || t < 6) { AVCER_WD_ISSUE_A1((S) + 2, t - NFN); }                                         \
                else { AVCER_WD_ISSUE_A1((S) + 2, 2); AVCER_WD_ISSUE_A1((S) + 2, 3); }                                  \
                if (t == NFM - 1) AVCER_WD_ADVANCE_A((S) + 2);                                                          \
            }                                                                                                           \
            asm volatile("" ::: "memory");                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            AVCER_WD_MFMA(t, ah[cur_], al[cur_], WH, WL);                                                               \
            if (t == 3) {                                                                                               \
                /* the step's one barrier, in the MIDDLE of its MFMA work: behind it the tile of step S+1 (issued in  */ \
                /* the second half of the previous step) is complete for every wave, and every wave has finished      */ \
                /* reading the slot of step S-1, which the pieces issued behind this barrier overwrite (four slots:   */ \
                /* the one being read, the one that just landed, the one about to be filled, the one just freed)      */ \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                        \
                __builtin_amdgcn_s_barrier();                                                                           \
                asm volatile("" ::: "memory");                                                                          \
            }                                                                                                           \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        AVCER_WD_WAIT(4, WHN, WLN); /* the weight fragments of step S+1 are in their registers */                       \
    } while (0)

    int rslot = 0;  // ring slot the current step reads
    spx8_t ah[2], al[2];
    AVCER_WD_ISSUE_A(0);
    asm volatile("" ::: "memory");
    AVCER_WD_LOAD_W(wh0, wl0);
    asm volatile("" ::: "memory");
    AVCER_WD_ISSUE_A(1);
    asm volatile("" ::: "memory");
    AVCER_WD_WAIT(4, wh0, wl0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    AVCER_WD_READ(smem, 0, ah[0], al[0]);
    for (int s = 0; s < nk; s += 2) {  // nk is even (checked by the launcher): after two steps tile 0 is back in pair 0
        AVCER_WD_STEP(s, 0, wh0, wl0, wh1, wl1);
        AVCER_WD_STEP(s + 1, NFM & 1, wh1, wl1, wh0, wl0);
    }
    asm volatile("" : "+v"(ah[0]), "+v"(al[0]));  // the last step's look-ahead read (a free slot): consumed by nobody
    // the dummy operations of the last steps are still in flight: drain them before the registers are reused
    AVCER_WD_WAIT(0, wh0, wl0);
    AVCER_WD_WAIT(0, wh1, wl1);
#undef AVCER_WD_STEP
#undef AVCER_WD_MFMA
#undef AVCER_WD_READ
#undef AVCER_WD_WAIT
#undef AVCER_WD_ISSUE_A
#undef AVCER_WD_ISSUE_A1
#undef AVCER_WD_ADVANCE_A
#undef AVCER_WD_LOAD_W1
#undef AVCER_WD_LOAD_W
    const int c0 = n_base + wave * (BN / 4);
    if (p.act == 3) wd_epilogue<OUT, 3, NFN, NFM>(p, acc, m_base, c0, lane, wmul);
    else if (p.act == 2) wd_epilogue<OUT, 2, NFN, NFM>(p, acc, m_base, c0, lane, wmul);
    else if (p.act == 1) wd_epilogue<OUT, 1, NFN, NFM>(p, acc, m_base, c0, lane, wmul);
    else wd_epilogue<OUT, 0, NFN, NFM>(p, acc, m_base, c0, lane, wmul);
#endif
}


// ------------------------------------------------------------------------------------------------ skinny form (dtype 9 / 10)
// The same contraction for a HANDFUL of positions (M <= 256: one frame through the drop-in mirrors).
// There a launch of the tiled forms is one or two blocks walking K in 72-144 dependent steps of 0.66 us each -- every step waits
// for its own LDS-DMA round trip -- and the one-frame static CNN is 65 such launches: 1.55 ms (profiles/r04_per_call_latency.json).
// Here a block is ONE wave owning 16 NFM positions x 16 output channels (one weight fragment tile; grid N / 16 x M / (16 NFM),
// so that hundreds of CUs stream weights instead of one or two), nothing goes through LDS and nothing is synchronised: the
// activation fragments (an sp32 row IS the fragment layout) and the fragment-order weights are loaded straight into a
// register ring D K-steps deep (one wave per SIMD has the whole register file: 48 KiB in flight per wave), and everything the
// epilogue needs from memory (scale, bias, residual) is requested before the K walk.  With so few waves the launch is paced by
// memory latency x bytes per wave, so the launcher picks the SMALLEST tile that still fits the chip in one round.  Same
// products in the same order per output element as the tiled forms (the K walk follows p.tap_inner), the same epilogue
// arithmetic: bit-identical (tests/test_gpu_gemm_wd.py), so results do not depend on which form a batch size selects.
// Byte offset of the residual piece of output row m, channels n0 .. n0 + 3 (res_load's addressing)
template <int OUT>
__device__ __forceinline__ unsigned res_off4(const GemmParams& p, long m, int n0) {
    if (p.rsub > 1) {
        const int ohw = p.OH * p.OW;
        const int b = (int)(m / ohw), rem = (int)m - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        m = ((long)b * p.rH + (long)oy * p.rsub) * p.rW + (long)ox * p.rsub;
    }
    const long e = m * p.ldR + p.roff + n0;
    return (unsigned)(OUT == 0 ? e * 4 : sp32_byte(e));
}

// finish8 for the 4 channels one lane of ONE fragment tile holds (element for element the same arithmetic)
template <int OUT, int ACT>
__device__ __forceinline__ void finish4(const GemmParams& p, long m, int n0, const float4 a, const uint2 r0, const uint2 r1,
                                        sp_flags_t& ovm) {
#pragma clang fp contract(off)
    static_assert(OUT == 0 || OUT == 2, "f32 or sp32 output");
    float v[4] = {a.x, a.y, a.z, a.w};
    float r[4];
    if constexpr (OUT == 0) {
        r[0] = __builtin_bit_cast(float, r0.x); r[1] = __builtin_bit_cast(float, r0.y);
        r[2] = __builtin_bit_cast(float, r1.x); r[3] = __builtin_bit_cast(float, r1.y);
    } else {
        const uint32_t wh[2] = {r0.x, r0.y}, wl[2] = {r1.x, r1.y};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            r[2 * j] = sp2f((uint16_t)(wh[j] & 0xffff)) + sp2f((uint16_t)(wl[j] & 0xffff));
            r[2 * j + 1] = sp2f((uint16_t)(wh[j] >> 16)) + sp2f((uint16_t)(wl[j] >> 16));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = p.res_after ? v[j] : v[j] + r[j];
        if constexpr (ACT == 1) x = relu_nan(x);
        if constexpr (ACT == 2) x = gelu_erf(x);
        if constexpr (ACT == 3) x = gelu_fast(x);
        v[j] = p.res_after ? x + r[j] : x;
    }
    const long e = m * p.ldY + p.yoff + n0;
    if constexpr (OUT == 0) {
        *reinterpret_cast<float4*>(p.Y + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint32_t h[2], l[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sp_value(v[j]);
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            amax = sp_max2(amax, v[2 * j], v[2 * j + 1]);
            const uint16_t h0 = f2sp(v[2 * j]), h1 = f2sp(v[2 * j + 1]);
            h[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            l[j] = (uint32_t)f2sp(v[2 * j] - sp2f(h0)) | ((uint32_t)f2sp(v[2 * j + 1] - sp2f(h1)) << 16);
        }
        sp_flag(ovm, amax);
        char* yp = p.Y + sp32_byte(e);
        *reinterpret_cast<uint2*>(yp) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(yp + 64) = make_uint2(l[0], l[1]);
    }
}

template <int OUT, int GATHER, int NFM, int D>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) conv_gemm_skinny_kernel(const GemmParams p) {
    const int lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    // grouped convolution: the fragment tiles of all groups side by side (weight rows are stacked [groups * N][K]); group grp
    // reads its own Cin channels of the input pixel and owns channels grp * N .. + N of the output
    const int nt = blockIdx.x, m_base = blockIdx.y * (16 * NFM);
    const int grp = p.groups > 1 ? nt / (p.N >> 4) : 0;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X), (short)0, (int)p.x_bytes, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.WF), (short)0, (int)p.w_bytes, 0x00020000);
    const int nk = p.K >> 5, nq = p.Cin >> 5;
    // stored row 16 t + 4 g + r of a 32-channel group is channel 8 g + 4 t + r (split_weight_rows_kernel): this lane's four
    const int ch = (nt >> 1) * 32 + 8 * g + 4 * (nt & 1);
    // everything the epilogue reads, requested now: its latency hides behind the K walk.  Buffer loads, not branches: an absent
    // operand is a descriptor of zero bytes (a load inside `if (p.scale)` is followed by its own s_waitcnt vmcnt(0))
    const float wmul = split_wmul(p.WF, p.w_bytes);
    const int n_all = p.N * (p.groups > 1 ? p.groups : 1);
    const auto srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale), (short)0, p.scale ? n_all * 4 : 0, 0x00020000);
    const auto brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), (short)0, p.bias ? n_all * 4 : 0, 0x00020000);
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.R), (short)0, p.R ? 0x7ffffff0 : 0, 0x00020000);
    const u32x4_t scv = __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)ch * 4u, 0, 0);
    const u32x4_t biv = __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)ch * 4u, 0, 0);
    u32x2_t rr[NFM][2];  // OUT 0: four f32; OUT 2: four hi, four lo
#pragma unroll
    for (int fm = 0; fm < NFM; ++fm) {
        const long m = (long)m_base + fm * 16 + l15;
        const unsigned ro = (p.R && m < p.M) ? res_off4<OUT>(p, m, ch) : OOB;
        rr[fm][0] = __builtin_amdgcn_raw_buffer_load_b64(rrs, ro, 0, 0);
        rr[fm][1] = __builtin_amdgcn_raw_buffer_load_b64(rrs, ro + (OUT == 0 ? 8u : 64u), 0, 0);
    }
    unsigned a_off[NFM];
    int a_iy[GATHER == 3 ? NFM : 1], a_ix[GATHER == 3 ? NFM : 1];
#pragma unroll
    for (int fm = 0; fm < NFM; ++fm) {
        const int m = m_base + fm * 16 + l15;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int ox = mm % p.OW, t = mm / p.OW;
        const int oy = t % p.OH, b = t / p.OH;
        const int iy = oy * p.sh - p.ph, ix = ox * p.sw - p.pw;
        a_off[fm] = ok ? (unsigned)(((long)b * p.sB + (long)iy * p.sH + (long)ix * p.sW + p.coff + grp * p.Cin) * 4) + 16u * g : OOB;
        if constexpr (GATHER == 3) {
            a_iy[fm] = ok ? iy : -(1 << 28);
            a_ix[fm] = ix;
        }
    }
    // weight fragments: [(n / 16)][nk][hi, lo][64 lanes][16 B]
    const unsigned w_v = (unsigned)lane * 16u;
    const unsigned w_t = (unsigned)nt * (unsigned)nk * 2048u;
    // K position of the next issue: tap (ky, kx) and channel chunk kq, all scalars (the walk of launch_conv_gemm's tap_inner
    // flag); `left` K-steps remain to be requested.  Past the end the ring is fed from out-of-range offsets (zeros, no memory
    // traffic), so every slot is refilled unconditionally and the counted waits stay static.
    // A macro, not a lambda: captured by reference the counters went to scratch memory (every issue then waited for them)
    int kq = 0, kx = 0, ky = 0, left = nk;
    u32x4_t ra[D][NFM][2], rw[D][2];
#define AVCER_SK_ISSUE(A, W)                                                                                            \
    do {                                                                                                                \
        const bool live = left > 0;                                                                                     \
        --left;                                                                                                         \
        const unsigned a_so = (unsigned)(ky * p.tapH4 + kx * p.tapW4 + kq * ROWB);                                      \
        const unsigned ks = (unsigned)((ky * p.KW + kx) * nq + kq) * 2048u; /* K-step of the [N][kh][kw][Cin] rows */    \
        _Pragma("unroll") for (int fm = 0; fm < NFM; ++fm) {                                                            \
            unsigned vo = live ? a_off[fm] : OOB, so = a_so;                                                            \
            if constexpr (GATHER == 3) { /* padded: the bounds test ignores the scalar offset, and a border row's base */ \
                                         /* offset is negative (wrapped) until its tap is added                       */ \
                const int iy = a_iy[GATHER == 3 ? fm : 0] + ky * p.dh, ix = a_ix[GATHER == 3 ? fm : 0] + kx * p.dw;     \
                const bool ok = live & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.Wd);                \
                vo = ok ? a_off[fm] + a_so : OOB;                                                                       \
                so = 0u;                                                                                                \
            }                                                                                                           \
            A[fm][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, so, 0);                                           \
            A[fm][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vo + 64u, so, 0);                                     \
        }                                                                                                               \
        W[0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, live ? w_v : OOB, w_t + ks, 0);                               \
        W[1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, live ? w_v + 1024u : OOB, w_t + ks, 0);                       \
        if (p.tap_inner) { /* (channel chunk, ky, kx) */                                                                \
            if (++kx == p.KW) { kx = 0; if (++ky == p.KH) { ky = 0; ++kq; } }                                           \
        } else { /* (ky, kx, channel chunk) */                                                                          \
            if (++kq == nq) { kq = 0; if (++kx == p.KW) { kx = 0; ++ky; } }                                             \
        }                                                                                                               \
    } while (0)
    f32x4_t acc[NFM];
#pragma unroll
    for (int b = 0; b < NFM; ++b) acc[b] = f32x4_t{0};
#pragma unroll
    for (int j = 0; j < D; ++j) AVCER_SK_ISSUE(ra[j], rw[j]);
    for (int s = 0; s < nk; s += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + j < nk) {
                const spx8_t whi = __builtin_bit_cast(spx8_t, rw[j][0]), wlo = __builtin_bit_cast(spx8_t, rw[j][1]);
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm) {
                    const spx8_t ahi = __builtin_bit_cast(spx8_t, ra[j][fm][0]), alo = __builtin_bit_cast(spx8_t, ra[j][fm][1]);
                    acc[fm] = mfma_sp(wlo, ahi, acc[fm]);
                    acc[fm] = mfma_sp(whi, alo, acc[fm]);
                    acc[fm] = mfma_sp(whi, ahi, acc[fm]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            AVCER_SK_ISSUE(ra[j], rw[j]);  // the slot just consumed: step s + j + D
        }
    }
#undef AVCER_SK_ISSUE
    // epilogue: the direct epilogue of the tiled forms for one fragment tile
    sp_flags_t ovm = 0;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.scale) sc = __builtin_bit_cast(float4, scv);
    const float4 bi = __builtin_bit_cast(float4, biv);  // zeros without a bias
    sc = make_float4(sc.x * wmul, sc.y * wmul, sc.z * wmul, sc.w * wmul);
    auto fin = [&](auto act) {
        constexpr int ACT = decltype(act)::value;
#pragma unroll
        for (int fm = 0; fm < NFM; ++fm) {
            const long m = (long)m_base + fm * 16 + l15;
            if (m < p.M) finish4<OUT, ACT>(p, m, ch, scale_bias4(acc[fm], sc, bi), __builtin_bit_cast(uint2, rr[fm][0]),
                                         __builtin_bit_cast(uint2, rr[fm][1]), ovm);
        }
    };
    if (p.act == 3) fin(std::integral_constant<int, 3>{});
    else if (p.act == 2) fin(std::integral_constant<int, 2>{});
    else if (p.act == 1) fin(std::integral_constant<int, 1>{});
    else fin(std::integral_constant<int, 0>{});
    if constexpr (OUT == 2) sp_commit(p.ovf, ovm);
}

template <int OUT, int NFM, int D>
void launch_skinny_t(const GemmParams& p, hipStream_t st) {
    const dim3 grid(p.N / 16 * (p.groups > 1 ? p.groups : 1), (p.M + 16 * NFM - 1) / (16 * NFM));
    if (p.fast && p.KH * p.KW == 1) conv_gemm_skinny_kernel<OUT, 0, NFM, D><<<grid, dim3(64), 0, st>>>(p);
    else if (p.fast) conv_gemm_skinny_kernel<OUT, 2, NFM, D><<<grid, dim3(64), 0, st>>>(p);
    else conv_gemm_skinny_kernel<OUT, 3, NFM, D><<<grid, dim3(64), 0, st>>>(p);
}

// The smallest tile whose grid still fits the chip in one round of one wave per SIMD (bit-identical results either way)
template <int OUT>
void launch_skinny(const GemmParams& p, hipStream_t st) {
    const long simds = (long)p.slots * 2;  // block slots = 2 per CU, 4 SIMDs per CU
    // 16-position tiles while they leave half the SIMDs free (a wave is paced by its round trips -- the weights come from beyond
    // the L2 once per launch -- so the smallest tile wins); past that the round trips lengthen with the load, and the 32-position
    // tile moves 6 KiB per K-step where two 16s move 8 (tools/ab_layers.py --skinny, profiles/r05_skinny_ab.txt)
    const long nt = p.N / 16 * (p.groups > 1 ? p.groups : 1);
    if (p.tile_m == 16 || (p.tile_m == 0 && (long)((p.M + 15) / 16) * nt <= simds / 2)) launch_skinny_t<OUT, 1, 12>(p, st);
    else if (p.tile_m == 32 || (p.tile_m == 0 && (long)((p.M + 31) / 32) * nt <= simds)) launch_skinny_t<OUT, 2, 8>(p, st);
    else launch_skinny_t<OUT, 4, 4>(p, st);
}

template <int OUT, int NFM>
void launch_wd_t(GemmParams& p, hipStream_t st) {
    p.ntm = (p.M + 16 * NFM - 1) / (16 * NFM);
    p.gm = 8;
    p.ntn = p.N / 256;
    p.nwg = p.ntm * p.ntn;
    if (p.X2) conv_gemm_wd_kernel<OUT, 1, NFM><<<dim3(p.nwg), dim3(256), 0, st>>>(p);
    else if (p.fast && p.KH * p.KW == 1) conv_gemm_wd_kernel<OUT, 0, NFM><<<dim3(p.nwg), dim3(256), 0, st>>>(p);
    else if (p.fast) conv_gemm_wd_kernel<OUT, 2, NFM><<<dim3(p.nwg), dim3(256), 0, st>>>(p);
    else conv_gemm_wd_kernel<OUT, 3, NFM><<<dim3(p.nwg), dim3(256), 0, st>>>(p);
}

// 128 or 112 positions per tile: whichever grid costs fewer (rounds x rows per tile); avcer_conv_desc.tile_m overrides.
template <int OUT>
void launch_wd(const GemmParams& p0, hipStream_t st) {
    GemmParams p = p0;
    const long ntn = p.N / 256;
    const double c128 = grid_rounds((p.M + 127L) / 128 * ntn, p.slots) * 128, c112 = grid_rounds((p.M + 111L) / 112 * ntn, p.slots) * 112;
    const bool m112 = p.tile_m == 112 || (p.tile_m == 0 && c112 < 0.97 * c128);
    if (m112) launch_wd_t<OUT, 7>(p, st);
    else launch_wd_t<OUT, 8>(p, st);
}

// Tile width by shape.  K <= 128: bandwidth-bound 1x1 convolutions, the 48 KiB BN = 64 tile lets three blocks share a CU.
inline bool choose_bn128(const GemmParams& p) { return p.K > 128; }

template <int MODE, int OUT>
void launch_t(const GemmParams& p0, hipStream_t st) {
    GemmParams p = p0;
    bool bn128 = p.N % 128 == 0 && choose_bn128(p);
    // a grid of at most one block per two CUs (the LSTM's recurrent GEMMs, fc1, single-frame calls): twice the blocks at half the width
    if (bn128 && (long)((p.M + 127) / 128) * (p.N / 128) <= p.slots / 4) bn128 = false;
    if (p.tile_n == 64) bn128 = false;
    if (p.tile_n == 128 && p.N % 128 == 0) bn128 = true;
    p.ntm = (p.M + 127) / 128;
    p.gm = 8;  // grouped block order: 8 m-tiles x all n-tiles per group (0/4/8/16 measured within +-3 %)
    p.ntn = p.N / (bn128 ? 128 : 64);
    p.nwg = p.ntm * p.ntn;
    if (bn128) conv_gemm_kernel<MODE, OUT, 128><<<dim3(p.nwg, p.groups), dim3(256), 0, st>>>(p);
    else conv_gemm_kernel<MODE, OUT, 64><<<dim3(p.nwg, p.groups), dim3(256), 0, st>>>(p);
}

}  // namespace

// Live timing of the MFMA kernels (avcer_profile_enable): a pair of events around each launch, on the launch stream.
// Records *ev0 now; the caller records *ev1 behind its launch.  Both stay null while profiling is off.
int prof_begin(avcer_ctx* ctx, hipStream_t st, hipEvent_t* ev0, hipEvent_t* ev1, int family, double flops, double bytes, long M, long N,
               long K) {
    *ev0 = *ev1 = nullptr;
    ctx->fam_launches[family] += 1;
    ctx->fam_flops[family] += flops;
    ctx->fam_bytes[family] += bytes;
    if (!ctx->prof) return AVCER_OK;
    if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
        for (int i = 0; i < 512; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return set_err(ctx, AVCER_EHIP, "hipEventCreate failed");
            ctx->prof_ev.push_back(e);
        }
    }
    if (ctx->prof_fam.size() < ctx->prof_ev.size() / 2) {
        ctx->prof_fam.resize(ctx->prof_ev.size() / 2, 0);
        ctx->prof_log.resize(ctx->prof_ev.size() / 2);
    }
    ctx->prof_fam[ctx->prof_used / 2] = family;
    ctx->prof_log[ctx->prof_used / 2] = {flops, bytes, M, N, K};
    *ev0 = ctx->prof_ev[ctx->prof_used++];
    *ev1 = ctx->prof_ev[ctx->prof_used++];
    (void)hipEventRecord(*ev0, st);
    return AVCER_OK;
}

int launch_conv_gemm(avcer_ctx* ctx, const avcer_conv_desc& d, int dtype, const void* x, const void* w,
                     const float* scale, const float* bias, const void* residual, void* y, hipStream_t st,
                     const void* x2) {
    if (dtype < 0 || dtype > 10) return set_err(ctx, AVCER_EINVAL, "conv_gemm: dtype %d", dtype);
    const bool skinny = dtype >= 9;   // 9 / 10: dtype 7 / 8 for a handful of positions (conv_gemm_skinny_kernel)
    const int es = (dtype == 1 || dtype == 2) ? 2 : 4;
    const bool wdirect = dtype >= 7;  // 7 / 8 (and 9 / 10): dtype 5 / 6 with the weights in fragment order (conv_gemm_wd_kernel)
    const bool a_split = dtype == 5 || dtype == 6 || wdirect, o_split = dtype == 4 || dtype == 5 || dtype == 7 || dtype == 9;
    const int vec = 16 / es;
    const int bk = ROWB / es;  // 32 elements (f32, split-fp16) or 64 (bf16) per K-step
    const long M = (long)d.batch * d.out_h * d.out_w;
    const long K1 = (long)d.kh * d.kw * d.cin;
    const long K = K1 + (x2 ? d.x2_cin : 0);
    if (x2 && (d.kh != 1 || d.kw != 1 || d.pad_h || d.pad_w || d.groups > 1 || d.x2_cin <= 0 || K1 % bk || d.x2_cin % bk ||
               d.x2_stride < 1 || d.x2_coff % vec || d.x2_stride_b % vec || d.x2_stride_h % vec || d.x2_stride_w % vec))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: a second A source needs two 1x1 convolutions with K multiples of %d", bk);
    if (M <= 0 || M > 0x7fffff00L) return set_err(ctx, AVCER_EINVAL, "conv_gemm: M=%ld out of range", M);
    if (d.n <= 0 || d.n % 64) return set_err(ctx, AVCER_EINVAL, "conv_gemm: N=%d must be a multiple of 64", d.n);
    if (K % bk) return set_err(ctx, AVCER_EINVAL, "conv_gemm: K=%ld must be a multiple of %d", K, bk);
    if (d.cin % vec || d.x_coff % vec || d.x_stride_b % vec || d.x_stride_h % vec)
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: cin/coff/strides must be multiples of %d", vec);
    if (d.x_stride_w % vec && !(d.kw == 1 && d.pad_w == 0 && (d.stride_w * d.x_stride_w) % vec == 0))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: x_stride_w=%ld breaks 16-byte alignment", (long)d.x_stride_w);
    const int ovec = o_split ? 32 : 16 / ((dtype == 1) ? 2 : 4);  // 16-byte vectors / whole sp32 groups
    if (d.y_ld % ovec || d.y_coff % ovec || (residual && (d.r_ld % ovec || d.r_coff % ovec)))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: output/residual leading dims and offsets must be multiples of %d", ovec);
    if (a_split && x2 && (d.x2_cin % 32 || d.x2_coff % 32 || d.x2_stride_b % 32 || d.x2_stride_h % 32 || d.x2_stride_w % 32))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: sp32 second source needs cin/coff/strides in whole groups of 32");
    if (a_split && (d.cin % 32 || d.x_coff % 32 || d.x_stride_b % 32 || d.x_stride_h % 32 || d.x_stride_w % 32))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: sp32 activations need cin/coff/strides in whole groups of 32");
    if (!x || !w || !y) return set_err(ctx, AVCER_EINVAL, "conv_gemm: null pointer");
    if (d.act < 0 || d.act > 3) return set_err(ctx, AVCER_EINVAL, "conv_gemm: act %d (0 none, 1 relu, 2 gelu, 3 gelu with the short erf)", d.act);
    GemmParams p;
    p.X = (const char*)x; p.W = (const char*)w; p.scale = scale; p.bias = bias; p.R = (const char*)residual;
    p.Y = (char*)y;
    p.M = (int)M; p.N = d.n; p.K = (int)K;
    p.OH = d.out_h; p.OW = d.out_w; p.H = d.in_h; p.Wd = d.in_w; p.Cin = d.cin; p.KW = d.kw; p.KH = d.kh;
    // Multi-tap convolutions whose Cin is a whole number of K-steps walk K as (channel chunk, ky, kx): the taps of one
    // chunk re-read the same or neighbouring pixels in CONSECUTIVE K-steps, while they are still in the XCD's 4 MiB L2.
    // In (ky, kx, chunk) order a pixel comes back only after Cin/BK steps of every block of the XCD, and the 3x3 layers of
    // stages 3-4 fetched 5.8-13 x their compulsory bytes through the fabric (per-dispatch FETCH_SIZE, round 2).  The
    // weight rows are addressed by the same (ky, kx, chunk) offset, so the [N][kh][kw][Cin] layout is unchanged; only the
    // order of the f32 accumulation differs.
    p.tap_inner = !x2 && d.kh * d.kw > 1 && d.cin % bk == 0 && d.cin > bk;
    p.sh = d.stride_h; p.sw = d.stride_w; p.ph = d.pad_h; p.pw = d.pad_w; p.dh = d.dil_h; p.dw = d.dil_w;
    p.sB = d.x_stride_b; p.sH = d.x_stride_h; p.sW = d.x_stride_w; p.coff = d.x_coff;
    p.ldY = d.y_ld; p.yoff = d.y_coff; p.ldR = d.r_ld; p.roff = d.r_coff;
    p.act = d.act; p.res_after = d.res_after_act;
    p.rsub = d.r_sub > 1 ? d.r_sub : 1; p.rH = d.r_h; p.rW = d.r_w;
    if (residual && d.r_sub > 1 && ((long)(d.out_h - 1) * d.r_sub >= d.r_h || (long)(d.out_w - 1) * d.r_sub >= d.r_w))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: residual grid %d x %d too small for %d x %d outputs at step %d", d.r_h, d.r_w,
                       d.out_h, d.out_w, d.r_sub);
    const int groups = d.groups > 1 ? d.groups : 1;
    const long x_extent = ((long)(d.batch - 1) * d.x_stride_b + (long)(d.in_h - 1) * d.x_stride_h +
                           (long)(d.in_w - 1) * d.x_stride_w + d.x_coff + (long)groups * d.cin) * es;
    const long w_extent = (long)groups * d.n * K * es;
    if (x_extent >= (long)OOB || w_extent >= (long)OOB)
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: operand larger than 4 GiB (%ld / %ld bytes): split the batch", x_extent,
                       w_extent);
    p.x_bytes = (unsigned)x_extent; p.w_bytes = (unsigned)w_extent;
    p.X2 = (const char*)x2; p.x2_bytes = 0; p.K1 = (int)K; p.sB2 = p.sH2 = p.sW2 = 0; p.coff2 = 0; p.st2 = 1;
    if (x2) {
        const long x2_extent = ((long)(d.batch - 1) * d.x2_stride_b + (long)(d.out_h - 1) * d.x2_stride * d.x2_stride_h +
                                (long)(d.out_w - 1) * d.x2_stride * d.x2_stride_w + d.x2_coff + d.x2_cin) * es;
        if (x2_extent >= (long)OOB) return set_err(ctx, AVCER_EINVAL, "conv_gemm: second source larger than 4 GiB");
        p.x2_bytes = (unsigned)x2_extent; p.K1 = (int)K1;
        p.sB2 = d.x2_stride_b; p.sH2 = d.x2_stride_h; p.sW2 = d.x2_stride_w; p.coff2 = d.x2_coff; p.st2 = d.x2_stride;
    }
    p.ntn = 0; p.nwg = 0; p.groups = groups; p.slots = ctx->block_slots; p.ovf = ctx->ovf;
    if (d.tile_n != 0 && d.tile_n != 64 && d.tile_n != 128 && d.tile_n != 256)
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: tile_n %d (0, 64, 128 or 256)", d.tile_n);
    p.tile_n = d.tile_n;
    if (skinny ? (d.tile_m != 0 && d.tile_m != 16 && d.tile_m != 32 && d.tile_m != 64) : (d.tile_m != 0 && d.tile_m != 112 && d.tile_m != 128))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: tile_m %d (0, 112 or 128; dtype 9 / 10: 0, 16, 32 or 64)", d.tile_m);
    p.tile_m = d.tile_m;
    p.WF = wdirect ? (const char*)w : nullptr;
    if (skinny && (x2 || d.cin % 32 || d.n % 32 || M > 4096))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: dtype %d (skinny form) needs one source, cin %% 32 == 0, n %% 32 == 0, M <= 4096 (M=%ld)", dtype, M);
    if (wdirect && !skinny && (d.n % 256 || (K / bk) % 2 || groups != 1 || d.tile_n == 64 || d.tile_n == 128))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: dtype %d needs N %% 256 == 0, an even number of K-steps, one group (N=%d, K=%ld)",
                       dtype, d.n, K);
    p.tapH4 = (int)((long)d.dil_h * d.x_stride_h * 4);
    p.tapW4 = (int)((long)d.dil_w * d.x_stride_w * 4);
    // Fast gather: Cin a multiple of the K-step, no padding, and the last tap of the last output position inside the
    // input -- true for every Linear, 1x1 convolution and un-padded Conv1d of both models.
    p.fast = d.cin % bk == 0 && d.pad_h == 0 && d.pad_w == 0 &&
             (long)(d.out_h - 1) * d.stride_h + (long)(d.kh - 1) * d.dil_h < d.in_h &&
             (long)(d.out_w - 1) * d.stride_w + (long)(d.kw - 1) * d.dil_w < d.in_w;
    if (wdirect && x2 && !p.fast) return set_err(ctx, AVCER_EINVAL, "conv_gemm: dtype %d with a second source needs a pad-free gather", dtype);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    {
        // compulsory traffic: the input once (each position's taps overlap), the weights once, the output (+ residual) once
        const double in_el = (double)d.batch * d.in_h * d.in_w * d.cin * groups + (x2 ? (double)M * d.x2_cin : 0.0);
        const double bytes = in_el * es + (double)w_extent + (double)M * d.n * groups * (dtype == 1 ? 2 : 4) * (residual ? 2 : 1);
        TRY(prof_begin(ctx, st, &ev0, &ev1, skinny ? FAM_SKINNY : wdirect ? FAM_GEMM_WD : FAM_GEMM, 2.0 * (double)M * (double)d.n * (double)K * groups, bytes,
                       M, (long)d.n * groups, K));
    }
    switch (dtype) {
        case 0: launch_t<0, 0>(p, st); break;  // f32
        case 1: launch_t<1, 1>(p, st); break;  // bf16 -> bf16
        case 2: launch_t<1, 0>(p, st); break;  // bf16 -> f32
        case 3: launch_t<2, 0>(p, st); break;  // f32 (split on the fly) -> f32
        case 4: launch_t<2, 2>(p, st); break;  // f32 (split on the fly) -> sp32
        case 5: launch_t<3, 2>(p, st); break;  // sp32 -> sp32
        case 6: launch_t<3, 0>(p, st); break;  // sp32 -> f32
        case 7: launch_wd<2>(p, st); break;    // sp32 -> sp32, weights direct
        case 8: launch_wd<0>(p, st); break;    // sp32 -> f32, weights direct
        case 9: launch_skinny<2>(p, st); break;   // sp32 -> sp32, a handful of positions
        default: launch_skinny<0>(p, st); break;  // sp32 -> f32, a handful of positions
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "conv_gemm launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += 2.0 * (double)M * (double)d.n * (double)K * groups;
    return AVCER_OK;
}
