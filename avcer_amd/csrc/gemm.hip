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
//   * 256 threads = 4 waves (2x2); block tile 128(m) x BN(n), BN in {128, 64}; one K-step = 128 bytes per row
//     (32 f32 or 64 bf16), so the global->LDS staging pattern is identical for both element types.
//   * A and W tiles are staged through registers (the A gather needs zero fill at image borders) into a
//     double-buffered LDS image of 128-byte rows whose 16-byte chunks are XOR-swizzled with (row>>1)&7:
//     conflict-free for the ds_read_b128 fragment reads of both MFMA shapes (MI355X LDS: 64 banks x 4 B,
//     b128 reads served in 16-lane groups).
//   * MFMA operands are swapped (weights = A operand, activations = B operand) so that each lane's accumulator
//     registers hold 4 CONSECUTIVE output channels of one position: NHWC stores and the per-channel
//     scale/bias loads become 16-byte (f32) / 8-byte (bf16) vector accesses.
//   * f32 mode uses v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, parity mode); bf16 mode uses
//     v_mfma_f32_16x16x32_bf16 with f32 accumulation.
//   * blockIdx is remapped so that the 8 XCDs (private L2 each) get contiguous runs of tiles.
#include "common.h"

#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

constexpr int BM = 128;
constexpr int ROWB = 128;  // bytes per tile row per K-step

struct GemmParams {
    const char* X;
    const char* W;
    const float* scale;
    const float* bias;
    const char* R;
    char* Y;
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
    int ntn, nwg;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : (v != v ? v : 0.f);  // relu keeps NaN like torch
    if (act == 2) return gelu_erf(v);
    return v;
}

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

template <typename OutT>
__device__ __forceinline__ void epilogue4(const GemmParams& p, long m, int n0, float v0, float v1, float v2, float v3) {
    float v[4] = {v0, v1, v2, v3};
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.R) {
        const char* rp = p.R + (m * p.ldR + p.roff + n0) * (long)sizeof(OutT);
        if constexpr (sizeof(OutT) == 4) {
            const float4 t = *reinterpret_cast<const float4*>(rp);
            r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
        } else {
            const uint2 t = *reinterpret_cast<const uint2*>(rp);
            r[0] = bf2f((bf16_t)(t.x & 0xffff)); r[1] = bf2f((bf16_t)(t.x >> 16));
            r[2] = bf2f((bf16_t)(t.y & 0xffff)); r[3] = bf2f((bf16_t)(t.y >> 16));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (p.res_after) v[j] = apply_act(v[j], p.act) + r[j];
        else v[j] = apply_act(v[j] + r[j], p.act);
    }
    char* yp = p.Y + (m * p.ldY + p.yoff + n0) * (long)sizeof(OutT);
    if constexpr (sizeof(OutT) == 4) {
        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 t;
        t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(yp) = t;
    }
}

template <typename T, typename OutT, int BN>
__global__ void __launch_bounds__(256) conv_gemm_kernel(const GemmParams p) {
    constexpr bool IS_F32 = sizeof(T) == 4;
    constexpr int ES = sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int BK = ROWB / ES;
    constexpr int A_VPT = BM / 32;
    constexpr int B_VPT = BN / 32;
    constexpr int TILE_BYTES = (BM + BN) * ROWB;
    constexpr int WN = BN / 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: blocks b and b+8 share an XCD (speed only, never correctness)
    int bid = blockIdx.x;
    {
        const int q = p.nwg >> 3, r = p.nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.ntn;
    const int tile_m = bid / p.ntn;
    const int m_base = tile_m * BM;
    const int n_base = tile_n * BN;

    const int chunk = tid & 7;
    const int rbase = tid >> 3;

    long a_base[A_VPT];
    int a_iy[A_VPT], a_ix[A_VPT];
#pragma unroll
    for (int i = 0; i < A_VPT; ++i) {
        const int m = m_base + rbase + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int ox = mm % p.OW;
        const int t = mm / p.OW;
        const int oy = t % p.OH;
        const int b = t / p.OH;
        const int iy = oy * p.sh - p.ph;
        const int ix = ox * p.sw - p.pw;
        a_iy[i] = ok ? iy : -(1 << 28);  // out-of-range rows fail the bounds test below
        a_ix[i] = ix;
        a_base[i] = (long)b * p.sB + (long)iy * p.sH + (long)ix * p.sW + p.coff;
    }
    int kc = chunk * VEC, kx = 0, ky = 0;
    while (kc >= p.Cin) {
        kc -= p.Cin;
        if (++kx == p.KW) { kx = 0; ++ky; }
    }
    const char* wptr = p.W + ((long)(n_base + rbase) * p.K + chunk * VEC) * ES;
    const long w_rowstep = (long)32 * p.K * ES;

    uint4 ra[A_VPT], rb[B_VPT];
    auto load_tiles = [&]() {
        const int dy = ky * p.dh, dx = kx * p.dw;
        const long toff = (long)dy * p.sH + (long)dx * p.sW + kc;
#pragma unroll
        for (int i = 0; i < A_VPT; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ok) v = *reinterpret_cast<const uint4*>(p.X + (a_base[i] + toff) * ES);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_VPT; ++i) rb[i] = *reinterpret_cast<const uint4*>(wptr + i * w_rowstep);
        wptr += ROWB;
        kc += BK;
        while (kc >= p.Cin) {
            kc -= p.Cin;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };
    auto store_tiles = [&](int buf) {
        char* sa = smem + buf * TILE_BYTES;
        char* sb = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_VPT; ++i) *reinterpret_cast<uint4*>(sa + swz(rbase + 32 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_VPT; ++i) *reinterpret_cast<uint4*>(sb + swz(rbase + 32 * i, chunk)) = rb[i];
    };

    constexpr int NFN = IS_F32 ? WN / 32 : WN / 16;
    constexpr int NFM = IS_F32 ? 2 : 4;
    using acc_t = typename std::conditional<IS_F32, f32x16_t, f32x4_t>::type;
    acc_t acc[NFN][NFM];
#pragma unroll
    for (int a = 0; a < NFN; ++a)
#pragma unroll
        for (int b = 0; b < NFM; ++b) acc[a][b] = acc_t{0};

    const int nk = p.K / BK;
    load_tiles();
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < nk; ++step) {
        const bool more = step + 1 < nk;
        if (more) load_tiles();
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * ROWB;
        if constexpr (IS_F32) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int ch = kq * 2 + (lane >> 5);
                float4 af[NFM], wf[NFN];
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm)
                    af[fm] = *reinterpret_cast<const float4*>(sa + swz(wm * 64 + fm * 32 + (lane & 31), ch));
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
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ks * 4 + (lane >> 4);
                bf16x8_t af[NFM], wf[NFN];
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm)
                    af[fm] = *reinterpret_cast<const bf16x8_t*>(sa + swz(wm * 64 + fm * 16 + (lane & 15), ch));
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
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: accumulator register group g of a lane = 4 consecutive output channels of one position
    if constexpr (IS_F32) {
#pragma unroll
        for (int fn = 0; fn < NFN; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = n_base + wn * WN + fn * 32 + 8 * g + 4 * (lane >> 5);
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n0);
                if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + n0);
#pragma unroll
                for (int fm = 0; fm < NFM; ++fm) {
                    const int m = m_base + wm * 64 + fm * 32 + (lane & 31);
                    if (m < p.M)
                        epilogue4<OutT>(p, m, n0, acc[fn][fm][4 * g + 0] * sc.x + bi.x, acc[fn][fm][4 * g + 1] * sc.y + bi.y,
                                        acc[fn][fm][4 * g + 2] * sc.z + bi.z, acc[fn][fm][4 * g + 3] * sc.w + bi.w);
                }
            }
    } else {
#pragma unroll
        for (int fn = 0; fn < NFN; ++fn) {
            const int n0 = n_base + wn * WN + fn * 16 + 4 * (lane >> 4);
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n0);
            if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + n0);
#pragma unroll
            for (int fm = 0; fm < NFM; ++fm) {
                const int m = m_base + wm * 64 + fm * 16 + (lane & 15);
                if (m < p.M)
                    epilogue4<OutT>(p, m, n0, acc[fn][fm][0] * sc.x + bi.x, acc[fn][fm][1] * sc.y + bi.y,
                                    acc[fn][fm][2] * sc.z + bi.z, acc[fn][fm][3] * sc.w + bi.w);
            }
        }
    }
}

template <typename T, typename OutT>
void launch_t(const GemmParams& p0, hipStream_t st) {
    GemmParams p = p0;
    const int ntm = (p.M + BM - 1) / BM;
    if (p.N % 128 == 0) {
        p.ntn = p.N / 128;
        p.nwg = ntm * p.ntn;
        conv_gemm_kernel<T, OutT, 128><<<dim3(p.nwg), dim3(256), 0, st>>>(p);
    } else {
        p.ntn = p.N / 64;
        p.nwg = ntm * p.ntn;
        conv_gemm_kernel<T, OutT, 64><<<dim3(p.nwg), dim3(256), 0, st>>>(p);
    }
}

}  // namespace

int launch_conv_gemm(avcer_ctx* ctx, const avcer_conv_desc& d, int dtype, const void* x, const void* w,
                     const float* scale, const float* bias, const void* residual, void* y, hipStream_t st) {
    if (dtype < 0 || dtype > 2) return set_err(ctx, AVCER_EINVAL, "conv_gemm: dtype %d", dtype);
    const int es = dtype == 0 ? 4 : 2;
    const int vec = 16 / es;
    const int bk = ROWB / es;
    const long M = (long)d.batch * d.out_h * d.out_w;
    const long K = (long)d.kh * d.kw * d.cin;
    if (M <= 0 || M > 0x7fffff00L) return set_err(ctx, AVCER_EINVAL, "conv_gemm: M=%ld out of range", M);
    if (d.n <= 0 || d.n % 64) return set_err(ctx, AVCER_EINVAL, "conv_gemm: N=%d must be a multiple of 64", d.n);
    if (K % bk) return set_err(ctx, AVCER_EINVAL, "conv_gemm: K=%ld must be a multiple of %d", K, bk);
    if (d.cin % vec || d.x_coff % vec || d.x_stride_b % vec || d.x_stride_h % vec)
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: cin/coff/strides must be multiples of %d", vec);
    if (d.x_stride_w % vec && !(d.kw == 1 && d.pad_w == 0 && (d.stride_w * d.x_stride_w) % vec == 0))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: x_stride_w=%ld breaks 16-byte alignment", (long)d.x_stride_w);
    if (d.y_ld % 4 || d.y_coff % 4 || (residual && (d.r_ld % 4 || d.r_coff % 4)))
        return set_err(ctx, AVCER_EINVAL, "conv_gemm: output/residual leading dims must be multiples of 4");
    if (!x || !w || !y) return set_err(ctx, AVCER_EINVAL, "conv_gemm: null pointer");
    GemmParams p;
    p.X = (const char*)x; p.W = (const char*)w; p.scale = scale; p.bias = bias; p.R = (const char*)residual;
    p.Y = (char*)y;
    p.M = (int)M; p.N = d.n; p.K = (int)K;
    p.OH = d.out_h; p.OW = d.out_w; p.H = d.in_h; p.Wd = d.in_w; p.Cin = d.cin; p.KW = d.kw;
    p.sh = d.stride_h; p.sw = d.stride_w; p.ph = d.pad_h; p.pw = d.pad_w; p.dh = d.dil_h; p.dw = d.dil_w;
    p.sB = d.x_stride_b; p.sH = d.x_stride_h; p.sW = d.x_stride_w; p.coff = d.x_coff;
    p.ldY = d.y_ld; p.yoff = d.y_coff; p.ldR = d.r_ld; p.roff = d.r_coff;
    p.act = d.act; p.res_after = d.res_after_act;
    p.ntn = 0; p.nwg = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->prof) {
        if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
            for (int i = 0; i < 512; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return set_err(ctx, AVCER_EHIP, "hipEventCreate failed");
                ctx->prof_ev.push_back(e);
            }
        }
        ev0 = ctx->prof_ev[ctx->prof_used++];
        ev1 = ctx->prof_ev[ctx->prof_used++];
        (void)hipEventRecord(ev0, st);
    }
    if (dtype == 0) launch_t<float, float>(p, st);
    else if (dtype == 1) launch_t<bf16_t, bf16_t>(p, st);
    else launch_t<bf16_t, float>(p, st);
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "conv_gemm launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += 2.0 * (double)M * (double)d.n * (double)K;
    return AVCER_OK;
}
