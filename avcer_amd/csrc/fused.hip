// Fused kernels of the static CNN in the split-fp16 ("x3") arithmetic (gfx950 only; csrc/split_dev.h).
//
//   stem_pool_kernel   preprocessed image (planar fp16 hi / lo) -> 7x7/2 convolution + BN + ReLU + 3x3/2 max-pool
//                      ref: architectures/video.py:63-90,98-103,116-117
//   bneck_kernel       the tail of one bottleneck and the head of the next one in ONE launch:
//                        T1 --3x3 conv2+BN+ReLU--> T2 --1x1 conv3+BN, +X, ReLU--> OUT --1x1 conv1'+BN+ReLU--> T1'
//                      ref: architectures/video.py:43-60 (Bottleneck.forward), one call per non-first block of a stage
//
// Why: layer by layer, the sp32 activations (4 bytes per element) of the 55x55 and 28x28 stages make conv3 + residual
// and the following conv1 pure HBM streams (4.4-5.5 TB/s measured, profiles/r01_layers_x3.txt), and the stem writes a
// 3.3 GB tensor per 1024 frames that the max-pool immediately re-reads.  Here T2 and the conv1' operand never leave
// the registers: with weights as the MFMA A operand, a lane's accumulator holds 4 consecutive output channels of one
// position; two 16-row weight tiles whose rows are stored in the order  row 16t + 4g + r  <->  channel 8g + 4t + r
// (t = 0,1; g = lane >> 4; r = 0..3) give every lane group g the 8 consecutive channels 8g..8g+7 of a 32-channel
// group, which is exactly the B-operand fragment of the NEXT contraction's K-step in natural K order -- and 16
// contiguous bytes of the sp32 output row, so the stores are whole 64-byte half-lines.  The row permutation is applied
// to the weights once, on the host (avcer_amd/packing.py: *.wp tensors).
#include "common.h"
#include "gemm_dev.h"

#include <algorithm>

namespace {

// a.w ~= ah.wh + ah.wl + al.wh on the f16 MFMA with f32 accumulation (same product order as conv_gemm MODE 3)
__device__ __forceinline__ void mfma3(f32x4_t& acc, const spx8_t wh, const spx8_t wl, const spx8_t ah, const spx8_t al) {
    acc = mfma_sp(wl, ah, acc);
    acc = mfma_sp(wh, al, acc);
    acc = mfma_sp(wh, ah, acc);
}

// Keeps every MFMA that produced these accumulators in front of what follows (a barrier): left alone, hipcc hoists the
// s_waitcnt vmcnt(0) + s_barrier that ends a K-step above the last quarter to half of the step's MFMAs, and the LDS-DMA
// issued at the top of the step gets that much less time to land.
template <int A, int B>
__device__ __forceinline__ void pin(f32x4_t (&acc)[A][B]) {
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b) asm volatile("" : "+v"(acc[a][b]));
}
template <int A>
__device__ __forceinline__ void pin(f32x4_t (&acc)[A]) {
#pragma unroll
    for (int a = 0; a < A; ++a) asm volatile("" : "+v"(acc[a]));
}

__device__ __forceinline__ spx8_t ldfrag(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const spx8_t*>(tile + swz(row, chunk));
}

// 8 f32 values -> fp16 hi / lo fragments (value = hi + lo + O(2^-22))
__device__ __forceinline__ void split8v(const float (&v)[8], spx8_t& hi, spx8_t& lo, sp_flags_t& ovm) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float x0 = sp_value(v[j]), x1 = sp_value(v[j + 1]);  // one f32 number for both halves (split_dev.h)
        a = sp_max2(a, x0, x1);
        const spe_t h0 = (spe_t)x0, h1 = (spe_t)x1;
        hi[j] = h0;
        hi[j + 1] = h1;
        lo[j] = (spe_t)(x0 - (float)h0);
        lo[j + 1] = (spe_t)(x1 - (float)h1);
    }
    sp_flag(ovm, a);  // range contract: |x| < 65504 (split_dev.h sp_commit)
}

__device__ __forceinline__ void unpack8(const uint4 h, const uint4 l, float (&r)[8]) {
    const uint32_t wh[4] = {h.x, h.y, h.z, h.w}, wl[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[2 * j] = sp2f((uint16_t)(wh[j] & 0xffff)) + sp2f((uint16_t)(wl[j] & 0xffff));
        r[2 * j + 1] = sp2f((uint16_t)(wh[j] >> 16)) + sp2f((uint16_t)(wl[j] >> 16));
    }
}

// 8 consecutive floats from LDS by inline asm.  A ds_read that hipcc can see makes it drain every LDS-DMA in flight first
// (s_waitcnt vmcnt(0): it cannot prove that the read does not alias the DMA destination), which would undo the counted
// wait of bneck_tail2_kernel; the table read here was written long before (barriers in between).
__device__ __forceinline__ void lds_read8(const float* ptr, f32x4_t& a, f32x4_t& b) {
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)ptr;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------------ stem + max-pool
struct StemParams {
    const char* P;         // fp16 hi plane [n][230][230][4]; the lo plane starts plane_bytes later
    const uint8_t* F;      // u8 form: the frames themselves, u8 [n][in_h][in_w][3] RGB (data/utils.py:19-39 happens in the kernel)
    int in_h, in_w;
    const float* bias9;    // u8 form: [9 border classes][64]: BN shift - BN scale * sum over the VALID taps of w * channel mean
    unsigned p_bytes;      // extent of both planes (hardware bounds check)
    unsigned plane_bytes;
    const char* W;         // split weights [64][7 tap rows x 32] (sp32 groups of 32 K-elements, rows permuted)
    const float* scale;
    const float* bias;
    char* Y;               // sp32 [n][55][55][64]
    int n;
    unsigned* ovf;         // the context's range-contract counter (split_dev.h sp_commit)
    // FACE form (stem_pool_u8_kernel<true>: the detector's stem, any frame size): stem map oh x ow, pooled map mh x mw, tiles of
    // 8 x 7 pooled outputs per frame, channel order of the frames, the integer channel means of the net's (B, G, R) inputs
    int oh, ow, mh, mw, tiles_y, tiles_x, swap_rb;
    int mean[3];
};

constexpr int ST_TH = 8, ST_TW = 7;                       // pooled tile
constexpr int ST_RH = 2 * ST_TH + 1, ST_RW = 2 * ST_TW + 1;  // stem region 17 x 15 = 255 positions
constexpr int ST_TY = 7, ST_TX = 8;                       // tiles per frame (7 x 8 >= 55 / 8 x 55 / 7)

__device__ __forceinline__ int stage64(int row, int chunk) { return row * 256 + ((chunk ^ (row & 7)) << 4); }

// One block = one 8 x 7 tile of pooled outputs of one frame: the 17 x 15 stem positions under it are one 256-row
// implicit-GEMM tile (K = 7 tap rows x 8 pixels x 4 channels, N = 64), the pool runs on the f32 image in LDS.
// Neighbouring tiles recompute one shared stem row / column (255 positions per 224 useful ones).
//
// The A operand is never expanded: the 39 x 36-pixel patch of the (already padded) planar image under the tile is copied
// to LDS once, as it lies in memory (hi plane, then lo plane; 22 KiB), and a fragment is read straight out of it -- the
// 8 K-elements of lane group g in tap row ky are the 2 pixels x 4 channels at patch[(2 ry + ky)][2 rx + 2 g], 16
// contiguous, 16-byte-aligned bytes.  (An im2col tile per tap row moved 224 KiB through the L2 -> LDS path per block.)
// All seven weight tiles (56 KiB) arrive with the same burst, so the K loop has no barrier and no wait in it; the second
// block of the CU computes while this one waits for its burst.  Rows of positions outside the 112 x 112 map read whatever
// follows in memory (or zeros past the planes): each output row depends on its own A row only, and the pool never reads them.
__global__ void __launch_bounds__(256, 2) stem_pool_kernel(const StemParams p) {
    constexpr int BN = 64, NK = 7;
    constexpr int PROWS = 2 * (ST_RH - 1) + NK, PCH = (2 * (ST_RW - 1) + 8) / 2;  // 39 patch rows of 18 16-byte chunks (36 pixels)
    constexpr int PLANE = PROWS * PCH * 16;                                       // 11 232 B per plane
    constexpr int NCH = 2 * PROWS * PCH;                                          // 1 404 chunks
    constexpr int WOFF = ((2 * PLANE + 1023) / 1024) * 1024;                      // weights behind the patch
    constexpr int WT = BN * ROWB;                                                 // one tap row's weight tile, 8 KiB
    constexpr int LDS_BYTES = WOFF + NK * WT > 256 * 256 ? WOFF + NK * WT : 256 * 256;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int b = blk / (ST_TY * ST_TX), t = blk % (ST_TY * ST_TX);
    const int ty = t / ST_TX, tx = t % ST_TX;

    const float wmul = split_wmul(p.W, 64 * NK * ROWB);
    const auto prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.P), (short)0, (int)p.p_bytes, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W), (short)0, 64 * NK * ROWB, 0x00020000);
    // patch: chunk c of the LDS image = chunk (c % 702) of plane c / 702; one DMA instruction copies 64 consecutive chunks
    const unsigned origin = (unsigned)((((long)b * 230 + 4 * ST_TH * ty) * 230 + 4 * ST_TW * tx) * 8);
#pragma unroll
    for (int j = 0; j < (NCH + 255) / 256; ++j) {
        const int piece = wave + 4 * j;
        const int c = piece * 64 + lane;
        const int plane = c >= NCH / 2, ci = c - plane * (NCH / 2);
        const int prow = ci / PCH, pc = ci - prow * PCH;
        const unsigned off = c < NCH ? origin + (unsigned)(prow * (230 * 8) + pc * 16) + plane * p.plane_bytes : OOB;
        if (piece * 64 < NCH) dma16(prs, smem + piece * 1024, off);
    }
    // weights: tap row ky = rows [64][32 K] of p.W at column block ky, 8 DMA instructions of 8 rows each, swizzled like every tile
    {
        const int lrow8 = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < NK * 8 / 4; ++j) {
            const int piece = wave + 4 * j;  // 0 .. 55
            const int ky = piece >> 3, row = (piece & 7) * 8 + lrow8;
            dma16(wrs, smem + WOFF + piece * 1024, (unsigned)(row * (NK * ROWB) + ky * ROWB + ((slot ^ swz_key(row)) << 4)));
        }
    }

    f32x4_t acc[4][4];  // [channel tile][position tile]: wave = 64 positions x 64 channels
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0};
    const int g = lane >> 4, l15 = lane & 15;
    int abase[4];  // byte offset of this lane's fragment in tap row 0 of the hi plane
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        const int row = wave * 64 + fm * 16 + l15;
        const int ry = row / ST_RW, rx = row - ry * ST_RW;
        abase[fm] = row < ST_RH * ST_RW ? ((2 * ry) * PCH + rx + g) * 16 : 0;
    }
    __syncthreads();  // patch and weights have landed
#pragma unroll
    for (int ky = 0; ky < NK; ++ky) {
        const char* sb = smem + WOFF + ky * WT;
        spx8_t ah[4], al[4];
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            ah[fm] = *reinterpret_cast<const spx8_t*>(smem + abase[fm] + ky * (PCH * 16));
            al[fm] = *reinterpret_cast<const spx8_t*>(smem + PLANE + abase[fm] + ky * (PCH * 16));
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            const spx8_t wh = ldfrag(sb, fn * 16 + l15, g), wl = ldfrag(sb, fn * 16 + l15, 4 + g);
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) mfma3(acc[fn][fm], wh, wl, ah[fm], al[fm]);
        }
    }
    pin(acc);
    __syncthreads();  // every wave is done with the patch: the f32 image below overwrites it
    // BN + ReLU, parked as an f32 [256 positions][64 channels] image (64 KiB of the 80 KiB tile buffers)
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        const int ch = 32 * (fn >> 1) + 8 * g + 4 * (fn & 1);  // weight rows are stored permuted (split_weight_rows_kernel)
        float4 sc = *reinterpret_cast<const float4*>(p.scale + ch);
        const float4 bi = *reinterpret_cast<const float4*>(p.bias + ch);
        sc = make_float4(sc.x * wmul, sc.y * wmul, sc.z * wmul, sc.w * wmul);  // undoes the power of two of the weight split
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int row = wave * 64 + fm * 16 + l15;
            *reinterpret_cast<float4*>(smem + stage64(row, ch >> 2)) =
                make_float4(relu_nan(acc[fn][fm][0] * sc.x + bi.x), relu_nan(acc[fn][fm][1] * sc.y + bi.y),
                            relu_nan(acc[fn][fm][2] * sc.z + bi.z), relu_nan(acc[fn][fm][3] * sc.w + bi.w));
        }
    }
    __syncthreads();
    // 3x3/2 max-pool over the image (no padding: video.py:103), 8 channels per thread, whole-line sp32 stores
    sp_flags_t ovm = 0;  // range contract of the sp32 output (split_dev.h sp_commit)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int item = pass * 256 + tid;
        const int pp = item >> 3, c8 = item & 7;
        const int ppy = pp / ST_TW, ppx = pp - ppy * ST_TW;
        const int py = ST_TH * ty + ppy, px = ST_TW * tx + ppx;
        if (pp >= ST_TH * ST_TW || py >= 55 || px >= 55) continue;
        float m[8];
        bool nan = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = 0.f;  // post-ReLU values are >= 0
        bool anynan[8] = {false, false, false, false, false, false, false, false};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int row = (2 * ppy + dy) * ST_RW + 2 * ppx + dx;
                const float4 u = *reinterpret_cast<const float4*>(smem + stage64(row, 2 * c8));
                const float4 v = *reinterpret_cast<const float4*>(smem + stage64(row, 2 * c8 + 1));
                const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) { m[j] = fmaxf(m[j], x[j]); anynan[j] |= x[j] != x[j]; }
            }
        (void)nan;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (anynan[j]) m[j] = NAN;  // like torch's max-pool
        spx8_t hi, lo;
        split8v(m, hi, lo, ovm);
        const long e = (((long)b * 55 + py) * 55 + px) * 64 + c8 * 8;
        char* yp = p.Y + sp32_byte(e);
        *reinterpret_cast<spx8_t*>(yp) = hi;
        *reinterpret_cast<spx8_t*>(yp + 64) = lo;
    }
    sp_commit(p.ovf, ovm);
}

// The stem of the x3 mode when the input is the u8 frames (avcer_static_forward): same tiling, but
//  * the block builds its patch itself (BGR flip, PIL NEAREST resize when the frame is not 224 x 224, zeros outside the image)
//    as RAW PIXEL VALUES: integers 0..255 are exact in fp16, so the activation has no lo half -- two MFMAs per product, one
//    patch plane, no preprocessing launch and no 846 KB-per-frame image in HBM.  The mean subtraction of data/utils.py:36-38
//    moves into the shift: sum_valid w (p - mu) = sum_valid w p - sum_valid w mu, "valid" = the taps inside the image
//    (Conv2dSame pads the NORMALISED image with zeros, video.py:68-80): a constant per output channel and BORDER CLASS of the
//    stem position (first row / interior / row 110, columns alike: 9 classes; avcer_amd/packing.py stem_border_shifts,
//    folded with the BatchNorm shift in float64);
//  * the block is small enough for THREE per CU (the planar form: two): the weight tiles stream through a three-slot ring,
//    one tap row per step (counted wait: the tile requested at the top of a step may still be in flight at its barrier), and
//    the BN + ReLU image is parked and pooled in two halves of 32 channels (32 KiB instead of 64).  The block is a serial
//    chain -- patch, MFMAs, image, pool, store -- so what hides one block's latencies is the other blocks of its CU.
//
// FACE = true (round 6): the stem of the RetinaFace body (torchvision ResNet-50: conv 7x7/2 pad 3, BN, ReLU, max-pool 3x3/2 pad 1;
// retina_face.py:46-76, retina_face_predictor.py:59-66) on frames of ANY size.  Same tiling -- a block = 8 x 7 pooled outputs, the
// 17 x 15 stem positions under them, a 39 x 36-pixel patch -- with the window origins of symmetric padding (patch origin
// (4 py0 - 5, 4 px0 - 5); stem row -1 / column -1 and those past the map are the pool's padding and are skipped).  The net's
// input is pixel - (104, 117, 123): INTEGER means, so the mean-subtracted value itself is exact in fp16 -- no lo half, two
// MFMAs per product, zero padding is exactly zero and no border classes are needed (the BN shift is the plain one).
// Replaces face_pre_kernel + the f32-input stem contraction + maxpool3s2p1_kernel: 0.9 + 7.9 + 3.2 ms per 750 frames of
// 640 x 360 (profiles/r06_face_trace_before_stem.txt), the stem map [n,180,320,64] written and read once each for nothing.
template <bool FACE>
__global__ void __launch_bounds__(256, 3) stem_pool_u8_kernel(const StemParams p) {
    constexpr int NK = 7, WRING = 3;
    constexpr int PROWS = 2 * (ST_RH - 1) + NK, PCH = (2 * (ST_RW - 1) + 8) / 2;  // 39 patch rows of 18 16-byte chunks (36 pixels)
    constexpr int NCH = PROWS * PCH;                                              // 702 chunks, 11 232 B
    constexpr int WOFF = ((NCH * 16 + 1023) / 1024) * 1024;                       // weight ring behind the patch
    constexpr int WT = 64 * ROWB;                                                 // one tap row's weight tile, 8 KiB
    constexpr int IMG = 256 * 128;                                                // f32 [256 positions][32 channels]
    constexpr int LDS_BYTES = WOFF + WRING * WT > IMG ? WOFF + WRING * WT : IMG;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_x = FACE ? p.tiles_x : ST_TX, tiles = FACE ? p.tiles_y * p.tiles_x : ST_TY * ST_TX;
    const int b = blk / tiles, t = blk % tiles;
    const int ty = t / tiles_x, tx = t % tiles_x;
    const int lrow8 = lane >> 3, slot = lane & 7;
    const float wmul = split_wmul(p.W, 64 * NK * ROWB);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W), (short)0, 64 * NK * ROWB, 0x00020000);
    // tap row ky = rows [64][32 K] of p.W at column block ky: 8 DMA pieces of 8 rows, two per wave, swizzled like every tile
    unsigned w_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + lrow8;
        w_off[j] = (unsigned)(row * (NK * ROWB) + ((slot ^ swz_key(row)) << 4));
    }
#define AVCER_STEM_W(KY)                                                                                              \
    do {                                                                                                              \
        char* dst_ = smem + WOFF + ((KY) % WRING) * WT + wave * 2048;                                                 \
        dma16(wrs, dst_, w_off[0], (unsigned)((KY) * ROWB));                                                          \
        dma16(wrs, dst_ + 1024, w_off[1], (unsigned)((KY) * ROWB));                                                   \
    } while (0)
    AVCER_STEM_W(0);
    AVCER_STEM_W(1);
    // patch chunk ci = patch row ci / 18, pixels 2 (ci % 18), + 1 of the zero-bordered 230 x 230 image: [B G R 0] x 2 as fp16
    {
        // every byte load of the thread is issued before the first one is used: the addresses are clamped into the frame
        // (pixels outside the image and chunks past the patch load a valid byte and are zeroed afterwards), so nothing here
        // branches around a load and the three passes cost one memory round trip, not six
        const uint8_t* fr = p.F + (long)b * p.in_h * p.in_w * 3;
        const bool resize = !FACE && (p.in_h != 224 || p.in_w != 224);
        const int ih = FACE ? p.in_h : 224, iw = FACE ? p.in_w : 224, org = FACE ? 5 : 2;  // extent of the (resized) image, padding offset
        constexpr int PASSES = (NCH + 255) / 256;
        uint8_t raw[PASSES][2][3];
        bool ok[PASSES][2];
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const int ci = min(j * 256 + tid, NCH - 1);
            const int prow = ci / PCH, pc = ci - prow * PCH;
            const int iy = 4 * ST_TH * ty + prow - org;
            int sy = min(max(iy, 0), ih - 1);
            if (resize) sy = min((int)(((double)sy + 0.5) * ((double)p.in_h / 224.0)), p.in_h - 1);  // PIL NEAREST
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int ix = 4 * ST_TW * tx + 2 * pc + px - org;
                int sx = min(max(ix, 0), iw - 1);
                if (resize) sx = min((int)(((double)sx + 0.5) * ((double)p.in_w / 224.0)), p.in_w - 1);
                ok[j][px] = (unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw;
                const uint8_t* q = fr + ((long)sy * p.in_w + sx) * 3;
                raw[j][px][0] = q[0]; raw[j][px][1] = q[1]; raw[j][px][2] = q[2];
            }
        }
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const int ci = j * 256 + tid;
            uint32_t wds[4];
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                // [B G R 0]; integers up to 255 need 8 significant bits: the fp16 is exact
                uint32_t lo2, hi2;
                if constexpr (FACE) {  // pixel - integer mean, |.| <= 151: exact as well; frames are BGR unless swap_rb
                    const int c0 = (int)raw[j][px][p.swap_rb ? 2 : 0] - p.mean[0], c1 = (int)raw[j][px][1] - p.mean[1];
                    const int c2 = (int)raw[j][px][p.swap_rb ? 0 : 2] - p.mean[2];
                    lo2 = (uint32_t)f2sp((float)c0) | ((uint32_t)f2sp((float)c1) << 16);
                    hi2 = (uint32_t)f2sp((float)c2);
                } else {
                    lo2 = (uint32_t)f2sp((float)raw[j][px][2]) | ((uint32_t)f2sp((float)raw[j][px][1]) << 16);
                    hi2 = (uint32_t)f2sp((float)raw[j][px][0]);
                }
                wds[2 * px] = ok[j][px] ? lo2 : 0u;
                wds[2 * px + 1] = ok[j][px] ? hi2 : 0u;
            }
            if (ci < NCH) *reinterpret_cast<uint4*>(smem + ci * 16) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
    }
    f32x4_t acc[4][4];  // [channel tile][position tile]: wave = 64 positions x 64 channels
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0};
    const int g = lane >> 4, l15 = lane & 15;
    int abase[4];  // byte offset of this lane's fragment in tap row 0 of the patch
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        const int row = wave * 64 + fm * 16 + l15;
        const int ry = row / ST_RW, rx = row - ry * ST_RW;
        abase[fm] = row < ST_RH * ST_RW ? ((2 * ry) * PCH + rx + g) * 16 : 0;
    }
    __syncthreads();  // the patch is written, tap rows 0 and 1 have landed (this one drains everything)
#pragma unroll
    for (int ky = 0; ky < NK; ++ky) {
        // slot (ky + 2) % 3 held tap row ky - 1: every wave left it at the barrier that ended the previous step
        if (ky + 2 < NK) AVCER_STEM_W(ky + 2);
        asm volatile("" ::: "memory");
        const char* sb = smem + WOFF + (ky % WRING) * WT;
        spx8_t ah[4];
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) ah[fm] = *reinterpret_cast<const spx8_t*>(smem + abase[fm] + ky * (PCH * 16));
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            const spx8_t wh = ldfrag(sb, fn * 16 + l15, g), wl = ldfrag(sb, fn * 16 + l15, 4 + g);
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) {  // a = ah exactly: a.w = ah.wl + ah.wh (mfma3's order without its al term)
                acc[fn][fm] = mfma_sp(wl, ah[fm], acc[fn][fm]);
                acc[fn][fm] = mfma_sp(wh, ah[fm], acc[fn][fm]);
            }
        }
        pin(acc);
        // tap row ky + 1 (requested a step ago) must be in; the two pieces just requested may stay in flight
        if (ky + 2 < NK) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
#undef AVCER_STEM_W
    // BN + ReLU -> f32 [256 positions][32 channels] image (over the patch and the ring: every wave is past the last barrier),
    // then the 3x3/2 max-pool (no padding: video.py:103) of those 32 channels; twice
    sp_flags_t ovm = 0;  // range contract of the sp32 output (split_dev.h sp_commit)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
            const int fn = 2 * half + fh;
            const int ch = 32 * half + 8 * g + 4 * fh;  // weight rows are stored permuted (split_weight_rows_kernel)
            float4 sc = *reinterpret_cast<const float4*>(p.scale + ch);
            sc = make_float4(sc.x * wmul, sc.y * wmul, sc.z * wmul, sc.w * wmul);  // undoes the power of two of the weight split
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) {
                const int row = wave * 64 + fm * 16 + l15;
                // border class of stem position (cy, cx): taps above / left of the image at 0, below / right of it at 110
                // (111 is computed but never pooled)
                const int ry = row / ST_RW, rx = row - ry * ST_RW;
                const int cy = 2 * ST_TH * ty + ry, cx = 2 * ST_TW * tx + rx;
                const int cls = 3 * (cy == 0 ? 0 : (cy >= 110 ? 2 : 1)) + (cx == 0 ? 0 : (cx >= 110 ? 2 : 1));
                const float4 bi = *reinterpret_cast<const float4*>(FACE ? p.bias + ch : p.bias9 + cls * 64 + ch);
                const int chunk = 2 * g + fh;  // 16-byte chunk of the 32-channel row
                *reinterpret_cast<float4*>(smem + row * 128 + ((chunk ^ (row & 7)) << 4)) =
                    make_float4(relu_nan(acc[fn][fm][0] * sc.x + bi.x), relu_nan(acc[fn][fm][1] * sc.y + bi.y),
                                relu_nan(acc[fn][fm][2] * sc.z + bi.z), relu_nan(acc[fn][fm][3] * sc.w + bi.w));
            }
        }
        __syncthreads();
        {
            const int pp = tid >> 2, c8 = tid & 3;  // pooled position of the tile, group of 8 channels of this half
            const int ppy = pp / ST_TW, ppx = pp - ppy * ST_TW;
            const int py = ST_TH * ty + ppy, px = ST_TW * tx + ppx;
            if (pp < ST_TH * ST_TW && py < (FACE ? p.mh : 55) && px < (FACE ? p.mw : 55)) {
                float m[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = 0.f;  // post-ReLU values are >= 0
                bool anynan[8] = {false, false, false, false, false, false, false, false};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int row = (2 * ppy + dy) * ST_RW + 2 * ppx + dx;
                        if constexpr (FACE) {  // region row r <-> stem row 2 py0 - 1 + r: the pool's padding and the map's end are skipped
                            const int cy = 2 * py - 1 + dy, cx = 2 * px - 1 + dx;
                            if ((unsigned)cy >= (unsigned)p.oh || (unsigned)cx >= (unsigned)p.ow) continue;
                        }
                        const float4 u = *reinterpret_cast<const float4*>(smem + row * 128 + (((2 * c8) ^ (row & 7)) << 4));
                        const float4 v = *reinterpret_cast<const float4*>(smem + row * 128 + (((2 * c8 + 1) ^ (row & 7)) << 4));
                        const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) { m[j] = fmaxf(m[j], x[j]); anynan[j] |= x[j] != x[j]; }
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j) if (anynan[j]) m[j] = NAN;  // like torch's max-pool
                spx8_t hi, lo;
                split8v(m, hi, lo, ovm);
                const long e = (((long)b * (FACE ? p.mh : 55) + py) * (FACE ? p.mw : 55) + px) * 64 + 32 * half + c8 * 8;
                char* yp = p.Y + sp32_byte(e);
                *reinterpret_cast<spx8_t*>(yp) = hi;
                *reinterpret_cast<spx8_t*>(yp + 64) = lo;
            }
        }
        if (half == 0) __syncthreads();  // the pool of the first half has read the image the second half overwrites
    }
    sp_commit(p.ovf, ovm);
}

// ------------------------------------------------------------------------------------------------ bottleneck chain
struct BneckParams {
    const char* T1;     // sp32 [M][P]: conv1 output of THIS block (3x3 input)
    const char* X;      // sp32 [M][4P]: block input (residual)
    char* OUT;          // sp32 [M][4P]
    char* T1N;          // sp32 [M][P]: conv1 output of the NEXT block (null when there is none)
    const char* W2;     // [P][9P] split, rows permuted, BN scale folded into the rows
    const char* W2F;    // the same in MFMA fragment order (avcer_weight_frags): the T11 form loads it straight into registers
    const char* W3;     // [4P][P] split, rows permuted, BN scale folded
    const char* W1N;    // [P][4P] split, rows permuted, BN scale folded (next block's conv1)
    const float *b2, *b3, *b1n;  // folded BN shifts, natural channel order
    unsigned t1_bytes;
    int M, H, Wd;       // M = nb * H * Wd positions (SUB > 1: nb * OH * OW, see bneck_kernel)
    int OH, OW;         // SUB > 1: the output grid, position (oy, ox) <-> input position (SUB oy, SUB ox)
    unsigned* ovf;      // the context's range-contract counter (split_dev.h sp_commit)
    int nblocks;        // logical blocks (tiles); the grid may be smaller: a block then walks tiles blockIdx.x, + gridDim.x, ...
};

// BM positions per block, 4 waves, each wave owns BM/4 positions and ALL channels (so that a position's whole T2 /
// OUT row lives in one wave's registers).  LDS: phase A double-buffers [BM + P rows][128 B]; phases B/C double-buffer
// the weights of one 32-channel output group: P/32 tiles of W3 [32 rows][128 B] + one tile of W1N [P rows][128 B].
// NQX = 0: X is the block input [M][4P], added as the residual.  NQX > 0 (first block of a stage without spatial stride):
// X is the block input [M][32 NQX], the operand of the downsample convolution, whose weights are the K tail of W3
// ([4P][P + 32 NQX], both BN scales folded, packing.py c3d): its rows are loaded straight into B-operand fragments
// (an sp32 row IS the fragment layout: 16 bytes hi + 16 bytes lo per lane group and K-step) and there is no residual.
//
// PATCH (conv2 phase): instead of gathering a fresh [BM][32] activation tile per (tap, channel chunk) -- 9 x the bytes --
// the block keeps, per 32-channel chunk, the HALO PATCH of its 128 consecutive positions resident in LDS: every image row
// the positions touch plus one above and below, each with one zero slot left and right (image borders and the gap between
// two images are zero rows / slots supplied by the DMA's bounds check).  A tap is then a constant slot offset
// (dy * (W + 2) + dx) on the fragment reads, and only the weight tile streams per K-step (L2 -> LDS bytes of the phase
// -37 % at planes 128).  Used at planes 128; see launch_bneck for why not at planes 64.
//
// SUB = 2: the LAST block of a stage.  Its output is read by the next stage's conv1 and downsample convolution only, both
// 1x1 with stride 2 and no padding (video.py:12-19,140-149), i.e. at positions (2 oy, 2 ox): the block is evaluated at
// those positions alone -- M counts them, OUT is the compact [nb][OH][OW][4P] tensor, T1 and X keep their [nb][H][Wd]
// grids (the 3x3 taps of an even position touch odd ones; the residual is picked at the even position).  A quarter of the
// conv2 / conv3 products, of the residual reads and of the output bytes; every value written is the one the full-
// resolution evaluation would have put at that position.
//
// T11 (round 5; planes 64 on the 55 x 55 images of stage 1): in-kernel stamps (tools/clock_lab.py,
// profiles/r05_inkernel_clock_shares.txt) put 48 % of a block's residency into the conv2 loop -- 18 K-steps of 24 MFMAs per
// wave, each waiting 1.7 us for its own LDS-DMA tile behind the other blocks' HBM streams -- and 32 % into the streaming loop.
// Here a block is an 11 x 11 SPATIAL tile of one image (25 tiles per image, 121 of 128 rows used): its 13 x 13 halo patch
// of T1 (both 32-channel chunks, 44 KiB: three blocks still share a CU) is copied to LDS ONCE, a tap is a constant slot
// offset on the fragment reads, and the conv2 weights never touch LDS: every wave loads its W2 fragments (fragment-order
// copy, 1 KiB per load, L2-resident) straight into registers two K-steps ahead.  After the patch barrier the conv2 phase has
// no barrier, no DMA and no wait on HBM.  Same K order (tap, channel chunk) as the gather form: bit-identical results.
template <int P, int BM, bool NEXT, int NQX, bool PATCH, int SUB = 1, bool T11 = false>
__global__ void __launch_bounds__(256, P == 64 ? 3 : 2) bneck_kernel(const BneckParams p) {
    static_assert(SUB == 1 || (SUB == 2 && !NEXT && NQX == 0 && !PATCH), "the strided form is the plain last block of a stage");
    static_assert(!T11 || (P == 64 && BM == 128 && !PATCH && SUB == 1), "the spatial-tile form serves planes 64 at full resolution");
    constexpr int NQ = P / 32;            // K-steps of a P-channel contraction
    constexpr int NQT = NQ + NQX;         // K-steps of conv3 (+ downsample)
    constexpr int PSLOTS = PATCH ? (P == 64 ? 456 : 304) : 0;  // patch slots (x 128 B): 10 rows x 30 (28x28); 8 x 57 would serve 55x55
    constexpr int T11_E = 11, T11_PW = T11_E + 2, T11_SLOTS = 176;  // tile edge, patch edge, patch slots per chunk plane (169 used: 22 DMA pieces)
    constexpr int NT = BM / 64;           // 16-position tiles per wave
    constexpr int NG = 4 * P / 32;        // 32-channel groups of the block output
    constexpr int TILE_A = (BM + P) * ROWB;
    constexpr int TILE_B = (32 * NQT + P) * ROWB;
    constexpr int PHASE_A = T11 ? NQ * T11_SLOTS * ROWB : (PATCH ? PSLOTS * ROWB + 2 * P * ROWB : 2 * TILE_A);  // patch (+ two weight tiles), or two full stages
    constexpr int TILES = PHASE_A > 2 * TILE_B ? PHASE_A : 2 * TILE_B;
    constexpr int NBIAS = 6 * P;  // b2 [P], b1n [P], b3 [4P]: read back as broadcast float4 pairs in the epilogues
    __shared__ __attribute__((aligned(16))) char smem[TILES + NBIAS * 4];
    float* sbias = reinterpret_cast<float*>(smem + TILES);
    for (int i = threadIdx.x; i < NBIAS; i += 256)
        sbias[i] = i < P ? p.b2[i] : (i < 2 * P ? (NEXT ? p.b1n[i - P] : 0.f) : p.b3[i - 2 * P]);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    sp_flags_t ovm = 0;  // lanes that split a finite |x| >= 65520 into an fp16 pair (split_dev.h sp_commit)
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, l15 = lane & 15;
    const int lrow8 = lane >> 3, slot = lane & 7;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int m_base = blk * BM;  // linear forms: first position of the block
    // T11: image and tile origin of this block; row r of the block <-> tile position (r / 11, r % 11), rows 121..127 idle
    const int t11_b = T11 ? blk / 25 : 0, t11_y0 = T11 ? ((blk % 25) / 5) * T11_E : 0, t11_x0 = T11 ? (blk % 5) * T11_E : 0;
    // block row -> position index of the [nb][H][Wd] grid (clamped to a valid one where the row is idle / past M)
    auto row_pos = [&](int r, bool& ok) -> int {
        if constexpr (T11) {
            ok = r < T11_E * T11_E;
            const int rr = ok ? r : T11_E * T11_E - 1;
            const int ry = rr / T11_E, rx = rr - ry * T11_E;
            return (t11_b * 55 + t11_y0 + ry) * 55 + t11_x0 + rx;
        } else {
            const long m = (long)m_base + r;
            ok = m < p.M;
            return ok ? (int)m : 0;
        }
    };

    // accumulator multipliers of the three scaled weight splits (trailers behind the matrices: split_dev.h)
    const float s2 = split_wmul(p.W2, P * 9 * P * 4), s3 = split_wmul(p.W3, 4 * P * (P + 32 * NQX) * 4);
    const float s1n = NEXT ? split_wmul(p.W1N, 4 * P * P * 4) : 1.f;
    (void)s1n;
    const auto t1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.T1), (short)0, (int)p.t1_bytes, 0x00020000);
    const auto w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W2), (short)0, P * 9 * P * 4, 0x00020000);
    const auto w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W3), (short)0, 4 * P * (P + 32 * NQX) * 4, 0x00020000);
    const auto w1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(NEXT ? p.W1N : p.W3), (short)0, 4 * P * P * 4, 0x00020000);

    // ---------------- phase A: T2 = relu(bn2(conv3x3(T1))), K = 9 taps x P channels, accumulators [P][BM/4] per wave
    constexpr int A_ISS = BM / 32;  // A-tile DMA instructions per wave per K-step (8 rows each)
    constexpr int W_ISS = P / 32;
    unsigned w_off[W_ISS];
#pragma unroll
    for (int j = 0; j < W_ISS; ++j) {
        const int row = wave * (W_ISS * 8) + j * 8 + lrow8;
        w_off[j] = (unsigned)((long)row * (9 * P * 4) + ((slot ^ swz_key(row)) << 4));
    }
    f32x4_t acc2[P / 16][NT];
#pragma unroll
    for (int i = 0; i < P / 16; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[i][t] = f32x4_t{0};
    spx8_t t2h[NQ][NT], t2l[NQ][NT];  // T2 as the B operand of conv3 (filled behind phase A)
    if constexpr (T11) {
        // ---- the halo patch: slot sl = py * 13 + px <-> image position (y0 - 1 + py, x0 - 1 + px); one plane per 32-channel chunk
        constexpr int PIECES = T11_SLOTS / 8;  // 22 DMA pieces of 8 slots per plane
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < (PIECES + 3) / 4; ++i) {
                const int ii = wave + 4 * i;
                const int sl = ii * 8 + lrow8;
                const int py = sl / T11_PW, px = sl - py * T11_PW;
                const int y = t11_y0 - 1 + py, x = t11_x0 - 1 + px;
                const bool ok = sl < T11_PW * T11_PW && (unsigned)y < 55u && (unsigned)x < 55u;
                // chunk swizzle keyed by py * 11 + px, not by the slot py * 13 + px: see the fragment reads below
                const unsigned off = ok ? (unsigned)(((t11_b * 55 + y) * 55 + x) * (P * 4) + q * ROWB + ((slot ^ swz_key(py * T11_E + px)) << 4)) : OOB;
                if (ii < PIECES) dma16(t1rs, smem + q * (T11_SLOTS * ROWB) + ii * 1024, off);
            }
        // ---- conv2, CHANNEL-split across the waves: wave w owns output-channel tile w (16 stored rows of W2) for ALL 128 rows of
        // the block.  Its weight fragments (fragment-order copy: tile i, K-step ks, hi | lo = 1 KiB at ((i * 18 + ks) * 2 + hl) *
        // 1024, lane * 16 inside) go straight into a four-slot register ring three K-steps ahead -- every byte of W2 enters the CU
        // once per block (the first spatial-tile form had each wave own 32 positions x all 64 channels: every wave loaded all of
        // W2, 588 KB per block through the L2 -> CU path that bounds this kernel, and was no faster for it).  Loaded by inline
        // asm: left to itself hipcc sinks every load next to its use and the loop waits an L2 round trip per step; the counted
        // waits name the registers they release, nothing else of this wave is in flight here (the patch DMA was drained at the
        // barrier).  tools/audit_asm_loads.py walks this kernel's ISA too.  The activation fragments of all eight position
        // tiles come from the patch (a tap = a constant slot offset).
        constexpr int NKS = 9 * NQ, RING = 4, NPT = BM / 16;
        const unsigned lane16 = (unsigned)lane * 16u;
        const char* w2f = p.W2F + (size_t)wave * (NKS * 2048);
        u32x4_t wq[RING][2];
#define AVCER_T11_LOAD(KS)                                                                                              \
    do {                                                                                                                \
        const char* a_ = w2f + (size_t)(KS) * 2048;                                                                     \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wq[(KS) % RING][0]) : "v"(lane16), "s"(a_) : "memory");    \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(wq[(KS) % RING][1]) : "v"(lane16), "s"(a_) : "memory"); \
    } while (0)
#define AVCER_T11_WAIT(N, KS) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wq[(KS) % RING][0]), "+v"(wq[(KS) % RING][1]) :: "memory")
        static_assert(P == 64, "four waves, four channel tiles");
        int sb8[NPT];  // patch slot of this lane's position in each of the block's eight position tiles, at tap (0, 0)
        // Swizzle index of the same position (round 6).  The 16 lanes of a fragment read hold 16 consecutive tile positions r; their
        // SLOTS py * 13 + px jump by 2 at the end of a tile row, so they span 18 slot numbers and, keyed by the slot, two lane
        // pairs landed on the same banks: 39 % of this form's LDS cycles were conflicts (profiles/experiments/README.md, "SQ
        // counters of the chains").  Keyed by py * 11 + px = r + (a constant per tap) the 16 lanes are 16 CONSECUTIVE indices for
        // every tap, the case the searched key serves without conflicts; the slot's own parity still alternates lane by lane.
        int sq8[NPT];
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int r = min(t * 16 + l15, T11_E * T11_E - 1);
            const int ry = r / T11_E, rx = r - ry * T11_E;
            sb8[t] = (ry + 1) * T11_PW + rx + 1;
            sq8[t] = (ry + 1) * T11_E + rx + 1;
        }
        f32x4_t acc2c[NPT];
#pragma unroll
        for (int t = 0; t < NPT; ++t) acc2c[t] = f32x4_t{0};
        __syncthreads();  // the patch has landed (hipcc drains the DMA in front of the barrier) and the bias table is written
        AVCER_T11_LOAD(0);
        AVCER_T11_LOAD(1);
        AVCER_T11_LOAD(2);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 3 < NKS) { AVCER_T11_LOAD(ks + 3); }
            const int tap = ks / NQ, q = ks % NQ;
            const int toff = (tap / 3 - 1) * T11_PW + (tap % 3 - 1), toffq = (tap / 3 - 1) * T11_E + (tap % 3 - 1);
            const char* plane = smem + q * (T11_SLOTS * ROWB);
            // the fragments of step ks are in their registers; the loads of up to three later steps may fly
            if (ks + 3 < NKS) AVCER_T11_WAIT(6, ks);
            else if (ks + 2 < NKS) AVCER_T11_WAIT(4, ks);
            else if (ks + 1 < NKS) AVCER_T11_WAIT(2, ks);
            else AVCER_T11_WAIT(0, ks);
            __builtin_amdgcn_sched_barrier(0);
            const spx8_t wh = __builtin_bit_cast(spx8_t, wq[ks % RING][0]), wl = __builtin_bit_cast(spx8_t, wq[ks % RING][1]);
#pragma unroll
            for (int t = 0; t < NPT; ++t) {
                const char* row_ = plane + (sb8[t] + toff) * ROWB;
                const int key_ = swz_key(sq8[t] + toffq);
                const spx8_t ah = *reinterpret_cast<const spx8_t*>(row_ + ((g ^ key_) << 4));
                const spx8_t al = *reinterpret_cast<const spx8_t*>(row_ + (((4 + g) ^ key_) << 4));
                mfma3(acc2c[t], wh, wl, ah, al);
            }
        }
#undef AVCER_T11_LOAD
#undef AVCER_T11_WAIT
        pin(acc2c);
        __syncthreads();  // every wave is done with the patch
        // ---- T2 = relu(bn2(.)) of this wave's 16 channels x 128 rows -> LDS as sp32 rows (two 32-channel planes of 128 rows, the
        // fragment layout and swizzle of every other tile here); each wave then reads the B fragments of ITS 32 rows back.
        // Stored row 16 t + 4 g + r of a 32-channel group is channel 8 g + 4 t + r: lane group g of tile (wave & 1) holds
        // channels 8 g + 4 (wave & 1) .. + 3 of group wave >> 1 -- half of the 16-byte piece lane group g reads back.
        {
            const int qw = wave >> 1, half = wave & 1;
            const float4 b4 = *reinterpret_cast<const float4*>(sbias + 32 * qw + 8 * g + 4 * half);
            char* t2p = smem + qw * (BM * ROWB);
#pragma unroll
            for (int t = 0; t < NPT; ++t) {
                const f32x4_t a4 = acc2c[t];
                const float x0 = sp_value(relu_nan(__builtin_fmaf(a4[0], s2, b4.x))), x1 = sp_value(relu_nan(__builtin_fmaf(a4[1], s2, b4.y)));
                const float x2 = sp_value(relu_nan(__builtin_fmaf(a4[2], s2, b4.z))), x3 = sp_value(relu_nan(__builtin_fmaf(a4[3], s2, b4.w)));
                sp_flag(ovm, sp_max2(sp_max2(0.f, x0, x1), x2, x3));
                typedef __attribute__((ext_vector_type(4))) spe_t spx4_t;
                spx4_t h4, l4;
                h4[0] = (spe_t)x0; h4[1] = (spe_t)x1; h4[2] = (spe_t)x2; h4[3] = (spe_t)x3;
                l4[0] = (spe_t)(x0 - (float)h4[0]); l4[1] = (spe_t)(x1 - (float)h4[1]);
                l4[2] = (spe_t)(x2 - (float)h4[2]); l4[3] = (spe_t)(x3 - (float)h4[3]);
                const int row = t * 16 + l15;
                *reinterpret_cast<spx4_t*>(t2p + swz(row, g) + 8 * half) = h4;
                *reinterpret_cast<spx4_t*>(t2p + swz(row, 4 + g) + 8 * half) = l4;
            }
        }
        __syncthreads();  // T2 is complete
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int row = wave * (BM / 4) + t * 16 + l15;
                t2h[q][t] = ldfrag(smem + q * (BM * ROWB), row, g);
                t2l[q][t] = ldfrag(smem + q * (BM * ROWB), row, 4 + g);
            }
        __syncthreads();  // ... and read: the streaming phase's weight tiles overwrite it
    } else if constexpr (PATCH) {
        // virtual rows: image b, row y (-1 .. H) -> b * (H + 2) + y + 1; a virtual row has W + 2 slots (x = -1 .. W)
        const int PW = p.Wd + 2, VH = p.H + 2;
        const int m_last = min(m_base + BM, p.M) - 1;
        const int b0 = m_base / (p.H * p.Wd), y0 = (m_base / p.Wd) % p.H;
        const int b1 = m_last / (p.H * p.Wd), y1 = (m_last / p.Wd) % p.H;
        const int vr0 = b0 * VH + y0;                       // one virtual row above the first position's row
        const int nslots = (b1 * VH + y1 + 3 - vr0) * PW;   // ... through one below the last position's row (<= PSLOTS, checked on the host)
        const int ninstr = (nslots + 7) >> 3;
        constexpr int PI = (PSLOTS / 8 + 3) / 4;            // patch DMA instructions per wave (instruction ii = wave + 4 i)
        unsigned p_off[PI];
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int sl = (wave + 4 * i) * 8 + lrow8;
            const int vr = vr0 + sl / PW, px = sl - (sl / PW) * PW;
            const int b = vr / VH, yy = vr - b * VH - 1, xx = px - 1;
            const bool ok = sl < nslots && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.Wd;
            p_off[i] = ok ? (unsigned)(((long)(b * p.H + yy) * p.Wd + xx) * (P * 4) + ((slot ^ swz_key(sl)) << 4)) : OOB;
        }
        // slot of this lane's positions at tap (0, 0): position m -> (virtual row - vr0) * PW + x + 1
        int sb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int m = min(m_base + wave * (BM / 4) + t * 16 + l15, p.M - 1);
            const int x = m % p.Wd, yb = m / p.Wd;           // yb = b * H + y
            const int b = yb / p.H, y = yb - b * p.H;
            sb[t] = (b * VH + y + 1 - vr0) * PW + x + 1;
        }
        char* patch = smem;
        char* wt0 = smem + PSLOTS * ROWB;
        for (int c = 0; c < NQ; ++c) {
            // the previous chunk's last barrier freed the patch and both weight tiles
#pragma unroll
            for (int i = 0; i < PI; ++i)
                if (wave + 4 * i < ninstr) dma16(t1rs, patch + (wave + 4 * i) * 1024, p_off[i], (unsigned)(c * ROWB));
#pragma unroll
            for (int j = 0; j < W_ISS; ++j) dma16(w2rs, wt0 + wave * (W_ISS * 1024) + j * 1024, w_off[j], (unsigned)(c * ROWB));
            __syncthreads();
            int cur = 0;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) {
#pragma unroll
                    for (int j = 0; j < W_ISS; ++j)
                        dma16(w2rs, wt0 + (cur ^ 1) * (P * ROWB) + wave * (W_ISS * 1024) + j * 1024, w_off[j],
                              (unsigned)(((tap + 1) * P + 32 * c) * 4));
                }
                const int toff = (tap / 3 - 1) * PW + (tap % 3 - 1);
                const char* sbw = wt0 + cur * (P * ROWB);
                spx8_t ah[NT], al[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    ah[t] = ldfrag(patch, sb[t] + toff, g);
                    al[t] = ldfrag(patch, sb[t] + toff, 4 + g);
                }
#pragma unroll
                for (int i = 0; i < P / 16; ++i) {
                    const spx8_t wh = ldfrag(sbw, i * 16 + l15, g), wl = ldfrag(sbw, i * 16 + l15, 4 + g);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mfma3(acc2[i][t], wh, wl, ah[t], al[t]);
                }
                pin(acc2);
                __syncthreads();
                cur ^= 1;
            }
        }
    } else {
    unsigned a_off[A_ISS];
    int a_y[A_ISS], a_x[A_ISS];
#pragma unroll
    for (int j = 0; j < A_ISS; ++j) {
        const int row = wave * (A_ISS * 8) + j * 8 + lrow8;
        const int m = m_base + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        int x, y;
        long src = mm;  // T1 row of the centre tap
        if constexpr (SUB == 1) {
            x = mm % p.Wd;
            y = (mm / p.Wd) % p.H;
        } else {
            const int ox = mm % p.OW, yb = mm / p.OW;
            const int oy = yb % p.OH, b = yb / p.OH;
            x = SUB * ox;
            y = SUB * oy;
            src = ((long)b * p.H + y) * p.Wd + x;
        }
        a_y[j] = ok ? y : -(1 << 28);
        a_x[j] = x;
        a_off[j] = (unsigned)(src * (P * 4) + ((slot ^ swz_key(row)) << 4));
    }
    {
        int ky = 0, kx = 0, kq = 0;
        unsigned wk = 0;
        auto issue = [&](int buf) {
            char* sa = smem + buf * TILE_A + wave * (A_ISS * 1024);
            char* sb = smem + buf * TILE_A + BM * ROWB + wave * (W_ISS * 1024);
            const int dy = ky - 1, dx = kx - 1;
            const int tap = ((dy * p.Wd + dx) * P + kq * 32) * 4;  // byte offset of this tap / channel chunk (may be negative)
#pragma unroll
            for (int j = 0; j < A_ISS; ++j) {
                const bool ok = ((unsigned)(a_y[j] + dy) < (unsigned)p.H) & ((unsigned)(a_x[j] + dx) < (unsigned)p.Wd);
                dma16(t1rs, sa + j * 1024, ok ? a_off[j] + (unsigned)tap : OOB);
            }
#pragma unroll
            for (int j = 0; j < W_ISS; ++j) dma16(w2rs, sb + j * 1024, w_off[j], wk);
            wk += ROWB;
            if (++kq == NQ) { kq = 0; if (++kx == 3) { kx = 0; ++ky; } }
        };
        issue(0);
        __syncthreads();
        int cur = 0;
        constexpr int NK = 9 * NQ;
        for (int step = 0; step < NK; ++step) {
            if (step + 1 < NK) issue(cur ^ 1);
            const char* sa = smem + cur * TILE_A;
            const char* sb = sa + BM * ROWB;
            spx8_t ah[NT], al[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int row = wave * (BM / 4) + t * 16 + l15;
                ah[t] = ldfrag(sa, row, g);
                al[t] = ldfrag(sa, row, 4 + g);
            }
#pragma unroll
            for (int i = 0; i < P / 16; ++i) {
                const spx8_t wh = ldfrag(sb, i * 16 + l15, g), wl = ldfrag(sb, i * 16 + l15, 4 + g);
#pragma unroll
                for (int t = 0; t < NT; ++t) mfma3(acc2[i][t], wh, wl, ah[t], al[t]);
            }
            pin(acc2);
            __syncthreads();
            cur ^= 1;
        }
    }
    }

    // ---------------- weights of output group G -> LDS buffer (G & 1)
    unsigned g3_off[NQT], g1_off[W_ISS];  // W3: NQT tiles of 32 rows = NQT*4 instr per block -> NQT per wave; W1N: P/8 instr -> P/32 per wave
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
        // instruction j of this wave fills rows [8 wave, 8 wave + 8) of K-step tile j of W3's 32-row group
        const int row = wave * 8 + lrow8;
        g3_off[j] = (unsigned)((long)row * (NQT * ROWB) + j * ROWB + ((slot ^ swz_key(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < W_ISS; ++j) {
        const int row = wave * (W_ISS * 8) + j * 8 + lrow8;
        g1_off[j] = (unsigned)((long)row * (4 * P * 4) + ((slot ^ swz_key(row)) << 4));
    }
    auto issue_group = [&](int G) {
        char* base = smem + (G & 1) * TILE_B;
#pragma unroll
        for (int j = 0; j < NQT; ++j) dma16(w3rs, base + j * (32 * ROWB) + wave * 1024, g3_off[j], (unsigned)(G * 32 * NQT * ROWB));
        if constexpr (NEXT) {
#pragma unroll
            for (int j = 0; j < W_ISS; ++j)
                dma16(w1rs, base + NQT * (32 * ROWB) + wave * (W_ISS * 1024) + j * 1024, g1_off[j], (unsigned)(G * ROWB));
        }
    };
    issue_group(0);  // the tile buffers are free: phase A ended on a barrier

    // residual rows of this lane: position m_t = m_base + wave*BM/4 + 16 t + (lane & 15); 16 bytes hi + 16 bytes lo per group
    // 32-bit byte offsets (the launcher keeps every tensor of a pass under 4 GiB; two registers fewer per row than pointers)
    unsigned x_row[NT];  // byte offset of this lane's piece of the residual row (SUB == 1: and of the block-output row, same shape)
    unsigned o_row[SUB > 1 ? NT : 1];  // SUB > 1: byte offset of the piece of the (compact) output row
    int m_row[NT];  // position (clamped to 0 past M: loads stay in bounds, stores are predicated)
    bool m_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        m_row[t] = row_pos(wave * (BM / 4) + t * 16 + l15, m_ok[t]);
        if constexpr (SUB == 1) {
            x_row[t] = (unsigned)(m_row[t] * (4L * P * 4) + 16 * g);
        } else {
            const int mm = m_row[t];
            const int ox = mm % p.OW, yb = mm / p.OW;
            const int oy = yb % p.OH, b = yb / p.OH;
            x_row[t] = (unsigned)((((long)b * p.H + SUB * oy) * p.Wd + SUB * ox) * (4L * P * 4) + 16 * g);
            o_row[t] = (unsigned)(m_row[t] * (4L * P * 4) + 16 * g);
        }
    }
    // downsample operand: the NQX K-steps of this lane's positions as B fragments, straight from global memory
    spx8_t xh[NQX > 0 ? NQX : 1][NT], xl[NQX > 0 ? NQX : 1][NT];
    if constexpr (NQX > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const char* xp = p.X + (long)m_row[t] * (NQX * 128L) + 16 * g;
#pragma unroll
            for (int q = 0; q < NQX; ++q) {
                xh[q][t] = *reinterpret_cast<const spx8_t*>(xp + q * 128);
                xl[q][t] = *reinterpret_cast<const spx8_t*>(xp + q * 128 + 64);
            }
        }
    }
    // residual of group G sits in ring slot G & 1 and is requested two groups ahead of its use
    uint4 rh[2][NT], rl[2][NT];
    auto load_res = [&](int G, uint4 (&h)[NT], uint4 (&l)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (NQX > 0) {
                h[t] = make_uint4(0u, 0u, 0u, 0u);  // no residual: the downsample branch is part of the contraction
                l[t] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                const char* rp = p.X + (size_t)x_row[t] + G * 128;
                h[t] = *reinterpret_cast<const uint4*>(rp);
                l[t] = *reinterpret_cast<const uint4*>(rp + 64);
            }
        }
    };
    load_res(0, rh[0], rl[0]);
    load_res(1, rh[1], rl[1]);

    // T2 as B-operand fragments: K-step q = channels 32q..32q+31, lane group g holds 8g..8g+7 (weight rows were permuted)
    // (the spatial-tile form has read them back from LDS already)
#pragma unroll
    for (int q = 0; q < (T11 ? 0 : NQ); ++q) {
        const float4 b0 = *reinterpret_cast<const float4*>(sbias + 32 * q + 8 * g), b1 = *reinterpret_cast<const float4*>(sbias + 32 * q + 8 * g + 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4_t lo4 = acc2[2 * q][t], hi4 = acc2[2 * q + 1][t];
            const float v[8] = {relu_nan(__builtin_fmaf(lo4[0], s2, b0.x)), relu_nan(__builtin_fmaf(lo4[1], s2, b0.y)),
                                relu_nan(__builtin_fmaf(lo4[2], s2, b0.z)), relu_nan(__builtin_fmaf(lo4[3], s2, b0.w)),
                                relu_nan(__builtin_fmaf(hi4[0], s2, b1.x)), relu_nan(__builtin_fmaf(hi4[1], s2, b1.y)),
                                relu_nan(__builtin_fmaf(hi4[2], s2, b1.z)), relu_nan(__builtin_fmaf(hi4[3], s2, b1.w))};
            split8v(v, t2h[q][t], t2l[q][t], ovm);
        }
    }
    f32x4_t acc1[NEXT ? P / 16 : 1][NT];
#pragma unroll
    for (int i = 0; i < (NEXT ? P / 16 : 1); ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc1[i][t] = f32x4_t{0};
    __syncthreads();  // group 0 has landed

    // ---------------- phases B + C per 32-channel output group (two groups per trip: the residual ring is indexed statically)
    auto group = [&](int G, uint4 (&h)[NT], uint4 (&l)[NT]) {
        // (a burst: spreading these pieces over the conv3 MFMAs, as conv_gemm_wd_kernel and bneck_tail2_kernel do, measured
        // +2.4 % at planes 64 and +0.6 % at planes 128 -- the group's barrier drains them, and they land later)
        if (G + 1 < NG) issue_group(G + 1);
        const char* w3t = smem + (G & 1) * TILE_B;
        const char* w1t = w3t + NQT * (32 * ROWB);
        f32x4_t acc3[2][NT];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc3[tp][t] = f32x4_t{0};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const spx8_t wh = ldfrag(w3t + q * (32 * ROWB), tp * 16 + l15, g), wl = ldfrag(w3t + q * (32 * ROWB), tp * 16 + l15, 4 + g);
#pragma unroll
                for (int t = 0; t < NT; ++t) mfma3(acc3[tp][t], wh, wl, t2h[q][t], t2l[q][t]);
            }
        if constexpr (NQX > 0) {
#pragma unroll
            for (int q = 0; q < NQX; ++q)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const char* wt = w3t + (NQ + q) * (32 * ROWB);
                    const spx8_t wh = ldfrag(wt, tp * 16 + l15, g), wl = ldfrag(wt, tp * 16 + l15, 4 + g);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mfma3(acc3[tp][t], wh, wl, xh[q][t], xl[q][t]);
                }
        }
        const float* bp = sbias + 2 * P + 32 * G + 8 * g;
        const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
        spx8_t oh[NT], ol[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float r[8];
            unpack8(h[t], l[t], r);
            const f32x4_t lo4 = acc3[0][t], hi4 = acc3[1][t];
            const float v[8] = {relu_nan(__builtin_fmaf(lo4[0], s3, b0.x) + r[0]), relu_nan(__builtin_fmaf(lo4[1], s3, b0.y) + r[1]),
                                relu_nan(__builtin_fmaf(lo4[2], s3, b0.z) + r[2]), relu_nan(__builtin_fmaf(lo4[3], s3, b0.w) + r[3]),
                                relu_nan(__builtin_fmaf(hi4[0], s3, b1.x) + r[4]), relu_nan(__builtin_fmaf(hi4[1], s3, b1.y) + r[5]),
                                relu_nan(__builtin_fmaf(hi4[2], s3, b1.z) + r[6]), relu_nan(__builtin_fmaf(hi4[3], s3, b1.w) + r[7])};
            split8v(v, oh[t], ol[t], ovm);
            if (m_ok[t]) {
                char* yp = p.OUT + (size_t)(SUB == 1 ? x_row[t] : o_row[SUB > 1 ? t : 0]) + G * 128;
                *reinterpret_cast<spx8_t*>(yp) = oh[t];
                *reinterpret_cast<spx8_t*>(yp + 64) = ol[t];
            }
        }
        if (G + 2 < NG) load_res(G + 2, h, l);  // the slot just consumed
        if constexpr (NEXT) {
#pragma unroll
            for (int i = 0; i < P / 16; ++i) {
                const spx8_t wh = ldfrag(w1t, i * 16 + l15, g), wl = ldfrag(w1t, i * 16 + l15, 4 + g);
#pragma unroll
                for (int t = 0; t < NT; ++t) mfma3(acc1[i][t], wh, wl, oh[t], ol[t]);
            }
            pin(acc1);
        }
        __syncthreads();
    };
    for (int G = 0; G < NG; G += 2) {
        group(G, rh[0], rl[0]);
        group(G + 1, rh[1], rl[1]);
    }

    // ---------------- T1' = relu(bn1'(conv1'(OUT)))
    if constexpr (NEXT) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 b0 = *reinterpret_cast<const float4*>(sbias + P + 32 * q + 8 * g), b1 = *reinterpret_cast<const float4*>(sbias + P + 32 * q + 8 * g + 4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4_t lo4 = acc1[2 * q][t], hi4 = acc1[2 * q + 1][t];
                const float v[8] = {relu_nan(__builtin_fmaf(lo4[0], s1n, b0.x)), relu_nan(__builtin_fmaf(lo4[1], s1n, b0.y)),
                                    relu_nan(__builtin_fmaf(lo4[2], s1n, b0.z)), relu_nan(__builtin_fmaf(lo4[3], s1n, b0.w)),
                                    relu_nan(__builtin_fmaf(hi4[0], s1n, b1.x)), relu_nan(__builtin_fmaf(hi4[1], s1n, b1.y)),
                                    relu_nan(__builtin_fmaf(hi4[2], s1n, b1.z)), relu_nan(__builtin_fmaf(hi4[3], s1n, b1.w))};
                spx8_t hi, lo;
                split8v(v, hi, lo, ovm);
                if (m_ok[t]) {
                    char* yp = p.T1N + (long)m_row[t] * (P * 4) + q * 128 + 16 * g;  // m_row == the position wherever m_ok
                    *reinterpret_cast<spx8_t*>(yp) = hi;
                    *reinterpret_cast<spx8_t*>(yp + 64) = lo;
                }
            }
        }
    }
    sp_commit(p.ovf, ovm);
}


// ------------------------------------------------------------------------------------------------ bottleneck tail (wide stages)
// Stage 3 (planes 256): the 3x3 convolution stays a plain conv_gemm launch (a wave cannot hold 256 x BM accumulators next
// to the fragments), but conv3 + residual + ReLU and the NEXT block's conv1 still share one launch: T2 is read from
// global memory straight into B fragments (an sp32 row is the fragment layout), OUT is written once and never re-read
// by a conv1 launch.  128 positions per block (16 per wave), weights of one 32-channel output group at a time.
//
// A first form of this kernel (two 32 KiB LDS halves, two __syncthreads() per group) is archived with its bit-identity
// evidence in profiles/experiments/r02_bneck_tail_first_form.hip.txt: its 228 VGPRs leave room for ONE 8-wave block per
// CU, so nothing else on the CU hides a stall, and each of its barriers drains vmcnt to zero -- including the residual
// loads just issued for two groups ahead, i.e. every group pays a whole HBM round trip.  Here the 133 KiB that one
// block may use hold BOTH weight tiles twice: the DMA of group G+1 (W3 group and W1N K-step) is issued at the top of group G,
// one raw s_barrier ends a group, and the wait in front of it is counted -- s_waitcnt vmcnt(4) leaves this group's two
// stores and two residual loads in flight.  Every vector-memory instruction of the loop is issued unconditionally (rows
// past M store through a buffer descriptor with an out-of-range offset, which the hardware drops), so the count is exact.
template <int P>
__global__ void __launch_bounds__(512, 2) bneck_tail2_kernel(const BneckParams p) {
    constexpr int NW = 8;
    constexpr int BM = 16 * NW;
    constexpr int NQ = P / 32;      // K-steps of conv3
    constexpr int NG = 4 * P / 32;  // 32-channel groups of the block output = K-steps of conv1'
    constexpr int W3B = 32 * NQ * ROWB, W1B = P * ROWB;
    constexpr int NBIAS = 5 * P;    // b1n [P], b3 [4P]
    __shared__ __attribute__((aligned(16))) char smem[2 * W3B + 2 * W1B + NBIAS * 4];
    float* sbias = reinterpret_cast<float*>(smem + 2 * W3B + 2 * W1B);
    for (int i = threadIdx.x; i < NBIAS; i += 64 * NW) sbias[i] = i < P ? p.b1n[i] : p.b3[i - P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int lrow8 = lane >> 3, slot = lane & 7;
    const int m_base = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const float s3 = split_wmul(p.W3, 4 * P * P * 4), s1n = split_wmul(p.W1N, 4 * P * P * 4);  // split_dev.h: weight trailers
    const auto w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W3), (short)0, 4 * P * P * 4, 0x00020000);
    const auto w1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.W1N), (short)0, 4 * P * P * 4, 0x00020000);
    const auto outrs = __builtin_amdgcn_make_buffer_rsrc(p.OUT, (short)0, (int)p.t1_bytes, 0x00020000);  // t1_bytes: bytes of OUT here
    constexpr int G3_ISS = NQ * 4 / NW, G1_ISS = P / 8 / NW;
    unsigned g3_off[G3_ISS], g1_off[G1_ISS];
    int g3_dst[G3_ISS];
#pragma unroll
    for (int j = 0; j < G3_ISS; ++j) {
        const int ii = wave * G3_ISS + j;
        const int row = (ii & 3) * 8 + lrow8;
        g3_off[j] = (unsigned)((long)row * (P * 4) + (ii >> 2) * ROWB + ((slot ^ swz_key(row)) << 4));
        g3_dst[j] = (ii >> 2) * (32 * ROWB) + (ii & 3) * 1024;
    }
#pragma unroll
    for (int j = 0; j < G1_ISS; ++j) {
        const int row = (wave * G1_ISS + j) * 8 + lrow8;
        g1_off[j] = (unsigned)((long)row * (4 * P * 4) + ((slot ^ swz_key(row)) << 4));
    }
    // both tiles of group G into the buffers of parity G & 1: G3_ISS + G1_ISS vector-memory operations.  G == NG (issued by the
    // last group so that every group issues the same operations): out-of-range offsets, zeros into the idle buffers, no fetch
    auto issue_weights = [&](int G) {
        char* b3 = smem + (G & 1) * W3B;
        char* b1 = smem + 2 * W3B + (G & 1) * W1B;
        const bool live = G < NG;
#pragma unroll
        for (int j = 0; j < G3_ISS; ++j) dma16(w3rs, b3 + g3_dst[j], live ? g3_off[j] : OOB, (unsigned)(G * 32 * P * 4));
#pragma unroll
        for (int j = 0; j < G1_ISS; ++j) dma16(w1rs, b1 + (wave * G1_ISS + j) * 1024, live ? g1_off[j] : OOB, (unsigned)(G * ROWB));
    };
    // one of the G3_ISS + G1_ISS pieces of group G (round 4: inside a group the pieces are issued one by one between the
    // conv3 MFMAs instead of as a burst in front of them -- what paid in conv_gemm_wd_kernel, gemm.hip AVCER_WD_STEP)
    static_assert(G3_ISS + G1_ISS == NQ, "one piece per conv3 K-step");
    auto issue_piece = [&](int G, int j) {
        const bool live = G < NG;
        if (j < G3_ISS) dma16(w3rs, smem + (G & 1) * W3B + g3_dst[j], live ? g3_off[j] : OOB, (unsigned)(G * 32 * P * 4));
        else dma16(w1rs, smem + 2 * W3B + (G & 1) * W1B + (wave * G1_ISS + (j - G3_ISS)) * 1024, live ? g1_off[j - G3_ISS] : OOB,
                   (unsigned)(G * ROWB));
    };
    issue_weights(0);

    const long m = (long)m_base + wave * 16 + l15;
    const bool m_ok = m < p.M;
    const long mc = m_ok ? m : 0;
    const long x_row = mc * (4L * P * 4) + 16 * g;
    const unsigned o_row = m_ok ? (unsigned)x_row : OOB;  // rows past M: the buffer store is dropped
    spx8_t t2h[NQ], t2l[NQ];
    {
        const char* tp = p.T1 + mc * (P * 4L) + 16 * g;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            t2h[q] = *reinterpret_cast<const spx8_t*>(tp + q * 128);
            t2l[q] = *reinterpret_cast<const spx8_t*>(tp + q * 128 + 64);
        }
    }
    // Residual ring, two groups deep.  The loads are inline asm so that hipcc does not wait for them itself (inside the loop
    // it can only drain to zero, DMA of the next group included); the counted wait that ends the previous group is what
    // guarantees them, and it names the registers so that no use can be scheduled in front of it.
    u32x4_t rh0, rl0, rh1, rl1;
    const char* xrow_p = p.X + x_row;
#define AVCER_LOAD_RES(G, H, L)                                                                              \
    do {                                                                                                     \
        const char* rp_ = xrow_p + (G) * 128;                                                                \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(H) : "v"(rp_) : "memory");                     \
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(L) : "v"(rp_) : "memory");           \
    } while (0)
    AVCER_LOAD_RES(0, rh0, rl0);
    AVCER_LOAD_RES(1, rh1, rl1);
    f32x4_t acc1[P / 16];
#pragma unroll
    for (int i = 0; i < P / 16; ++i) acc1[i] = f32x4_t{0};
    sp_flags_t ovm = 0;  // lanes that split a finite |x| >= 65520 into an fp16 pair (split_dev.h sp_commit)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rh0), "+v"(rl0), "+v"(rh1), "+v"(rl1)::"memory");
    __syncthreads();  // weights of group 0 and the bias table are in LDS (this one drains everything, once)

    // One 32-channel output group.  (H, L): residual of this group, reloaded for group G+2 once consumed; (HN, LN): residual
    // of the next group, which the wait at the end guarantees.  Vector-memory operations of a wave, in order:
    // [G3_ISS + G1_ISS DMA of group G+1] [2 stores] [2 residual loads of group G+2].
#define AVCER_TAIL_GROUP(G, H, L, HN, LN)                                                                                      \
    do {                                                                                                                       \
        const char* w3t = smem + ((G) & 1) * W3B;                                                                              \
        const char* w1t = smem + 2 * W3B + ((G) & 1) * W1B;                                                                    \
        f32x4_t acc3[2] = {f32x4_t{0}, f32x4_t{0}};                                                                            \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int tp = 0; tp < 2; ++tp) {                      \
            if (tp == 0) { /* piece q of group G+1 -> the buffers of the other parity (every wave left group G-1 at the */     \
                           /* barrier); all eight are issued before this group's stores and residual loads: the count holds */ \
                __builtin_amdgcn_sched_barrier(0);                                                                             \
                issue_piece((G) + 1, q);                                                                                       \
                asm volatile("" ::: "memory");                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                             \
            }                                                                                                                  \
            const char* wt = w3t + q * (32 * ROWB);                                                                            \
            mfma3(acc3[tp], ldfrag(wt, tp * 16 + l15, g), ldfrag(wt, tp * 16 + l15, 4 + g), t2h[q], t2l[q]);                   \
        }                                                                                                                      \
        asm volatile("" ::: "memory");            /* nothing below may be hoisted above the DMA: the count relies on it */       \
        f32x4_t b0, b1;                                                                                                        \
        lds_read8(sbias + P + 32 * (G) + 8 * g, b0, b1);                                                                       \
        float r[8];                                                                                                            \
        unpack8(__builtin_bit_cast(uint4, H), __builtin_bit_cast(uint4, L), r);                                                \
        const float v[8] = {relu_nan(__builtin_fmaf(acc3[0][0], s3, b0[0]) + r[0]), relu_nan(__builtin_fmaf(acc3[0][1], s3, b0[1]) + r[1]), \
                            relu_nan(__builtin_fmaf(acc3[0][2], s3, b0[2]) + r[2]), relu_nan(__builtin_fmaf(acc3[0][3], s3, b0[3]) + r[3]), \
                            relu_nan(__builtin_fmaf(acc3[1][0], s3, b1[0]) + r[4]), relu_nan(__builtin_fmaf(acc3[1][1], s3, b1[1]) + r[5]), \
                            relu_nan(__builtin_fmaf(acc3[1][2], s3, b1[2]) + r[6]), relu_nan(__builtin_fmaf(acc3[1][3], s3, b1[3]) + r[7])}; \
        spx8_t oh, ol;                                                                                                       \
        split8v(v, oh, ol, ovm);                                                                                                    \
        /* (oh / ol stay live as the next contraction's operand.  A 16-byte buffer store with its offset in an SGPR whose data  */ \
        /* registers die here would be a hazard on gfx950 that hipcc does not separate: tests/test_build_hygiene.py scans for it) */ \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oh), outrs, o_row, (G) * 128, 0);                   \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ol), outrs, o_row, (G) * 128 + 64, 0);              \
        asm volatile("" ::: "memory");                                                                                         \
        AVCER_LOAD_RES((G) + 2 < NG ? (G) + 2 : NG - 1, H, L); /* unconditional (the last two re-fetch group NG-1) */          \
        _Pragma("unroll") for (int i = 0; i < P / 16; ++i)                                                                     \
            mfma3(acc1[i], ldfrag(w1t, i * 16 + l15, g), ldfrag(w1t, i * 16 + l15, 4 + g), oh, ol);                            \
        pin(acc1);                                                                                                             \
        /* the weight DMA of group G+1 and the residual of group G+1 have landed; still in flight: this group's two stores */  \
        /* and its two residual loads.  Unconditional, the last group included (its DMA fetched nothing): every trip */         \
        /* through the loop issues the same operations, which is what tools/audit_asm_loads.py counts */                       \
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(HN), "+v"(LN)::"memory");                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
        __builtin_amdgcn_s_barrier();                                                                                          \
        asm volatile("" ::: "memory");                                                                                         \
    } while (0)
    for (int G = 0; G < NG; G += 2) {
        AVCER_TAIL_GROUP(G, rh0, rl0, rh1, rl1);
        AVCER_TAIL_GROUP(G + 1, rh1, rl1, rh0, rl0);
    }
#undef AVCER_TAIL_GROUP
#undef AVCER_LOAD_RES
    // The last two groups re-fetched group NG-1 into the ring (the count needs every group to issue the same operations) and
    // nothing consumed those loads: wait for them HERE, naming the registers, so that they stay allocated until the data
    // has landed -- hipcc does not know an asm load is asynchronous and could otherwise hand the registers to the epilogue.
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rh0), "+v"(rl0), "+v"(rh1), "+v"(rl1)::"memory");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 b0 = *reinterpret_cast<const float4*>(sbias + 32 * q + 8 * g), b1 = *reinterpret_cast<const float4*>(sbias + 32 * q + 8 * g + 4);
        const f32x4_t lo4 = acc1[2 * q], hi4 = acc1[2 * q + 1];
        const float v[8] = {relu_nan(__builtin_fmaf(lo4[0], s1n, b0.x)), relu_nan(__builtin_fmaf(lo4[1], s1n, b0.y)),
                            relu_nan(__builtin_fmaf(lo4[2], s1n, b0.z)), relu_nan(__builtin_fmaf(lo4[3], s1n, b0.w)),
                            relu_nan(__builtin_fmaf(hi4[0], s1n, b1.x)), relu_nan(__builtin_fmaf(hi4[1], s1n, b1.y)),
                            relu_nan(__builtin_fmaf(hi4[2], s1n, b1.z)), relu_nan(__builtin_fmaf(hi4[3], s1n, b1.w))};
        spx8_t hi, lo;
        split8v(v, hi, lo, ovm);
        if (m_ok) {
            char* yp = p.T1N + m * (P * 4L) + q * 128 + 16 * g;
            *reinterpret_cast<spx8_t*>(yp) = hi;
            *reinterpret_cast<spx8_t*>(yp + 64) = lo;
        }
    }
    sp_commit(p.ovf, ovm);
}

// ------------------------------------------------------------------------------------------------ on-box ceilings
// Two micro-kernels that bench.py runs once to put MEASURED ceilings of this very GPU next to the guide's peaks:
// back-to-back v_mfma_f32_16x16x32_bf16 on register operands (two waves per SIMD, 16 independent accumulators each,
// non-trivial operand bits), and a 16-byte-per-lane streaming copy.
__global__ void __launch_bounds__(256, 2) mfma_rate_kernel(float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    spx8_t a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (spe_t)(0.25f + 0.001f * (float)((lane * 8 + j) % 97));
        b[j] = (spe_t)(-0.5f + 0.002f * (float)((lane * 5 + j * 3 + blockIdx.x) % 89));
    }
    f32x4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = mfma_sp(a, b, acc[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;  // keeps the chain alive without a store on the measured path
}

__global__ void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    // ONE 16-byte load and store per thread, one thread per element: measured 6.19 TB/s on this pool against 5.0-5.7 with
    // 2-16 accesses in flight per thread, 4.4-4.8 for persistent grid-stride forms and 4.8 for hipMemcpyAsync
    // (tools/copy_sweep.hip, profiles/r03_copy_sweep.txt): wave-level parallelism, not per-thread unrolling, feeds HBM here
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
}  // namespace

// ------------------------------------------------------------------------------------------------ launchers
int launch_stem_pool(avcer_ctx* ctx, const void* planes, size_t plane_bytes, const void* w_x3, const float* scale,
                     const float* bias, void* y, int n, hipStream_t st) {
    if (!planes || !w_x3 || !scale || !bias || !y || n <= 0) return set_err(ctx, AVCER_EINVAL, "stem_pool: bad arguments");
    if (2 * plane_bytes >= (size_t)OOB) return set_err(ctx, AVCER_EINVAL, "stem_pool: %d frames exceed the 4 GiB descriptor", n);
    StemParams p;
    memset(&p, 0, sizeof(p));
    p.P = (const char*)planes; p.plane_bytes = (unsigned)plane_bytes; p.p_bytes = (unsigned)(2 * plane_bytes);
    p.W = (const char*)w_x3; p.scale = scale; p.bias = bias; p.Y = (char*)y; p.n = n; p.ovf = ctx->ovf;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    TRY(prof_begin(ctx, st, &ev0, &ev1, FAM_STEM, 2.0 * n * 112.0 * 112.0 * 64 * 147, (double)n * (2.0 * 230 * 230 * 4 * 2 + 55.0 * 55 * 64 * 4)));
    stem_pool_kernel<<<dim3(n * ST_TY * ST_TX), dim3(256), 0, st>>>(p);
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "stem_pool launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += 2.0 * n * 112.0 * 112.0 * 64 * 147;
    return AVCER_OK;
}

// The same launch from the u8 frames themselves (stem_pool_u8_kernel): no preprocessing pass, two MFMAs per product.
int launch_stem_pool_u8(avcer_ctx* ctx, const uint8_t* frames, int in_h, int in_w, const void* w_x3, const float* scale,
                        const float* bias9, void* y, int n, hipStream_t st) {
    if (!frames || !w_x3 || !scale || !bias9 || !y || n <= 0 || in_h <= 0 || in_w <= 0)
        return set_err(ctx, AVCER_EINVAL, "stem_pool_u8: bad arguments");
    StemParams p;
    memset(&p, 0, sizeof(p));
    p.F = frames; p.in_h = in_h; p.in_w = in_w; p.bias9 = bias9;
    p.W = (const char*)w_x3; p.scale = scale; p.Y = (char*)y; p.n = n; p.ovf = ctx->ovf;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    TRY(prof_begin(ctx, st, &ev0, &ev1, FAM_STEM, 2.0 * n * 112.0 * 112.0 * 64 * 147, (double)n * ((double)in_h * in_w * 3 + 55.0 * 55 * 64 * 4)));
    stem_pool_u8_kernel<false><<<dim3(n * ST_TY * ST_TX), dim3(256), 0, st>>>(p);
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "stem_pool_u8 launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += 2.0 * n * 112.0 * 112.0 * 64 * 147;
    return AVCER_OK;
}

// The detector's stem (stem_pool_u8_kernel<true>): frames u8 [n][h][w][3] (BGR unless rgb) -> sp32 [n][mh][mw][64], mh = ceil(ceil(h/2)/2).
int launch_stem_pool_face(avcer_ctx* ctx, const uint8_t* frames, int h, int w, int rgb, const void* w_x3, const float* scale,
                          const float* bias, void* y, int n, hipStream_t st) {
    if (!frames || !w_x3 || !scale || !bias || !y || n <= 0 || h <= 0 || w <= 0) return set_err(ctx, AVCER_EINVAL, "stem_pool_face: bad arguments");
    StemParams p;
    memset(&p, 0, sizeof(p));
    p.F = frames; p.in_h = h; p.in_w = w; p.bias = bias;
    p.W = (const char*)w_x3; p.scale = scale; p.Y = (char*)y; p.n = n; p.ovf = ctx->ovf;
    p.oh = (h - 1) / 2 + 1; p.ow = (w - 1) / 2 + 1;
    p.mh = (p.oh - 1) / 2 + 1; p.mw = (p.ow - 1) / 2 + 1;
    p.tiles_y = (p.mh + ST_TH - 1) / ST_TH; p.tiles_x = (p.mw + ST_TW - 1) / ST_TW;
    p.swap_rb = rgb ? 1 : 0;
    p.mean[0] = 104; p.mean[1] = 117; p.mean[2] = 123;  // retina_face_predictor.py:59-65, in the net's (B, G, R) order
    const long grid = (long)n * p.tiles_y * p.tiles_x;
    if (grid >= (1L << 31)) return set_err(ctx, AVCER_EINVAL, "stem_pool_face: %d frames of %d x %d are too many tiles for one launch", n, h, w);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const double flops = 2.0 * n * (double)p.oh * p.ow * 64 * 147;
    TRY(prof_begin(ctx, st, &ev0, &ev1, FAM_STEM, flops, (double)n * ((double)h * w * 3 + (double)p.mh * p.mw * 64 * 4), (long)n * p.oh * p.ow, 64, 147));
    stem_pool_u8_kernel<true><<<dim3((unsigned)grid), dim3(256), 0, st>>>(p);
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "stem_pool_face launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += flops;
    return AVCER_OK;
}

int launch_bneck(avcer_ctx* ctx, int planes, int nb, int h, int w, const void* t1, const void* x, int ds_cin, int out_step,
                 void* out, void* t1n, const void* w2, const float* b2, const void* w3, const float* b3, const void* w1n,
                 const float* b1n, hipStream_t st, const void* w2_frags) {
    if (out_step != 1 && out_step != 2) return set_err(ctx, AVCER_EINVAL, "bneck: out_step %d (1 or 2)", out_step);
    if (out_step == 2 && (t1n || ds_cin)) return set_err(ctx, AVCER_EINVAL, "bneck: the strided form is the last block of a stage (no next conv1, no downsample)");
    const int oh = (h - 1) / out_step + 1, ow = (w - 1) / out_step + 1;
    const long M = (long)nb * oh * ow, M_in = (long)nb * h * w;
    if (!t1 || !x || !out || !w2 || !w3 || !b2 || !b3 || M <= 0)
        return set_err(ctx, AVCER_EINVAL, "bneck: bad arguments");
    if ((t1n != nullptr) != (w1n != nullptr) || (t1n && !b1n))
        return set_err(ctx, AVCER_EINVAL, "bneck: next-block conv1 needs weights, bias and an output");
    if (planes != 64 && planes != 128) return set_err(ctx, AVCER_EINVAL, "bneck: planes %d (64 or 128)", planes);
    if (ds_cin != 0 && !(ds_cin == 64 && planes == 64 && t1n))
        return set_err(ctx, AVCER_EINVAL, "bneck: the downsample form exists for planes 64 with a 64-channel input and a next conv1");
    // The kernel walks every tensor of the pass with 32-bit byte offsets (x_row / o_row) and T1 through a buffer descriptor: the
    // LARGEST tensor bounds the pass -- X [M_in][4 planes | ds_cin] and OUT [M][4 planes], four times T1 -- not T1 alone
    const long x_bytes = M_in * (long)(ds_cin ? ds_cin : 4 * planes) * 4L, out_bytes = M * 4L * planes * 4L;
    if (M_in * planes * 4L >= (long)OOB || x_bytes >= (long)OOB || out_bytes >= (long)OOB)
        return set_err(ctx, AVCER_EINVAL, "bneck: %ld positions in, %ld out: a tensor of the pass reaches the 4 GiB offset range "
                                          "(at most %ld positions per call at planes %d)", M_in, M, ((long)OOB - 1) / (16L * planes), planes);
    BneckParams p;
    p.T1 = (const char*)t1; p.X = (const char*)x; p.OUT = (char*)out; p.T1N = (char*)t1n;
    p.W2 = (const char*)w2; p.W2F = (const char*)w2_frags; p.W3 = (const char*)w3; p.W1N = (const char*)w1n;
    p.b2 = b2; p.b3 = b3; p.b1n = b1n;
    p.t1_bytes = (unsigned)(M_in * planes * 4);
    p.M = (int)M; p.H = h; p.Wd = w; p.OH = oh; p.OW = ow; p.ovf = ctx->ovf;
    constexpr int BM = 128;
    // The spatial-tile form (bneck_kernel<..., T11>): planes 64 with a next conv1 on 55 x 55 images (5 x 5 tiles of 11 x 11),
    // when the caller brought the fragment-order copy of the conv2 weights; one block per tile.
    const bool t11 = planes == 64 && h == 55 && w == 55 && out_step == 1 && t1n && w2_frags;
    p.nblocks = t11 ? nb * 25 : (int)((M + BM - 1) / BM);
    const int grid = p.nblocks;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    {
        // per position: T1 in (P), residual or downsample operand in (4P | ds_cin), OUT (4P) and T1' (P) out, 4 bytes per element
        const double bytes = 4.0 * ((double)M_in * planes + (double)M * (ds_cin ? ds_cin : 4 * planes) + (double)M * 4 * planes +
                                    (t1n ? (double)M * planes : 0.0));
        TRY(prof_begin(ctx, st, &ev0, &ev1, FAM_CHAIN, 2.0 * (double)M * planes * (planes * (9.0 + 4.0 + (t1n ? 4.0 : 0.0)) + 4.0 * ds_cin), bytes, M,
                       4 * planes, planes));
    }
    // Resident halo patch for the conv2 phase: planes 128 only (28x28: -5..7 % per launch).  At planes 64 (55x55) the patch
    // costs the third resident block (76 KiB of LDS) and measured +2..4 %, so that form keeps the per-tap gather.
    // Worst case of the patch: the image rows 128 consecutive positions can touch (+ 2 when they cross into the next
    // image) + 2 halo rows, W + 2 slots each, must fit the kernel's LDS image.
    const int rows_worst = (w - 1 + BM + w - 1) / w + 2 + 2;
    const bool patch = planes == 128 && (long)h * w >= BM && rows_worst * (w + 2) <= 304;
    if (out_step == 2) {
        if (planes == 64) bneck_kernel<64, BM, false, 0, false, 2><<<dim3(grid), dim3(256), 0, st>>>(p);
        else bneck_kernel<128, BM, false, 0, false, 2><<<dim3(grid), dim3(256), 0, st>>>(p);
    } else if (ds_cin) {
        if (t11) bneck_kernel<64, BM, true, 2, false, 1, true><<<dim3(grid), dim3(256), 0, st>>>(p);
        else bneck_kernel<64, BM, true, 2, false><<<dim3(grid), dim3(256), 0, st>>>(p);
    } else if (planes == 64) {
        if (t11) bneck_kernel<64, BM, true, 0, false, 1, true><<<dim3(grid), dim3(256), 0, st>>>(p);
        else if (t1n) bneck_kernel<64, BM, true, 0, false><<<dim3(grid), dim3(256), 0, st>>>(p);
        else bneck_kernel<64, BM, false, 0, false><<<dim3(grid), dim3(256), 0, st>>>(p);
    } else {
        if (t1n) {
            if (patch) bneck_kernel<128, BM, true, 0, true><<<dim3(grid), dim3(256), 0, st>>>(p);
            else bneck_kernel<128, BM, true, 0, false><<<dim3(grid), dim3(256), 0, st>>>(p);
        } else {
            if (patch) bneck_kernel<128, BM, false, 0, true><<<dim3(grid), dim3(256), 0, st>>>(p);
            else bneck_kernel<128, BM, false, 0, false><<<dim3(grid), dim3(256), 0, st>>>(p);
        }
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "bneck launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += 2.0 * (double)M * planes * (planes * (9.0 + 4.0 + (t1n ? 4.0 : 0.0)) + 4.0 * ds_cin);
    return AVCER_OK;
}

int launch_bneck_tail(avcer_ctx* ctx, int planes, long M, const void* t2, const void* x, void* out, void* t1n, const void* w3,
                      const float* b3, const void* w1n, const float* b1n, hipStream_t st) {
    if (!t2 || !x || !out || !t1n || !w3 || !b3 || !w1n || !b1n || M <= 0) return set_err(ctx, AVCER_EINVAL, "bneck_tail: bad arguments");
    if (planes != 256) return set_err(ctx, AVCER_EINVAL, "bneck_tail: planes %d (256)", planes);
    // OUT is stored through a buffer descriptor (rows past M are dropped by its bounds check): it must stay under 4 GiB,
    // i.e. 5 349 frames of 14 x 14 -- a back pass of the static CNN is at most 2 048
    if (M * 4096L >= (1L << 32) - 4096) return set_err(ctx, AVCER_EINVAL, "bneck_tail: M=%ld exceeds the 4 GiB descriptor range", M);
    BneckParams p;
    memset(&p, 0, sizeof(p));
    p.T1 = (const char*)t2; p.X = (const char*)x; p.OUT = (char*)out; p.T1N = (char*)t1n;
    p.W3 = (const char*)w3; p.W1N = (const char*)w1n; p.b3 = b3; p.b1n = b1n;
    p.M = (int)M; p.ovf = ctx->ovf;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per position: T2 in (P), residual in (4P), OUT (4P) and T1' (P) out
    TRY(prof_begin(ctx, st, &ev0, &ev1, FAM_TAIL, 2.0 * (double)M * planes * planes * 8.0, 4.0 * (double)M * planes * 10.0, M, 4 * planes, planes));
    p.t1_bytes = (unsigned)(M * 4096L);
    bneck_tail2_kernel<256><<<dim3((int)((M + 127) / 128)), dim3(512), 0, st>>>(p);
    if (ev1) (void)hipEventRecord(ev1, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "bneck_tail launch: %s", hipGetErrorString(e));
    ctx->gemm_launches += 1;
    ctx->gemm_flops += 2.0 * (double)M * planes * planes * 8.0;
    return AVCER_OK;
}

// Measured ceilings of this GPU: dense 16-bit MFMA issue rate (the f16 form; TFLOP/s) and streaming-copy bandwidth (TB/s, read + write bytes).
int measure_ceilings(avcer_ctx* ctx, double* mfma_bf16_tflops, double* hbm_copy_tbs, hipStream_t st) {
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    void* buf = nullptr;
    const size_t bytes = (size_t)1 << 30;  // 1 GiB source + 1 GiB destination: far beyond the 256 MiB memory-side cache
    TRY(ws_reserve(ctx, 5, 2 * bytes, &buf));
    float ms = 0.f;
    const int blocks = 256 * 2, iters = 20000;
    mfma_rate_kernel<<<blocks, 256, 0, st>>>((float*)buf, 2000);  // warm-up (clock ramp)
    HIP_TRY(ctx, hipEventRecord(e0, st));
    mfma_rate_kernel<<<blocks, 256, 0, st>>>((float*)buf, iters);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (mfma_bf16_tflops) *mfma_bf16_tflops = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
    HIP_TRY(ctx, hipMemsetAsync(buf, 1, bytes, st));
    const int cgrid = (int)(bytes / 16 / 256);
    copy16_kernel<<<cgrid, 256, 0, st>>>((const uint4*)buf, (uint4*)((char*)buf + bytes), bytes / 16);
    HIP_TRY(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < 4; ++i) copy16_kernel<<<cgrid, 256, 0, st>>>((const uint4*)buf, (uint4*)((char*)buf + bytes), bytes / 16);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (hbm_copy_tbs) *hbm_copy_tbs = 4.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, AVCER_EHIP, "measure_ceilings: %s", hipGetErrorString(e));
    return AVCER_OK;
}
