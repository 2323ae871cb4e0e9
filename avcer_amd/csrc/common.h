// Internal declarations shared by the translation units of libavcer_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/avcer_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// One packed tensor of a weight blob, resident on the device in f32 and (lazily) in bf16.
struct Tensor {
    float* f32 = nullptr;
    bf16_t* bf16 = nullptr;
    bf16_t* x3 = nullptr;  // split-fp16 copy (scaled hi/lo per 32-element K group + trailer, split_dev.h) for AVCER_MODE_F16X3
    bf16_t* x3f = nullptr; // the same in MFMA fragment order (conv_gemm dtype 7 / 8), where the shape allows it
    size_t numel = 0;
    int64_t dims[4] = {0, 0, 0, 0};
    int ndim = 0;
};

struct Model {
    bool loaded = false;
    std::map<std::string, Tensor> t;
    std::vector<void*> allocs;
};

struct avcer_ctx {
    int device = 0;
    char err[512] = {0};
    Model stat, dyn, aud, face;
    int aud_classes = 0;
    int static_batch = 1024;  // frames per internal pass of the static CNN (4 GiB buffer-descriptor limit at f32)
    int static_back = 0;      // frames per back pass of the static CNN (0: two front passes; avcer_set_static_back_batch)
    int static_lanes = 2;     // calls of lane_min .. lane_max frames as two half-batches on two streams (avcer_set_static_lanes)
    int lane_min = 32, lane_max = 2048;  // avcer_set_static_lane_range (tools/two_lane_sweep.py: two lanes win from 16 to 2048 frames)
    hipStream_t lane_stream = nullptr;  // the second lane's stream, created on first use, and its fork / join events
    hipEvent_t lane_ev[2] = {nullptr, nullptr};
    int block_slots = 512;    // 2 x hipDeviceProp_t::multiProcessorCount: what grid_rounds() divides a grid by
    // grow-only workspace arenas (activations), one per pipeline
    DevBuf ws[8];
    // device counter of the x3 mode's range contract: += 1 per thread that split a finite |x| >= 65520 into an fp16 pair
    // (split_dev.h sp_commit); read and reset by avcer_x3_overflow_count
    unsigned* ovf = nullptr;
    int64_t gemm_launches = 0;
    double gemm_flops = 0.0;
    // live HIP-event timing of conv_gemm launches (avcer_profile_*): pairs of events on the launch stream
    bool prof = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_fam;  // kernel family of every event pair (AVCER_FAM_*)
    struct ProfLaunch { double flops, bytes; long m, n, k; };
    std::vector<ProfLaunch> prof_log;  // per event pair: algorithmic FLOPs, compulsory bytes and the contraction's M, N, K
    size_t prof_used = 0;
    // per kernel family since the last avcer_profile_read*: launches, algorithmic FLOPs and compulsory HBM bytes
    int64_t fam_launches[8] = {0};
    double fam_flops[8] = {0};
    double fam_bytes[8] = {0};
    // one-shot debug tap (avcer_debug_tap): copy the named intermediate activation to a caller buffer
    std::string tap_name;
    void* tap_dst = nullptr;
    size_t tap_bytes = 0;
    int64_t tap_copied = -1;
};

int set_err(avcer_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return set_err((ctx), AVCER_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                                    \
    } while (0)

#define TRY(expr)                  \
    do {                           \
        int _r = (expr);           \
        if (_r != AVCER_OK) return _r; \
    } while (0)

int ws_reserve(avcer_ctx* ctx, int slot, size_t bytes, void** out);

// ---- gemm.hip
int launch_conv_gemm(avcer_ctx* ctx, const avcer_conv_desc& d, int dtype, const void* x, const void* w,
                     const float* scale, const float* bias, const void* residual, void* y, hipStream_t st,
                     const void* x2 = nullptr);

// kernel families of the MFMA launches (avcer_profile_read_families; include/avcer_hip.h AVCER_FAM_*)
enum { FAM_GEMM = 0, FAM_GEMM_WD = 1, FAM_CHAIN = 2, FAM_TAIL = 3, FAM_STEM = 4, FAM_SKINNY = 5, FAM_COUNT = 6 };
// `flops` / `bytes`: algorithmic work and compulsory HBM traffic of the launch (operands read once + outputs written once)
int prof_begin(avcer_ctx* ctx, hipStream_t st, hipEvent_t* ev0, hipEvent_t* ev1, int family, double flops, double bytes, long M = 0,
               long N = 0, long K = 0);

// ---- fused.hip (split-fp16 mode only)
// planes: fp16 hi plane [n][230][230][4] followed plane_bytes later by the lo plane; y: sp32 [n][55][55][64]
int launch_stem_pool(avcer_ctx* ctx, const void* planes, size_t plane_bytes, const void* w_x3, const float* scale,
                     const float* bias, void* y, int n, hipStream_t st);
int launch_stem_pool_u8(avcer_ctx* ctx, const uint8_t* frames, int in_h, int in_w, const void* w_x3, const float* scale,
                        const float* bias9, void* y, int n, hipStream_t st);
int launch_stem_pool_face(avcer_ctx* ctx, const uint8_t* frames, int h, int w, int rgb, const void* w_x3, const float* scale,
                          const float* bias, void* y, int n, hipStream_t st);
// conv2 + conv3 (+ residual) of one bottleneck and conv1 of the next block (t1n / w1n null when there is none);
// ds_cin = 0: x [M][4 planes] is the residual; ds_cin = 64: x [M][64] is the downsample operand and w3 is [4 planes][planes + 64];
// all activations sp32, weights split-fp16 (scaled, row-permuted) with the BN scale folded in (packing.py: *.wf, c3d.w)
// Rounds a grid of 256-thread blocks takes on the chip's block slots (two per CU: `slots` = 2 x the CU count the context
// read from the device at creation, 512 on a whole MI355X), as the form / tile choices model them.
// Calibrated on tools/ab_layers.py (profiles/r03_ab_layers*.txt, 128- against 112-row tiles of the same layer): a grid of
// at most one block per CU runs in 0.62 of a round (a block alone on its CU is that much faster); behind whole rounds, a
// partial round that still fits one block per CU (fraction f <= 0.5) costs 0.25 + 0.7 f -- the whole rounds end ragged and
// absorb part of it --, a larger one a whole round (some CU runs two blocks from start to end).
inline double grid_rounds(long tiles, long slots) {
    const long whole = tiles / slots, rest = tiles % slots;
    if (whole == 0) return 2 * tiles <= slots ? 0.62 : 1.0;
    const double f = (double)rest / (double)slots;
    return (double)whole + (rest == 0 ? 0.0 : (2 * rest <= slots ? 0.25 + 0.7 * f : 1.0));
}

// w2_frags: the conv2 weights once more in MFMA fragment order (k_weight_frags), or null: selects the spatial-tile form
// where it applies (planes 64, 55 x 55 images, a next conv1)
int launch_bneck(avcer_ctx* ctx, int planes, int nb, int h, int w, const void* t1, const void* x, int ds_cin, int out_step,
                 void* out, void* t1n, const void* w2, const float* b2, const void* w3, const float* b3, const void* w1n,
                 const float* b1n, hipStream_t st, const void* w2_frags = nullptr);

// conv3 + residual + ReLU of a planes-256 bottleneck and conv1 of the next block in one launch (t2 = conv2 output [M][256])
int launch_bneck_tail(avcer_ctx* ctx, int planes, long M, const void* t2, const void* x, void* out, void* t1n, const void* w3,
                      const float* b3, const void* w1n, const float* b1n, hipStream_t st);

int measure_ceilings(avcer_ctx* ctx, double* mfma_bf16_tflops, double* hbm_copy_tbs, hipStream_t st);

// ---- kernels.hip (element-wise / reduction kernels; T selects f32 (0) or bf16 (1) activations)
// kind: 0 = f32 [n,230,230,4], 1 = bf16, 3 = planar fp16 hi / lo (two [n,230,230,4] planes, the stem_pool input)
int k_preprocess(avcer_ctx*, const uint8_t* frames, int n, int in_h, int in_w, void* out, int kind, hipStream_t);
int k_maxpool3s2(avcer_ctx*, const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int bf16, hipStream_t);
int k_avgpool_hw(avcer_ctx*, const void* x, float* y, void* y_sp32, int n, int hw, int c, int kind, hipStream_t);
int k_small_linear(avcer_ctx*, const float* x, const float* w, const float* b, float* logits, float* probs, int m,
                   int k, int n, int relu_in, hipStream_t);
int k_lstm_cell(avcer_ctx*, const float* xproj, int64_t xproj_ld, const float* hproj, float* c, float* h_out, void* h_sp,
                int64_t h_ld, int n, int hid, int first, hipStream_t);
int k_wav_normalize(avcer_ctx*, const float* x, float* y, int n, int t, hipStream_t);
int k_conv0_ln_gelu(avcer_ctx*, const float* x, const float* w, const float* b, const float* g, const float* beta,
                    void* y, int n, int t_in, int t_out, int bf16, hipStream_t);
int k_layernorm(avcer_ctx*, const void* x, const void* res, const float* g, const float* b, void* yf, void* yb,
                int64_t rows, int c, float eps, int act, int in_kind, int yb_kind, hipStream_t);
int k_add_pe(avcer_ctx*, const float* x, const float* pe, float* yf, void* yb, int n, int s, int c, int yb_kind, hipStream_t);
int k_attention(avcer_ctx*, const void* qkv, void* out, int n, int s, int heads, int d, float scale, int in_kind,
                int out_kind, hipStream_t);
int k_maxpool1d_relu(avcer_ctx*, const float* x, float* y, void* y_sp32, int n, int t_in, int t_out, int c, int k, hipStream_t);
int k_mean_time_relu(avcer_ctx*, const float* x, float* y, int n, int t, int c, hipStream_t);
int k_f32_to_bf16(avcer_ctx*, const float* x, bf16_t* y, size_t n, hipStream_t);
int k_frame_mean(avcer_ctx*, const float* win_logits, const int32_t* lo, const int32_t* hi, int n_win, int c,
                 int n_frames, float* out, int32_t* count, hipStream_t);
int k_fuse(avcer_ctx*, const float* stat, const float* dyn, const float* aud, int n, int n_aud, int aud_c,
           const double* w /*[21] device-side by value*/, int has_w1, int cwt, int cmask, double* comp_prob,
           int32_t* comp_argmax, hipStream_t);
int k_pack_nchw(avcer_ctx*, const float* x, int n, void* out, int kind, hipStream_t);
int k_gather_windows(avcer_ctx*, const float* feats, const int32_t* idx, int nwin, float* out, hipStream_t);
int k_audio_chunks(avcer_ctx*, const float* wav, const int32_t* starts, const int32_t* ends, int n, int window, int mode,
                   float* out, hipStream_t);
int k_split_weights(avcer_ctx*, const float* w, bf16_t* out, size_t n, hipStream_t);
int k_split_weight_rows(avcer_ctx*, const float* w, bf16_t* out, int n, int k, hipStream_t);
int k_weight_frags(avcer_ctx*, const bf16_t* rows, bf16_t* out, int n, int k, hipStream_t);
int k_face_decode(avcer_ctx*, const float* loc, const float* conf, const float* landms, const float* priors, int T, int P, int im_h,
                  int im_w, float var0, float var1, float* dets, hipStream_t);
int k_face_nms(avcer_ctx*, const float* dets, int T, int P, float conf_thresh, float nms_thresh, int nms_top_k, int top_k,
               float threshold, int32_t* order, int32_t* count, float* out, int32_t* out_n, hipStream_t);
int k_face_pre(avcer_ctx*, const uint8_t* frames, int n, int h, int w, int ph, int pw, int rgb, void* out, int bf16, hipStream_t);
int k_maxpool3s2p1(avcer_ctx*, const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int kind, hipStream_t);
int k_upsample_add(avcer_ctx*, void* y, const void* coarse, int n, int h, int w, int ch, int cw, int c, int kind, hipStream_t);
int k_face_head(avcer_ctx*, const float* hd, int ld, int n, int hw, int row0, int P, float* loc, float* conf, float* landms,
                hipStream_t);
int k_crop_tiles(avcer_ctx*, const uint8_t* frames, int T, int H, int W, const int32_t* rects, int n, int swap_rb,
                 uint8_t* tiles, hipStream_t);
