/*
 * libavcer_hip.so -- C ABI of the MI355X (gfx950) implementation of AVCER's inference hot path.
 *
 * AVCER (github.com/ElenaRyumina/AVCER) has no FFI/plugin layer: its seam is the Python call surface
 * of four callables and two numpy functions.  Each entry point below replaces one
 * of them; the reference location it stands in for is cited as  ref: <file>:<lines>  relative to the
 * reference's src/ directory.  The Python mirror with the reference's own call signatures lives in
 * avcer_amd/ (models.py, video_pipeline.py, audio_pipeline.py, fusion.py) and binds these symbols
 * with ctypes; INTEGRATION.md shows the stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative AVCER_E* code and never throws;
 *     avcer_last_error(ctx) returns a human-readable message for the last failure on that context;
 *   - all tensor pointers are DEVICE pointers owned by the caller unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on it;
 *   - weights and workspace are owned by the context (hipMalloc at load / first use; a call that has to
 *     grow the workspace synchronises the device once);
 *   - one context per (device, host thread): a context is not re-entrant.
 */
#ifndef AVCER_HIP_H
#define AVCER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avcer_ctx avcer_ctx;
typedef void* avcer_stream_t;

enum {
    AVCER_OK = 0,
    AVCER_EINVAL = -1,   /* bad argument / shape */
    AVCER_ENOMEM = -2,   /* device allocation failed */
    AVCER_EHIP = -3,     /* HIP runtime error (message holds hipGetErrorString) */
    AVCER_ESTATE = -4,   /* weights for this model not loaded */
    AVCER_EFORMAT = -5   /* packed weight blob malformed */
};

/* arithmetic mode of the MFMA contractions */
enum {
    AVCER_MODE_FP32 = 0, /* f32 operands, v_mfma_f32_32x32x2_f32: exact f32 FMA chains */
    AVCER_MODE_BF16 = 1, /* bf16 operands / f32 accumulate, v_mfma_f32_16x16x32_bf16: throughput mode, NOT parity-grade */
    AVCER_MODE_F16X3 = 2 /* the fast parity mode: f32-grade activations and results; every operand is carried as an fp16
                            pair hi + lo (11 + 11 significand bits) and every product as ah*wh + ah*wl + al*wh on
                            v_mfma_f32_16x16x32_f16 with f32 accumulation: 3 MFMAs per product, ~7e-8 relative error per
                            contraction (rounds 1-3 split into bf16 pairs: 4.5e-6; that mode was AVCER_MODE_BF16X3) */
};
/* Range contract of AVCER_MODE_F16X3 (csrc/split_dev.h).  Activations are stored unscaled as fp16 pairs: every
 * intermediate activation must satisfy |x| < 65504.  A larger value becomes +-inf in its hi half and NaN in hi + lo, and the
 * NaN reaches the outputs: an overflow is never a wrong finite number (tests/test_gpu_edges.py).  Because NaN is also what
 * the reference legitimately returns for an empty audio window (get_prob_audio_8_cl.py:78-92), the library COUNTS the event
 * itself: avcer_x3_overflow_count below; the host mirrors read it once per call and repeat the call in AVCER_MODE_FP32
 * (no range limit) when it is not zero.  Below 2^-3 the lo half is an fp16 subnormal: a small element carries an ABSOLUTE
 * error of at most 2^-25 -- the usable window of the f32-grade RELATIVE bound is [2^-3, 65504).  Weights are multiplied
 * by one power of two per matrix before the split (largest |w| -> [2^14, 2^15)); the inverse, also a power of two, rides
 * behind the split data (AVCER_SPLIT_TRAILER) and every consumer folds it into its epilogue, so no result depends on it.
 * Error bounds of the two parity-grade modes against the reference's CPU path are in DESIGN.md section 6 and asserted by
 * tests/test_gpu_parity_breadth.py: ONE gate of 1e-4 on probabilities for both modes at 1 x, 4 x and 8 x the synthetic
 * generator's logit scale. */

/* Bumped whenever a struct layout, an argument list or a buffer size of this header changes incompatibly; the Python binding
 * refuses a library whose avcer_abi_version() differs (avcer_amd/_lib.py).
 *   2: avcer_conv_desc grew r_sub / r_h / r_w / tile_m, avcer_bneck_chain gained out_step, avcer_set_option left (round 3);
 *      split weight buffers carry a trailer and AVCER_MODE_BF16X3 became AVCER_MODE_F16X3 (round 4).
 *   3: avcer_x3_overflow_count, avcer_profile_read_families; avcer_bneck_chain gained w2_frags (round 5).
 *   4: avcer_source_hash, avcer_set_static_back_batch, avcer_set_static_lanes, avcer_set_static_lane_range, avcer_face_decode_batch, avcer_track_faces,
 *      avcer_lsap, avcer_profile_read_launches (round 6). */
#define AVCER_ABI_VERSION 4
int avcer_abi_version(void);
/* Hash (16 hex digits) of the sources and headers this binary was compiled from, embedded at build time by
 * avcer_amd/build.py (source_hash()).  The Python binding refuses a library whose hash differs from the tree's, and bench.py
 * prints it beside the tree's hash: a stale binary with the right ABI number cannot be measured under a fresh label. */
const char* avcer_source_hash(void);

int avcer_ctx_create(int device, avcer_ctx** out);
void avcer_ctx_destroy(avcer_ctx* ctx);
const char* avcer_last_error(const avcer_ctx* ctx);

/* Run-time signal of AVCER_MODE_F16X3's range contract.  *count = how many GPU threads of this context's launches have
 * turned a FINITE activation of magnitude >= 65520 into an infinite fp16 hi half since the last reset (every split site:
 * GEMM / chain / stem epilogues, LayerNorm and GELU outputs, attention Q / K / V, the on-the-fly split of f32 operands).
 * 0: any NaN in an output came in through the input -- the reference's own result for an empty audio window.  > 0: outputs
 * produced since the last reset may hold NaN where the reference (fp32, no range limit) holds numbers
 *   ref: get_prob_video.py:107-112, get_prob_audio_8_cl.py:87-92 (fp32 forward passes)
 * -- repeat the call with AVCER_MODE_FP32.  Waits for `stream`; reset != 0 zeroes the counter behind the read.
 * count == NULL with reset != 0: an asynchronous reset queued on `stream`, nothing read, nothing waited for -- what a guarded call
 * puts in front of its launches so that counts left by earlier work on the context are not charged to it.  The counter sees the
 * launches of every stream; the read orders only behind `stream`, so join side streams into it first. */
int avcer_x3_overflow_count(avcer_ctx* ctx, int reset, int64_t* count, avcer_stream_t stream);

/* Packed weights (host pointers; the library copies them to the device and owns the copy).
 * Blob layout: see avcer_amd/packing.py (header "AVCERW01", tensor table, 64-byte aligned f32 payloads).
 *   ref: get_prob_video.py:22-25 (ResNet50 state_dict), :51-54 (LSTM state_dict),
 *        get_prob_audio_8_cl.py:52-66 (ExprModelV3 state_dict) */
int avcer_load_static(avcer_ctx* ctx, const void* blob_host, size_t nbytes);
int avcer_load_dynamic(avcer_ctx* ctx, const void* blob_host, size_t nbytes);
int avcer_load_audio(avcer_ctx* ctx, const void* blob_host, size_t nbytes);

/* Static visual model on face tiles.
 *   ref: data/utils.py:19-39 (pth_processing: NEAREST resize to 224, u8 HWC -> f32, RGB->BGR, mean subtract)
 *        architectures/video.py:115-133 (ResNet.extract_features / forward)
 *        get_prob_video.py:47-49,103-112 (fc1 forward hook "features", softmax(dim=1))
 * frames_hwc  u8 [n, in_h, in_w, 3] RGB.   logits/probs f32 [n,7] (video class order), feats f32 [n,512]
 * (fc1 output BEFORE ReLU).  Any of the three outputs may be NULL. */
int avcer_static_forward(avcer_ctx* ctx, const uint8_t* frames_hwc, int n, int in_h, int in_w, int mode,
                         float* logits, float* probs, float* feats, avcer_stream_t stream);

/* Frames per FRONT pass of the static CNN (stem, stage 1, first block of stage 2: the 55 x 55 tensors), 1..1024 (default 1024).
 * Results do not depend on it (every kernel's arithmetic per frame is independent of the batch around it). */
int avcer_set_static_batch(avcer_ctx* ctx, int frames);
/* Frames per BACK pass (rest of stage 2, stages 3-4, tail), 1..2048, or 0 = two front passes (the default: 2048 at the default
 * front pass).  The late layers' grids are small, so the back wants as many frames per launch as the 4 GiB descriptors allow
 * whatever the front pass is.  A scheduling knob like the one above: results do not depend on it. */
int avcer_set_static_back_batch(avcer_ctx* ctx, int frames);
/* Lanes of a static-CNN call: 2 (default) = a call of 32-2048 frames (avcer_set_static_lane_range) runs as two half-batches on two
 * HIP streams (the context's own second stream, forked from and joined into `stream` by events; a second workspace): below a few
 * thousand frames the grids of stages 3-4 are a fraction of a round of block slots, and the two halves fill each other's tails:
 * -4 % at 32 frames, -8 ... -12 % at 48-192, -5 % at 256-512, -2 ... -3 % at 640-1536, -1.7 % at 2048 (tools/two_lane_sweep.py,
 * profiles/r06_two_lane_sweep.txt; at 8 frames +4 %: latency chains).  1 = always on `stream` alone.  Results are bit-identical
 * either way (a frame's result does not depend on the batch around it); calls made while avcer_profile_enable is on or a debug tap
 * is armed are serial regardless, so that per-launch events stay meaningful. */
int avcer_set_static_lanes(avcer_ctx* ctx, int lanes);
/* The call sizes that take the two-lane schedule: min_frames <= n <= max_frames, 2 <= min <= max <= 4096 (default 32 .. 2048).
 * A scheduling knob: results do not depend on it. */
int avcer_set_static_lane_range(avcer_ctx* ctx, int min_frames, int max_frames);

/* The same model on an already preprocessed tensor, i.e. the exact argument of the reference's
 * `pth_model_static(x)`:  x f32 [n,3,224,224] (BGR, mean-subtracted).   ref: get_prob_video.py:103-109 */
int avcer_static_forward_nchw(avcer_ctx* ctx, const float* x, int n, int mode, float* logits, float* probs,
                              float* feats, avcer_stream_t stream);

/* LSTM window assembly: out[w, s, :] = relu(feats[idx[w, s], :]).
 *   ref: get_prob_video.py:115-123 (F.relu(features), 10-deep sliding window, first feature replicated x10)
 * feats f32 [*,512], idx i32 [nwin,10] (device), out f32 [nwin,10,512]. */
int avcer_gather_windows(avcer_ctx* ctx, const float* feats, const int32_t* idx, int nwin, float* out,
                         avcer_stream_t stream);

/* Dynamic visual model on windows of 10 ReLU'd fc1 features.
 *   ref: architectures/video.py:169-185 (LSTMPyTorch.forward), get_prob_video.py:122-129
 * windows f32 [n,10,512] -> logits f32 [n,7] (raw logits, no softmax). Always f32 arithmetic. */
int avcer_dynamic_forward(avcer_ctx* ctx, const float* windows, int n, float* logits, avcer_stream_t stream);
/* Same with an arithmetic mode: AVCER_MODE_F16X3 runs the projections on the split-fp16 MFMA (f32 state, f32-grade
 * results); AVCER_MODE_BF16 keeps them in f32 (the recurrence is latency-bound, not worth a third weight copy). */
int avcer_dynamic_forward_mode(avcer_ctx* ctx, const float* windows, int n, int mode, float* logits,
                               avcer_stream_t stream);

/* Audio model on padded waveform windows.
 *   ref: get_prob_audio_8_cl.py:87-92 (HF feature-extractor normalisation + audio_model(x))
 *        architectures/audio_8_cl.py:179-190 (ExprModelV3.forward), architectures/attention_layers.py:221-267
 *        transformers==4.36.2 Wav2Vec2Model (third party, config of audeering/wav2vec2-large-robust-12-ft-emotion-msp-dim)
 * wav f32 [n,t] (already padded to the window), normalize != 0 applies (x-mean)/sqrt(var+1e-7) per row first.
 * logits f32 [n, n_classes] raw logits (n_classes = 8 for ExprModelV3, 7 for ExprModelV2 weights). */
int avcer_audio_forward(avcer_ctx* ctx, const float* wav, int n, int t, int normalize, int mode,
                        float* logits, avcer_stream_t stream);
int avcer_audio_num_classes(const avcer_ctx* ctx);

/* Window slicing + padding of one waveform.
 *   ref: get_prob_audio_8_cl.py:78-86, data/utils.py:63-71 (pad_wav, "repeat"), :74-89 (pad_wav_zeros, "mean"/"constant")
 * wav f32 [len]; starts/ends i32 [n] (device) sample ranges; out f32 [n, window];
 * mode 0 = pad with the chunk mean (NaN for an empty chunk, as torch.mean does), 1 = zeros, 2 = repeat
 * (an empty chunk makes the reference raise ZeroDivisionError, data/utils.py:66; the host mirror raises the same,
 * and the kernel itself writes NaN for such a row instead of indexing with i % 0). */
int avcer_audio_chunks(avcer_ctx* ctx, const float* wav, const int32_t* starts, const int32_t* ends, int n,
                       int window, int mode, float* out, avcer_stream_t stream);

/* Per-frame mean of window logits.
 *   ref: get_prob_audio_8_cl.py:94-101 (logits replicated for frames [lo,hi) of each window),
 *        run.py:90 (audio_df.groupby("frames").mean())
 * win_logits f32 [n_win, c], frame_lo/hi i32 [n_win]; out f32 [n_frames, c]; count i32 [n_frames]
 * (count 0 -> row left as zeros: no window covers that frame). */
int avcer_audio_frame_mean(avcer_ctx* ctx, const float* win_logits, const int32_t* frame_lo, const int32_t* frame_hi,
                           int n_win, int c, int n_frames, float* out, int32_t* count, avcer_stream_t stream);

/* RetinaFace-R50 detector network (row f4).
 *   ref: retina_face/retina_face.py:46-115 (RetinaFace, phase "test"), retina_face_net.py:42-101 (SSH, FPN), torchvision
 *        ResNet-50 body (layer2/3/4 returned), retina_face_predictor.py:59-65 (pixels minus 104, 117, 123)
 * avcer_load_face: packed weights (avcer_amd.packing.pack_face of RetinaFace(cfg_re50).state_dict()).
 * avcer_face_forward: frames u8 [n,h,w,3] in cv2's BGR order (rgb != 0: RGB, flipped first, as the predictor does),
 *   any h, w >= 32 -> loc f32 [n,P,4], conf f32 [n,P,2] (softmaxed), landms f32 [n,P,10] with
 *   P = avcer_face_num_priors(h, w) rows in PriorBox order; feed them to avcer_face_decode. */
int avcer_load_face(avcer_ctx* ctx, const void* packed, size_t nbytes);
int avcer_face_num_priors(int h, int w);
int avcer_face_forward(avcer_ctx* ctx, const uint8_t* frames, int n, int h, int w, int rgb, int mode, float* loc,
                       float* conf, float* landms, avcer_stream_t stream);

/* The post-decode half of RetinaFacePredictor.__call__ for a batch of frames, on the device:
 *   ref: retina_face/retina_face_predictor.py:86-108 (confidence floor, NMS, top-k, final threshold),
 *        retina_face/py_cpu_nms.py:11-39 (greedy NMS, "+1 pixel" areas, visit order = descending score).
 * dets f32 [n_frames, n_priors, 15] as avcer_face_decode writes them; out f32 [n_frames, top_k, 15] receives, per frame,
 * the kept rows in the reference's order; out_n i32 [n_frames] their number.  nms_top_k <= 6144, top_k <= 1024.
 * Equal scores are visited HIGHER prior index first: the reference's `scores.argsort()[::-1]` is an ascending sort read
 * backwards, so this is its order wherever that sort keeps ties in index order (`kind="stable"`; numpy's default sort on
 * short arrays).  On long arrays numpy's default sort does not define the order of ties, so for exactly tied scores a
 * given numpy build may visit (and keep) a different member of the tie. */
int avcer_face_nms(avcer_ctx* ctx, const float* dets, int n_frames, int n_priors, float conf_thresh, float nms_thresh,
                   int nms_top_k, int top_k, float threshold, float* out, int32_t* out_n, avcer_stream_t stream);

/* Face stage ("next" row f4): the arithmetic either side of the RetinaFace network.
 *   avcer_face_decode  ref: data/face_detection/ibug/face_detection/retina_face/retina_face_predictor.py:70-82,
 *                            box_utils.py:210-249 (decode, decode_landm), scaled to pixels
 *     loc f32 [P,4], conf f32 [P,2] (softmaxed), landms f32 [P,10], priors f32 [P,4] (cx, cy, w, h; prior_box.py:16-33)
 *     -> dets f32 [P,15] = x0, y0, x1, y1, score, 5 x (lx, ly): the row layout the predictor returns, BEFORE its
 *     confidence filter / NMS / top-k (host side, avcer_amd/face_tiles.py, as in the reference).
 *   avcer_crop_tiles   ref: data/get_face_images.py:52-56 (crop fr[y0:y1, x0:x1]) + data/utils.py:34 (PIL NEAREST
 *                            resize to 224x224), without the JPEG file in between
 *     frames u8 [T,H,W,3], rects i32 [n,5] = frame, x0, y0, x1, y1 (end-exclusive, inside the frame),
 *     swap_rb != 0 when the frames are BGR (cv2 order) -> tiles u8 [n,224,224,3] RGB, the input of
 *     avcer_static_forward.  A rect that is empty or leaves the frame yields an all-zero tile. */
int avcer_face_decode(avcer_ctx* ctx, const float* loc, const float* conf, const float* landms, const float* priors,
                      int n_priors, int im_h, int im_w, float var0, float var1, float* dets, avcer_stream_t stream);
/* avcer_face_decode for a batch of frames in one launch: loc [T,P,4], conf [T,P,2], landms [T,P,10] (what avcer_face_forward
 * writes for T frames) -> dets [T,P,15]; the priors [P,4] are shared.  T <= 65535. */
int avcer_face_decode_batch(avcer_ctx* ctx, const float* loc, const float* conf, const float* landms, const float* priors,
                            int n_frames, int n_priors, int im_h, int im_w, float var0, float var1, float* dets,
                            avcer_stream_t stream);
int avcer_crop_tiles(avcer_ctx* ctx, const uint8_t* frames, int n_frames, int h, int w, const int32_t* rects, int n,
                     int swap_rb, uint8_t* tiles, avcer_stream_t stream);

/* The face tracker between detector and tiles, for a whole video in ONE call -- HOST code and HOST pointers (the tracker is
 * sequential in time and sees a handful of boxes per frame; the reference runs it on the host too):
 *   ref: data/face_detection/ibug/face_detection/utils/simple_face_tracker.py:10-90 (IoU distance, Hungarian assignment by
 *        scipy.optimize.linear_sum_assignment, tracklets dropped on an empty frame), data/get_face_images.py:38-63
 *        (VideoPredictor.process: tracker per frame, crop rectangle fr[y0:y1, x0:x1] of every detection).
 * dets host f32 [sum(counts), ld] (x0, y0, x1, y1 first; ld >= 4), counts host i32 [n_frames] detections per frame ->
 * records host i64 [sum(counts), 6] = frame, track directory (track id - 1), x0, y0, x1, y1 (the clamped half-open crop) in the
 * reference's write order, *n_records rows.  AVCER_EINVAL where the reference raises: a zero-area detection (no track id) or an
 * empty crop; avcer_last_error names the frame (ctx may be NULL: no device is touched, errors are then the code alone).  avcer_lsap is the assignment step on its own (scipy's algorithm and
 * tie-breaking; cost host f64 [nr, nc] -> min(nr, nc) pairs sorted by row), exported for the host-logic tests. */
int avcer_track_faces(avcer_ctx* ctx, const float* dets_host, int ld, const int32_t* counts_host, int n_frames, int frame_w,
                      int frame_h, double iou_threshold, double minimum_face_size, int64_t* records_host, int64_t* n_records);
int avcer_lsap(int nr, int nc, const double* cost_host, int32_t* rows_host, int32_t* cols_host);

/* Probability fusion and compound-expression rule.
 *   ref: run.py:25-165 (get_c_expr_db_pred), data/utils.py:125-127 (softmax), :222-241 (get_compound_expression)
 * stat f32 [n,7] softmaxed static probs and dyn_logits f32 [n,7], both in VIDEO column order;
 * aud_mean f32 [n_aud, c>=7] per-frame mean audio logits in audio order, rows >= n_aud repeat row n_aud-1 (run.py:99-103);
 * w1 host f64 [3,7] or NULL (plain mean, run.py:113-114), w2 host f64 [3];
 * comp_prob f64 [4,n,7] and comp_argmax i32 [4,n] in the order AV, VS, VD, A. */
int avcer_fuse(avcer_ctx* ctx, const float* stat, const float* dyn_logits, const float* aud_mean, int n, int n_aud,
               int aud_c, const double* w1_host, const double* w2_host, int ce_weights_type, int ce_mask,
               double* comp_prob, int32_t* comp_argmax, avcer_stream_t stream);

/* The contraction kernel itself (implicit-GEMM convolution with fused epilogue), exported for kernel-level
 * parity tests and micro-benchmarks:  Y[m, n] = act(scale[n] * sum_k A[m,k] * W[n,k] + bias[n] (+ R[m,n]))
 * where A is gathered from an NHWC tensor.  See avcer_conv_desc. dtype: 0 = f32 in/out, 1 = bf16 in/out,
 * 2 = bf16 in / f32 out; split-fp16 ("x3") arithmetic with w pre-split by avcer_split_weight_rows: 3 = f32 in / f32 out,
 * 4 = f32 in / sp32 out, 5 = sp32 in / sp32 out (+ sp32 residual), 6 = sp32 in / f32 out (+ f32 residual);
 * 7 / 8 = 5 / 6 with w in fragment order (avcer_weight_frags): the weights-direct form of the kernel, bit-identical results,
 * for n % 256 == 0, an even number of 32-element K-steps and groups <= 1 (anything else is AVCER_EINVAL: use 5 / 6);
 * 9 / 10 = 5 / 6 with w in fragment order once more, the "skinny" form for launches of few positions (one frame or one window
 * per call, the deep layers of a small batch): one WAVE per (16 | 32 | 64 positions) x 16 channels, registers only, no LDS, no
 * barrier; bit-identical results as well.  Requires cin % 32 == 0 and n % 32 == 0 (per group), no second source (x2_cin == 0);
 * grouped convolutions are supported (one launch, as for 5 / 6); M is not limited by the kernel.  tile_m = 16, 32 or 64 picks
 * the positions per wave tile (0: the smallest tile that leaves half of the chip's SIMDs free); tile_n does not apply.  Anything
 * else is AVCER_EINVAL: use 5 / 6.  The networks choose it while (M / 32) * (n / 16) wave tiles fit the chip's SIMDs in one
 * round and M <= 4096 (api.hip prefer_skinny).
 * "sp32" storage = per aligned group of 32 channels, 32 fp16 hi values then 32 fp16 lo values (x = hi + lo), i.e. the
 * layout avcer_split_weights produces; 4 bytes per element. */
typedef struct avcer_conv_desc {
    int32_t batch, in_h, in_w;       /* input extents used for bounds (zero padding outside) */
    int32_t out_h, out_w;            /* M = batch*out_h*out_w */
    int32_t cin;                     /* contiguous channels per tap (multiple of 8; of 4 for f32) */
    int32_t kh, kw;                  /* taps; K = kh*kw*cin */
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int64_t x_stride_b, x_stride_h, x_stride_w; /* element strides of the input */
    int32_t x_coff;                  /* channel offset into the input pixel */
    int32_t n;                       /* output channels (multiple of 64) */
    int64_t y_ld; int32_t y_coff;    /* output row stride (elements) and channel offset */
    int64_t r_ld; int32_t r_coff;    /* residual row stride / offset (if residual != NULL) */
    int32_t act;                     /* 0 none, 1 relu, 2 gelu(erf), 3 gelu with the Abramowitz-Stegun 7.1.26 erf: within 4.7e-7 of
                                        the exact function (the f32 rounding of the exact form itself), a third of the
                                        instructions; what the library uses around bf16 / split-fp16 contractions */
    int32_t res_after_act;           /* 0: act(v + r), 1: act(v) + r */
    int32_t groups;                  /* 0/1 = plain; G > 1 = grouped convolution in ONE launch: group g reads input
                                        channels x_coff + g*cin, uses weight rows [g*n, (g+1)*n) of w (and scale/bias
                                        entries g*n..), and writes / adds channels y_coff + g*n, r_coff + g*n */
    /* Optional second A source for avcer_conv_gemm_dual (two fused 1x1 convolutions, e.g. ResNet conv3 + downsample):
     * K elements [kh*kw*cin, kh*kw*cin + x2_cin) of every row come from x2 at position (oy*x2_stride, ox*x2_stride). */
    int32_t x2_cin, x2_coff, x2_stride;
    int64_t x2_stride_b, x2_stride_h, x2_stride_w;
    int32_t tile_n;                  /* output-channel width of the block tile: 0 = chosen by the library, 64 or 128 (n % 128 == 0);
                                        dtypes 7 / 8 always use 256 (0 or 256).  A tuning knob: results do not depend on it. */
    /* Sub-sampled residual (r_sub > 1): the residual tensor lives on a [batch, r_h, r_w] grid of rows of r_ld elements and
     * output position (b, oy, ox) adds its row (b, oy * r_sub, ox * r_sub) -- the last bottleneck of a ResNet stage computed
     * only at the positions the next stage's stride-2 1x1 convolutions read.  0 / 1: one residual row per output row. */
    int32_t r_sub, r_h, r_w;
    int32_t tile_m;                  /* dtypes 7 / 8: positions per block tile, 0 = chosen by the library (whichever of 112 / 128
                                        leaves the cheaper last round on the 512 block slots), 112 or 128.  Dtypes 9 / 10:
                                        positions per WAVE tile, 0 = chosen by the library, 16, 32 or 64.  A tuning knob like
                                        tile_n: an element's accumulation order, hence the result, does not depend on it. */
} avcer_conv_desc;

int avcer_conv_gemm(avcer_ctx* ctx, const avcer_conv_desc* d, int dtype, const void* x, const void* w,
                    const float* scale, const float* bias, const void* residual, void* y, avcer_stream_t stream);

/* Same contraction with a second activation tensor supplying the tail of K (see avcer_conv_desc.x2_*); w is
 * [n][kh*kw*cin + x2_cin].  Both sources must be 1x1 / unpadded. */
int avcer_conv_gemm_dual(avcer_ctx* ctx, const avcer_conv_desc* d, int dtype, const void* x, const void* x2, const void* w,
                         const float* scale, const float* bias, const void* residual, void* y, avcer_stream_t stream);

/* Fused kernels of the static CNN in the split-fp16 arithmetic (csrc/fused.hip), exported for kernel-level parity tests.
 *
 * avcer_bneck_chain: the tail of one ResNet bottleneck and the head of the next in ONE launch,
 *     T2 = relu(conv3x3(T1) + b2);  OUT = relu(conv1x1(T2) + b3 + X);  T1N = relu(conv1x1(OUT) + b1n)
 *   ref: architectures/video.py:43-60 (Bottleneck.forward; stride 1, no downsample), BatchNorm folded.
 *   t1 sp32 [nb,h,w,planes], x / out sp32 [nb,h,w,4*planes], t1n sp32 [nb,h,w,planes] or NULL (then w1n, b1n NULL);
 *   w2 [planes][9*planes], w3 [4*planes][planes], w1n [planes][4*planes]: BN scale folded into the rows, then split
 *   by avcer_split_weight_rows; b2 / b3 / b1n f32 BN shifts in natural channel order.  planes = 64 or 128.
 *   ds_cin = 64 (planes 64 only, t1n required): the FIRST block of a stage without spatial stride (video.py:43-60 with
 *   i_downsample): x is the 64-channel block input sp32 [nb,h,w,64], w3 is [4*planes][planes + 64] = conv3 and the
 *   downsample convolution concatenated along K (both BN scales folded, shifts summed into b3), and nothing is added.
 *   out_step = 2 (t1n NULL, ds_cin 0): the LAST block of a stage, whose output only the next stage's stride-2 1x1
 *   convolutions read (video.py:12-19,140-149): T2 and OUT are evaluated at positions (2 oy, 2 ox) alone and out is the compact
 *   sp32 [nb, (h+1)/2, (w+1)/2, 4*planes]; t1 and x keep their [nb,h,w] grids.  out_step = 1: every position.
 *   w2_frags (may be NULL): the conv2 weights once more in MFMA fragment order (avcer_weight_frags of the same matrix).  With
 *   it, planes 64 on 55 x 55 images with a next conv1 runs the spatial-tile form (11 x 11 tiles, resident halo patch, the
 *   conv2 output channels split across the block's waves so that every weight fragment goes straight into the registers of
 *   ONE wave); results are bit-identical to the form without it.
 *
 * avcer_stem_pool: conv 7x7/2 (TF-"same" padding) + BN + ReLU + max-pool 3x3/2 in one launch,
 *   ref: architectures/video.py:63-90,98-103,116-117.  planes_hi_lo: two fp16 planes [n,230,230,4] (hi, then lo plane_bytes
 *   later) of the zero-bordered preprocessed image as avcer_static_forward builds it; w split [64][7*32] (tap rows of
 *   8 pixels x 4 channels); scale / bias f32 [64]; y sp32 [n,55,55,64]. */
int avcer_bneck_chain(avcer_ctx* ctx, int planes, int nb, int h, int w, const void* t1, const void* x, int ds_cin, int out_step,
                      void* out, void* t1n, const void* w2, const float* b2, const void* w3, const float* b3, const void* w1n,
                      const float* b1n, const void* w2_frags, avcer_stream_t stream);
int avcer_stem_pool(avcer_ctx* ctx, const void* planes_hi_lo, size_t plane_bytes, const void* w, const float* scale,
                    const float* bias, void* y, int n, avcer_stream_t stream);
/* The same launch fed with the u8 frames themselves ([n,in_h,in_w,3] RGB; data/utils.py:19-39 -- NEAREST resize to 224, BGR flip,
 * mean subtraction -- happens inside): raw pixel values are exact in fp16, so the stem runs two MFMAs per product, and the
 * channel means move into shifts9 f32 [9][64] = BN shift - BN scale * (sum over the taps inside the image of w * mean), one
 * row per border class 3 * row_class + col_class (0: first stem row / column, 1: interior, 2: row / column 110; see
 * avcer_amd/packing.py stem_border_shifts).  What avcer_static_forward runs in AVCER_MODE_F16X3. */
int avcer_stem_pool_u8(avcer_ctx* ctx, const uint8_t* frames, int n, int in_h, int in_w, const void* w, const float* scale,
                       const float* shifts9, void* y, avcer_stream_t stream);

/* The sp32 split of an ACTIVATION tensor (what a producer's epilogue writes with dtype 4 / 5): for every group of 32
 * elements, 32 fp16 "hi" values then 32 fp16 "lo" values with x = hi + lo (+ O(2^-22 |x|); |x| < 65504).  x f32 [numel] (a
 * multiple of 32) -> out, same size in bytes, no scaling.  Both device pointers.  NOT a weight layout: the weights of dtypes
 * 3-8 are scaled, carry a trailer and have their rows permuted (avcer_split_weight_rows below). */
int avcer_split_weights(avcer_ctx* ctx, const float* w, void* out, size_t numel, avcer_stream_t stream);

/* Bytes behind the n * k * 4 bytes of a split WEIGHT matrix that hold its scale: float [0] = the power of two a consumer
 * multiplies its accumulators by (the inverse of the scale the weights were split at), word [1] = max |w| as float bits.
 * Every buffer written by avcer_split_weight_rows / avcer_weight_frags is n * k * 4 + AVCER_SPLIT_TRAILER bytes. */
#define AVCER_SPLIT_TRAILER 256

/* The split of a WEIGHT matrix w f32 [n][k] (n, k multiples of 32) as the x3 contractions (dtype 3-6 of avcer_conv_gemm,
 * avcer_bneck_chain, avcer_stem_pool) expect it: every value times the matrix's power-of-two scale (largest magnitude ->
 * [2^14, 2^15): keeps the lo halves of all significant weights normal fp16 numbers), the layout above along K, and the rows
 * of every group of 32 output channels re-ordered so that stored row 16t + 4g + r holds channel 8g + 4t + r (t = 0,1;
 * g = 0..3; r = 0..3).  With weights as the MFMA A operand this leaves each lane with 8 consecutive output channels, i.e.
 * 16-byte pieces of the output row (direct whole-line stores).  scale / bias / residual / output stay in natural channel
 * order and in real units: the kernels undo the weight scale themselves from the trailer.
 * out: n * k * 4 + AVCER_SPLIT_TRAILER bytes. */
int avcer_split_weight_rows(avcer_ctx* ctx, const float* w, void* out, int n, int k, avcer_stream_t stream);

/* The output of avcer_split_weight_rows once more in MFMA fragment order, the weight layout of avcer_conv_gemm dtypes 7 / 8
 * (conv_gemm_wd_kernel: weight fragments go straight from global memory to the registers, only the activation tile passes
 * through LDS): [n/16][k/32][hi, lo][64 lanes][16 bytes], lane l = stored row 16 t + (l & 15), K elements 8 (l >> 4) .. + 8,
 * then the trailer.  Dtypes 9 / 10 read the same copy.  rows, out: device pointers, n * k * 4 + AVCER_SPLIT_TRAILER bytes each; n a multiple of 16, k of 32. */
int avcer_weight_frags(avcer_ctx* ctx, const void* rows, void* out, int n, int k, avcer_stream_t stream);

/* The attention kernel on its own (kernel-level parity tests): softmax(Q K^T * scale) V per (row block, head).
 *   ref: architectures/attention_layers.py:80-144 (ScaledDotProductAttention inside MultiHeadAttention), transformers
 *        Wav2Vec2Attention (eager).
 * qkv [n, s, 3 * heads * head_dim]: per token the queries of all heads, then the keys, then the values (the packed output of
 * the fused q / k / v projection); out [n, s, heads * head_dim].  head_dim 64 or 32, s <= 256.  Storage kinds: 0 = f32,
 * 1 = bf16, 2 = sp32.  (in 0, out 0): exact f32 arithmetic on the VALU; (in 1, out 1): bf16 operands on the MFMA;
 * (in 0, out 2): what AVCER_MODE_F16X3 runs -- f32 in, sp32 out (see the kernel for its arithmetic). */
int avcer_attention(avcer_ctx* ctx, const void* qkv, void* out, int n, int s, int heads, int head_dim, float scale,
                    int in_kind, int out_kind, avcer_stream_t stream);

/* Measured ceilings of the GPU this context lives on (about 0.2 s): dense 16-bit MFMA issue rate of a register-only
 * v_mfma_f32_16x16x32_f16 loop in TFLOP/s (the instruction of AVCER_MODE_F16X3; the bf16 form issues at the same rate,
 * tools/f16_probe.hip), and the bandwidth of a 1 GiB -> 1 GiB 16-byte-per-lane copy in TB/s (bytes read + bytes written).
 * bench.py prints them next to the datasheet peaks it divides by.  Uses 2 GiB of workspace. */
int avcer_measure_ceilings(avcer_ctx* ctx, double* mfma16_tflops, double* hbm_copy_tbs, avcer_stream_t stream);

/* Last launch statistics of the dominant kernel (for bench.py's roofline object): number of conv_gemm
 * launches and their summed algorithmic FLOPs since the previous call to this function. */
int avcer_gemm_stats(avcer_ctx* ctx, int64_t* launches, double* flops, int reset);

/* Live timing of the dominant kernel for bench.py's roofline object: while enabled, every conv_gemm launch is
 * bracketed by a pair of HIP events on its launch stream; avcer_profile_read synchronises, returns the summed
 * event durations (ms) and the number of launches since the last read, and rewinds the event pool. */
int avcer_profile_enable(avcer_ctx* ctx, int on);
int avcer_profile_read(avcer_ctx* ctx, double* total_ms, int64_t* launches);
/* The same events by kernel family -- what bench.py's roofline.per_family prices each family with: summed event
 * milliseconds, launches, algorithmic FLOPs and compulsory HBM bytes (every operand read once, every output written once)
 * since avcer_profile_enable / the last read.  Arrays of n_fam <= AVCER_FAM_COUNT entries.  Use INSTEAD of
 * avcer_profile_read (both rewind the event pool). */
enum {
    AVCER_FAM_GEMM = 0,    /* conv_gemm_kernel: A and W tiles through LDS */
    AVCER_FAM_GEMM_WD = 1, /* conv_gemm_wd_kernel: weight fragments direct from global memory (dtype 7 / 8) */
    AVCER_FAM_CHAIN = 2,   /* bneck_kernel: fused bottleneck chains of ResNet stages 1-2 (HBM-bound) */
    AVCER_FAM_TAIL = 3,    /* bneck_tail2_kernel: conv3 + residual + next conv1 of stage 3 */
    AVCER_FAM_STEM = 4,    /* stem_pool(_u8)_kernel */
    AVCER_FAM_SKINNY = 5,  /* conv_gemm_skinny_kernel: one wave per tile, registers only (dtype 9 / 10; launches of few positions) */
    AVCER_FAM_COUNT = 6
};
int avcer_profile_read_families(avcer_ctx* ctx, int n_fam, double* ms, int64_t* launches, double* flops, double* bytes);
/* ... and launch by launch, in launch order (tools/wd_traffic.py matches this list with the dispatches of a rocprofv3 --pmc pass of
 * the same step): family (AVCER_FAM_*), event milliseconds, algorithmic FLOPs, compulsory HBM bytes and the contraction's
 * M, N, K (mnk [3 * max_n]) of up to max_n launches; *n = launches recorded since avcer_profile_enable / the last read.  Use
 * INSTEAD of the two reads above (all three rewind the event pool). */
int avcer_profile_read_launches(avcer_ctx* ctx, int64_t max_n, int32_t* fam, double* ms, double* flops, double* bytes, int64_t* mnk,
                                int64_t* n);

/* Debug aid for parity tests: arm a one-shot tap; the next forward pass copies up to `bytes` raw bytes of the
 * named intermediate activation (first sub-batch) into dst_dev.  Names: static "pre", "stem_conv", "stem",
 * "l1b0_c1", "l1b0_c2", "l1b0", "layer1".."layer4", "avgpool"; audio "norm", "conv0", "extract", "proj",
 * "posconv", "layer0".."layer11", "w2v", "tl1", "tl2", "td0", "mp", "td4", "pooled".  Activations are NHWC / time-major,
 * f32 in AVCER_MODE_FP32, bf16 in AVCER_MODE_BF16 and sp32 (fp16 hi / lo per 32 channels, see avcer_conv_gemm) in
 * AVCER_MODE_F16X3 (residual streams "proj".."tl2", "avgpool" and the heads are always f32).
 * What differs from the reference's tensors of the same name:
 *   - "layer1", "layer2", "layer3" hold the stage output on the grid the NEXT stage reads, i.e. the reference's tensor at
 *     [:, ::2, ::2] -- [n,28,28,256], [n,14,14,512], [n,7,7,1024]: the last bottleneck of stages 1-3 is evaluated at the
 *     even positions only (its other outputs feed nothing: the next stage's 1x1 convolutions have stride 2); "layer4" is
 *     the full [n,7,7,2048];
 *   - "pre", "stem_conv", "l1b0_c2" exist in AVCER_MODE_FP32 / _BF16 only: AVCER_MODE_F16X3 preprocesses inside the fused
 *     stem and keeps conv2 outputs in registers, so those taps do not fire there.
 * avcer_debug_tap_copied returns the number of bytes copied, or -1 if the tap did not fire (compare it with the size you
 * expect: a short copy means the tensor is smaller than the buffer, e.g. a sub-sampled stage tap). */
int avcer_debug_tap(avcer_ctx* ctx, const char* name, void* dst_dev, size_t bytes);
int64_t avcer_debug_tap_copied(const avcer_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* AVCER_HIP_H */
