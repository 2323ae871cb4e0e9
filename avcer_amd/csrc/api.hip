// C ABI of libavcer_hip.so: context, packed-weight loading, and the forward passes of the three AVCER models
// expressed as sequences of the kernels in gemm.hip / kernels.hip.  See include/avcer_hip.h for the contract.
#include "common.h"
#include "split_dev.h"

#include <algorithm>
#include <cmath>
#include <new>

// ------------------------------------------------------------------------------------------------ plumbing
int set_err(avcer_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

int ws_reserve(avcer_ctx* ctx, int slot, size_t bytes, void** out) {
    DevBuf& b = ctx->ws[slot];
    if (b.cap < bytes) {
        if (b.p) {
            HIP_TRY(ctx, hipDeviceSynchronize());
            HIP_TRY(ctx, hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        const size_t want = bytes + (bytes >> 3) + (1 << 20);
        if (hipMalloc(&b.p, want) != hipSuccess) {
            b.p = nullptr;
            (void)hipGetLastError();
            return set_err(ctx, AVCER_ENOMEM, "workspace slot %d: hipMalloc(%zu) failed", slot, want);
        }
        b.cap = want;
    }
    *out = b.p;
    return AVCER_OK;
}

namespace {

struct Arena {
    char* base;
    size_t off = 0, cap;
    Arena(void* p, size_t c) : base((char*)p), cap(c) {}
    void* get(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* r = base + off;
        off += bytes;
        return off <= cap ? r : nullptr;
    }
};

#pragma pack(push, 1)
struct BlobHeader {
    char magic[8];
    uint32_t count;
    uint32_t reserved;
};
struct BlobEntry {
    char name[96];
    uint32_t ndim;
    uint32_t pad;
    int64_t dims[4];
    uint64_t offset;
    uint64_t nbytes;
};
#pragma pack(pop)

void free_model(Model& m) {
    for (void* p : m.allocs) (void)hipFree(p);
    m.allocs.clear();
    m.t.clear();
    m.loaded = false;
}

int load_blob(avcer_ctx* ctx, Model& m, const void* blob, size_t nbytes) {
    if (!blob || nbytes < sizeof(BlobHeader)) return set_err(ctx, AVCER_EFORMAT, "weight blob too small");
    const BlobHeader* h = (const BlobHeader*)blob;
    if (memcmp(h->magic, "AVCERW01", 8) != 0) return set_err(ctx, AVCER_EFORMAT, "bad blob magic");
    const size_t table = sizeof(BlobHeader) + (size_t)h->count * sizeof(BlobEntry);
    if (table > nbytes) return set_err(ctx, AVCER_EFORMAT, "blob table truncated");
    const BlobEntry* e = (const BlobEntry*)((const char*)blob + sizeof(BlobHeader));
    size_t lo = nbytes, hi = 0;
    for (uint32_t i = 0; i < h->count; ++i) {
        if (e[i].offset % 64 || e[i].offset + e[i].nbytes > nbytes || e[i].offset < table)
            return set_err(ctx, AVCER_EFORMAT, "blob entry %u (%.95s) out of range", i, e[i].name);
        size_t numel = 1;
        for (uint32_t d = 0; d < e[i].ndim && d < 4; ++d) numel *= (size_t)e[i].dims[d];
        if (e[i].ndim > 4 || numel * 4 != e[i].nbytes)
            return set_err(ctx, AVCER_EFORMAT, "blob entry %u (%.95s) shape/size mismatch", i, e[i].name);
        lo = std::min(lo, (size_t)e[i].offset);
        hi = std::max(hi, (size_t)(e[i].offset + e[i].nbytes));
    }
    if (h->count == 0 || hi <= lo) return set_err(ctx, AVCER_EFORMAT, "empty blob");
    free_model(m);
    void* dev = nullptr;
    if (hipMalloc(&dev, hi - lo) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(ctx, AVCER_ENOMEM, "hipMalloc(%zu) for weights failed", hi - lo);
    }
    m.allocs.push_back(dev);
    HIP_TRY(ctx, hipMemcpy(dev, (const char*)blob + lo, hi - lo, hipMemcpyHostToDevice));
    for (uint32_t i = 0; i < h->count; ++i) {
        Tensor t;
        t.f32 = (float*)((char*)dev + (e[i].offset - lo));
        t.ndim = (int)e[i].ndim;
        t.numel = e[i].nbytes / 4;
        for (int d = 0; d < 4; ++d) t.dims[d] = e[i].dims[d];
        char nm[97];
        memcpy(nm, e[i].name, 96);
        nm[96] = 0;
        m.t[nm] = t;
    }
    m.loaded = true;
    return AVCER_OK;
}

// bf16 copies of every GEMM weight (names ending in ".w"), made once on the first bf16-mode call
int ensure_all_bf16(avcer_ctx* ctx, Model& m, hipStream_t st) {
    size_t total = 0;
    for (auto& kv : m.t)
        if (!kv.second.bf16 && kv.first.size() > 2 && kv.first.compare(kv.first.size() - 2, 2, ".w") == 0)
            total += (kv.second.numel * 2 + 255) & ~(size_t)255;
    if (!total) return AVCER_OK;
    void* dev = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(ctx, AVCER_ENOMEM, "hipMalloc(%zu) for bf16 weights failed", total);
    }
    m.allocs.push_back(dev);
    size_t off = 0;
    for (auto& kv : m.t)
        if (!kv.second.bf16 && kv.first.size() > 2 && kv.first.compare(kv.first.size() - 2, 2, ".w") == 0) {
            kv.second.bf16 = (bf16_t*)((char*)dev + off);
            TRY(k_f32_to_bf16(ctx, kv.second.f32, kv.second.bf16, kv.second.numel, st));
            off += (kv.second.numel * 2 + 255) & ~(size_t)255;
        }
    return AVCER_OK;
}

// split-fp16 copies (k_split_weight_rows: scaled hi/lo per 32-element K group + trailer, rows permuted inside every group of 32 output
// channels) of every GEMM weight whose K is a multiple of 32, made on the first x3 call
// device bytes of one split copy: the data, its scale trailer (split_dev.h), rounded to 256
inline size_t split_bytes(size_t numel) { return ((numel * 4 + 255) & ~(size_t)255) + AVCER_SPLIT_TRAILER_BYTES; }

int ensure_all_x3(avcer_ctx* ctx, Model& m, hipStream_t st) {
    auto wanted = [](const std::string& k, const Tensor& t) {
        const bool is_w = k.size() > 3 && (k.compare(k.size() - 2, 2, ".w") == 0 || k.compare(k.size() - 3, 3, ".wf") == 0);
        return !t.x3 && is_w && t.ndim == 2 && t.dims[1] % 32 == 0 && t.dims[0] % 64 == 0;
    };
    // fragment-order copy for the weights-direct kernel (dtype 7 / 8: N a multiple of 256, an even number of K-steps) and for
    // the skinny form (dtype 9 / 10: any shape the row split exists for)
    auto frag_ok = [](const Tensor&) { return true; };
    size_t total = 0;
    for (auto& kv : m.t)
        if (wanted(kv.first, kv.second)) total += split_bytes(kv.second.numel) * (frag_ok(kv.second) ? 2 : 1);
    if (!total) return AVCER_OK;
    void* dev = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(ctx, AVCER_ENOMEM, "hipMalloc(%zu) for split-fp16 weights failed", total);
    }
    m.allocs.push_back(dev);
    size_t off = 0;
    for (auto& kv : m.t)
        if (wanted(kv.first, kv.second)) {
            Tensor& t = kv.second;
            t.x3 = (bf16_t*)((char*)dev + off);
            // grouped weights ([groups*n][k], pos-conv) are stacked row blocks of multiples of 32 rows: same permutation
            TRY(k_split_weight_rows(ctx, t.f32, t.x3, (int)t.dims[0], (int)t.dims[1], st));
            off += split_bytes(t.numel);
            if (frag_ok(t)) {
                t.x3f = (bf16_t*)((char*)dev + off);
                TRY(k_weight_frags(ctx, t.x3, t.x3f, (int)t.dims[0], (int)t.dims[1], st));
                off += split_bytes(t.numel);
            }
        }
    return AVCER_OK;
}

// When the skinny form (conv_gemm dtype 9 / 10) serves a launch: while its grid of 32-position x 16-channel wave tiles fits the
// chip's SIMDs in one round.  tools/ab_layers.py --skinny at 1 / 4 / 16 / 64 frames and windows (profiles/r05_skinny_ab.txt):
// below that count it beats the better tiled form on every layer shape of both networks (by 1.05-3.8x), above it it loses.
static bool prefer_skinny(long M, int n, int block_slots) {
    return M <= 4096 && ((M + 31) / 32) * (long)(n / 16) <= 2L * block_slots;
}

// Which form of the x3 contraction serves a layer of M positions, N channels, K inputs: the weights-direct kernel
// (128 x 256 tiles, dtype 7 / 8) or the LDS-staged one (128 x 128, dtype 5 / 6).  Results are bit-identical, so this is
// speed only.  Model, calibrated on tools/ab_layers.py (profiles/r04_ab_layers*.txt; round 4's spread K loop moved it from
// 1.84-1.90): a 128 x 256 tile costs 1.72 tiles of 128 x 128 (qkv 1.73, out-proj 1.75, ffn1 1.69, l3.x.c2 1.68, fe6 1.72),
// a 112 x 256 tile 7/8 of that, and the grid takes common.h grid_rounds() rounds of the block slots.
bool prefer_weights_direct(long M, int N, long K, long slots) {
    (void)K;
    const long mt = (M + 127) / 128, mt112 = (M + 111) / 112;
    // the direct form also has a 112-row tile (gemm.hip launch_wd picks it by the same model)
    const double wd = std::min(grid_rounds(mt * (N / 256), slots), grid_rounds(mt112 * (N / 256), slots) * 112.0 / 128.0);
    return wd * 1.72 < grid_rounds(mt * (N / 128), slots);
}

struct Net {
    avcer_ctx* ctx;
    Model& m;
    int bf16;  // activation / weight type of the MFMA contractions
    hipStream_t st;
    int x3 = 0;  // f32-grade activations, split-fp16 MFMA (AVCER_MODE_F16X3)
    int err = AVCER_OK;

    const Tensor* T(const std::string& name) {
        auto it = m.t.find(name);
        if (it == m.t.end()) {
            if (err == AVCER_OK) err = set_err(ctx, AVCER_EFORMAT, "tensor '%s' missing from packed weights", name.c_str());
            return nullptr;
        }
        return &it->second;
    }
    const float* F(const std::string& name) {
        const Tensor* t = T(name);
        return t ? t->f32 : nullptr;
    }
    // contraction with weights `wname`; akind / okind = storage of the activations read / written
    // (0 = f32, 1 = bf16, 2 = sp32 pairs), which selects the kernel instantiation (avcer_conv_gemm dtype)
    void gemm(avcer_conv_desc d, const std::string& wname, const float* scale, const float* bias, const void* x,
              const void* res, void* y, int akind, int okind, const void* x2 = nullptr) {
        if (err != AVCER_OK) return;
        const Tensor* w = T(wname);
        if (!w) return;
        int dtype = -1;
        const void* wp = nullptr;
        const long K = (long)d.kh * d.kw * d.cin + (x2 ? d.x2_cin : 0);
        if (akind == 0 && okind == 0) {
            dtype = 0;
            wp = w->f32;
            if (x3 && w->x3) {
                dtype = 3;
                wp = w->x3;
            }
        } else if (akind == 1 && okind <= 1) {
            dtype = okind == 1 ? 1 : 2;
            wp = w->bf16;
        } else if (x3 && ((akind == 0 && okind == 2) || akind == 2) && okind != 1) {
            dtype = akind == 0 ? 4 : (okind == 2 ? 5 : 6);
            wp = w->x3;
            // sp32 activations: the weights-direct kernel wherever a fragment-order copy exists (one group, pad-free second source)
            const long M = (long)d.batch * d.out_h * d.out_w;
            if (akind == 2 && w->x3f && d.n % 256 == 0 && (K / 32) % 2 == 0 && d.groups <= 1 && (d.tile_n == 256 || (d.tile_n == 0 &&
                    prefer_weights_direct(M, d.n, K, ctx->block_slots)))) {
                dtype = okind == 2 ? 7 : 8;
                wp = w->x3f;
            }
            // a handful of positions (a frame or a window per call through the mirrors; the deep layers of a small batch): one
            // wave per (16-64) x 16 tile, registers only.  Bit-identical to the tiled forms, so the choice by M does not show in
            // any result.  tile_n is a hint for the tiled forms and does not apply.
            if (akind == 2 && w->x3f && !x2 && prefer_skinny(M, d.n * (d.groups > 1 ? d.groups : 1), ctx->block_slots)) {
                dtype = okind == 2 ? 9 : 10;
                wp = w->x3f;
                d.tile_n = 0;
                d.tile_m = 0;
            }
        }
        if (dtype < 0 || !wp) {
            err = set_err(ctx, AVCER_ESTATE, "gemm %s: storage kinds %d -> %d unsupported or weights not prepared",
                          wname.c_str(), akind, okind);
            return;
        }
        if ((long)w->numel != (long)d.n * K * (d.groups > 1 ? d.groups : 1)) {
            err = set_err(ctx, AVCER_EFORMAT, "gemm %s: weight has %zu elements, expected %ld x %ld", wname.c_str(),
                          w->numel, (long)d.n, K);
            return;
        }
        err = launch_conv_gemm(ctx, d, dtype, x, wp, scale, bias, res, y, st, x2);
    }
    void chk(int r) { if (err == AVCER_OK && r != AVCER_OK) err = r; }
    // debug tap: raw copy of min(requested, available) bytes of an intermediate tensor
    void tap(const char* name, const void* src, size_t bytes) {
        if (err != AVCER_OK || !ctx->tap_dst || ctx->tap_name != name) return;
        const size_t nb = std::min(bytes, ctx->tap_bytes);
        if (hipMemcpyAsync(ctx->tap_dst, src, nb, hipMemcpyDeviceToDevice, st) != hipSuccess)
            err = set_err(ctx, AVCER_EHIP, "debug tap %s: copy failed", name);
        ctx->tap_copied = (int64_t)nb;
        ctx->tap_dst = nullptr;
    }
};

avcer_conv_desc linear_desc(long m, int k, int n, int act) {
    avcer_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.batch = (int32_t)m; d.in_h = 1; d.in_w = 1; d.out_h = 1; d.out_w = 1;
    d.cin = k; d.kh = 1; d.kw = 1;
    d.stride_h = d.stride_w = 1; d.dil_h = d.dil_w = 1;
    d.x_stride_b = k; d.x_stride_h = k; d.x_stride_w = k;
    d.n = n; d.y_ld = n; d.r_ld = n; d.act = act;
    return d;
}

avcer_conv_desc conv2d_desc(int nb, int h, int w, int c, int kh, int kw, int stride, int pad, int n, int act) {
    avcer_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.batch = nb; d.in_h = h; d.in_w = w;
    d.out_h = (h + 2 * pad - kh) / stride + 1;
    d.out_w = (w + 2 * pad - kw) / stride + 1;
    d.cin = c; d.kh = kh; d.kw = kw;
    d.stride_h = d.stride_w = stride; d.pad_h = d.pad_w = pad; d.dil_h = d.dil_w = 1;
    d.x_stride_b = (int64_t)h * w * c; d.x_stride_h = (int64_t)w * c; d.x_stride_w = c;
    d.n = n; d.y_ld = n; d.r_ld = n; d.act = act;
    return d;
}

// Conv1d over a time-major [nb, len, c] tensor
avcer_conv_desc conv1d_desc(int nb, int len, int c, int k, int stride, int pad, int dil, int out_len, int n, int act) {
    avcer_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.batch = nb; d.in_h = len; d.in_w = 1; d.out_h = out_len; d.out_w = 1;
    d.cin = c; d.kh = k; d.kw = 1;
    d.stride_h = stride; d.stride_w = 1; d.pad_h = pad; d.pad_w = 0; d.dil_h = dil; d.dil_w = 1;
    d.x_stride_b = (int64_t)len * c; d.x_stride_h = c; d.x_stride_w = c;
    d.n = n; d.y_ld = n; d.r_ld = n; d.act = act;
    return d;
}

// Stage-3 tails (conv3 + residual + the next block's conv1) of at most this many positions run as two contractions instead of
// the fused bneck_tail2_kernel: bit-identical, and faster up to 64 frames per call (0.66 vs 0.90 ms at one frame, 2.39 vs 2.42
// at 64, 3.51 vs 3.44 at 128: tools/tail_pair_probe.py, profiles/experiments/r05_tail_pair_probe.txt)
constexpr long kTailPairRows = 16384;
constexpr int kStages[4][3] = {{64, 3, 1}, {128, 4, 2}, {256, 6, 2}, {512, 3, 2}};  // video.py:105-108,165

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI: context
extern "C" int avcer_abi_version(void) { return AVCER_ABI_VERSION; }

extern "C" int avcer_ctx_create(int device, avcer_ctx** out) {
    if (!out) return AVCER_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return AVCER_EHIP;
    if (hipSetDevice(device) != hipSuccess) return AVCER_EHIP;
    avcer_ctx* c = new (std::nothrow) avcer_ctx();
    if (!c) return AVCER_ENOMEM;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->block_slots = 2 * prop.multiProcessorCount;  // two 256-thread MFMA blocks per CU (partitioned modes have fewer CUs)
    if (hipMalloc((void**)&c->ovf, 256) != hipSuccess || hipMemset(c->ovf, 0, 256) != hipSuccess) {
        (void)hipGetLastError();
        if (c->ovf) (void)hipFree(c->ovf);
        delete c;
        return AVCER_ENOMEM;
    }
    *out = c;
    return AVCER_OK;
}

extern "C" void avcer_ctx_destroy(avcer_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_model(ctx->stat);
    free_model(ctx->dyn);
    free_model(ctx->aud);
    free_model(ctx->face);
    for (auto& b : ctx->ws)
        if (b.p) (void)hipFree(b.p);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->ovf) (void)hipFree(ctx->ovf);
    delete ctx;
}

extern "C" const char* avcer_last_error(const avcer_ctx* ctx) { return ctx ? ctx->err : "null context"; }

// The run-time signal of the x3 mode's range contract (split_dev.h): how many threads have turned a FINITE activation of
// magnitude >= 65520 into an infinite fp16 hi half since the last reset -- every forward pass of this context, every mode-2
// kernel-level entry.  0 = every NaN in an output came in through the input (the reference's empty audio window); > 0 = the
// outputs since the last reset are not to be trusted: run the call again in AVCER_MODE_FP32.  Waits for `stream`.
extern "C" int avcer_x3_overflow_count(avcer_ctx* ctx, int reset, int64_t* count, avcer_stream_t stream) {
    if (!ctx || !count) return ctx ? set_err(ctx, AVCER_EINVAL, "x3_overflow_count: null count") : AVCER_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned host = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&host, ctx->ovf, sizeof(host), hipMemcpyDeviceToHost, st));
    if (reset) HIP_TRY(ctx, hipMemsetAsync(ctx->ovf, 0, sizeof(host), st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *count = (int64_t)host;
    return AVCER_OK;
}

extern "C" int avcer_load_static(avcer_ctx* ctx, const void* blob, size_t nbytes) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return load_blob(ctx, ctx->stat, blob, nbytes);
}
extern "C" int avcer_load_dynamic(avcer_ctx* ctx, const void* blob, size_t nbytes) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return load_blob(ctx, ctx->dyn, blob, nbytes);
}
extern "C" int avcer_load_audio(avcer_ctx* ctx, const void* blob, size_t nbytes) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TRY(load_blob(ctx, ctx->aud, blob, nbytes));
    auto it = ctx->aud.t.find("fd.w");
    if (it == ctx->aud.t.end()) return set_err(ctx, AVCER_EFORMAT, "audio blob lacks fd.w");
    ctx->aud_classes = (int)it->second.dims[0];
    return AVCER_OK;
}
extern "C" int avcer_audio_num_classes(const avcer_ctx* ctx) { return ctx ? ctx->aud_classes : 0; }

// ------------------------------------------------------------------------------------------------ static CNN
// ref: architectures/video.py:93-166 (ResNet-50, stride on the first 1x1 of a bottleneck, BN eps 1e-3 folded into
// per-channel scale/bias by avcer_amd/packing.py), data/utils.py:19-39, get_prob_video.py:47-49,103-112.
static int static_forward_impl(avcer_ctx* ctx, const uint8_t* frames, const float* nchw, int n, int in_h, int in_w, int mode,
                               float* logits, float* probs, float* feats, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!ctx->stat.loaded) return set_err(ctx, AVCER_ESTATE, "static weights not loaded");
    if ((!frames && !nchw) || n <= 0 || in_h <= 0 || in_w <= 0) return set_err(ctx, AVCER_EINVAL, "static_forward: bad arguments");
    if (mode < AVCER_MODE_FP32 || mode > AVCER_MODE_F16X3) return set_err(ctx, AVCER_EINVAL, "static_forward: mode %d", mode);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int bf = mode == AVCER_MODE_BF16;
    const int act = bf ? 1 : (mode == AVCER_MODE_F16X3 ? 2 : 0);  // activation storage: f32 / bf16 / sp32 pairs
    if (bf) TRY(ensure_all_bf16(ctx, ctx->stat, st));
    if (mode == AVCER_MODE_F16X3) TRY(ensure_all_x3(ctx, ctx->stat, st));
    const size_t es = bf ? 2 : 4;
    // Two granularities.  The FRONT (stem, stage 1, first block of stage 2: the 55x55 tensors) runs in passes of NB <= 1024
    // frames, the 4 GiB range of a buffer descriptor at 4 bytes per element.  The BACK (rest of stage 2, stages 3-4, tail)
    // runs once over NS = up to two front passes: its grids are small (stage 4: 1568 tiles per 1024 frames on 512 block
    // slots), and twice the rows per launch means a smaller last partial round (a tail of 0.06 rounds costs a whole one).
    const int NB = std::min(n, ctx->static_batch);
    const int NS = std::min(n, 2 * NB);
    const size_t act_elems = (size_t)NB * 112 * 112 * 64;  // largest activation of the front (stem output)
    const size_t back_elems = (size_t)NS * 28 * 28 * 512;  // largest activation of the back (stage 2)
    const size_t pre_elems = (size_t)NB * 230 * 230 * 4;
    if (back_elems * es >= 0xF0000000ull) return set_err(ctx, AVCER_EINVAL, "static_forward: %d frames per back pass", NS);
    const size_t total = pre_elems * es + 4 * (act_elems * es + 256) + 4 * (back_elems * es + 256) +
                         (size_t)NS * (2048 * 2 + 512) * 4 + 4096;
    void* wsp = nullptr;
    TRY(ws_reserve(ctx, 0, total, &wsp));
    Arena ar(wsp, ctx->ws[0].cap);
    void* P = ar.get(pre_elems * es);
    void* buf[4];
    for (auto& b : buf) b = ar.get(act_elems * es);
    void* bbuf[4];
    for (auto& b : bbuf) b = ar.get(back_elems * es);
    float* pooled = (float*)ar.get((size_t)NS * 2048 * 4);
    void* pooled_sp = ar.get((size_t)NS * 2048 * 4);
    float* feat_ws = (float*)ar.get((size_t)NS * 512 * 4);
    if (!feat_ws) return set_err(ctx, AVCER_ENOMEM, "static workspace arithmetic");

    Net net{ctx, ctx->stat, bf, st, mode == AVCER_MODE_F16X3};
    // stem + max-pool of `nb` frames starting at frame f0 of the call: -> B[1] (55x55x64), B[0] is scratch
    auto run_stem = [&](int f0, int nb, void** B) {
        if (net.x3) {
            // x3 mode: ONE kernel for conv 7x7/2 + BN + ReLU + max-pool 3x3/2 (fused.hip).  From u8 frames it also does the
            // preprocessing (raw pixels are exact in fp16: two MFMAs per product, the mean lives in the shift table stem.b9);
            // the preprocessed-tensor entry point carries arbitrary floats and goes through the planar fp16 hi / lo image.
            const Tensor* w = net.T("stem7.w");
            if (net.err != AVCER_OK) return;
            if (!w->x3) { net.err = set_err(ctx, AVCER_ESTATE, "stem7.w: split weights not prepared"); return; }
            if (frames) {
                net.chk(launch_stem_pool_u8(ctx, frames + (size_t)f0 * in_h * in_w * 3, in_h, in_w, w->x3, net.F("stem.s"),
                                            net.F("stem.b9"), B[1], nb, st));
                net.tap("stem", B[1], (size_t)nb * 55 * 55 * 64 * es);
                return;
            }
            net.chk(k_pack_nchw(ctx, nchw + (size_t)f0 * 3 * 224 * 224, nb, P, 3, st));
            net.chk(launch_stem_pool(ctx, P, (size_t)nb * 230 * 230 * 4 * 2, w->x3, net.F("stem.s"), net.F("stem.b"), B[1], nb, st));
            net.tap("stem", B[1], (size_t)nb * 55 * 55 * 64 * es);
            return;
        }
        if (frames) net.chk(k_preprocess(ctx, frames + (size_t)f0 * in_h * in_w * 3, nb, in_h, in_w, P, bf, st));
        else net.chk(k_pack_nchw(ctx, nchw + (size_t)f0 * 3 * 224 * 224, nb, P, bf, st));
        // 8 tap rows x (8 pixels x 4 channels) over the zero-bordered 230x230x4 image, stride 2
        avcer_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.batch = nb; d.in_h = 230; d.in_w = 230; d.out_h = 112; d.out_w = 112;
        d.cin = 32; d.kh = 8; d.kw = 1;
        d.stride_h = 2; d.stride_w = 2; d.dil_h = d.dil_w = 1;
        d.x_stride_b = 230L * 230 * 4; d.x_stride_h = 230 * 4; d.x_stride_w = 4;
        d.n = 64; d.y_ld = 64; d.r_ld = 64; d.act = 1;
        net.gemm(d, "stem.w", net.F("stem.s"), net.F("stem.b"), P, nullptr, B[0], bf, act);  // P is f32 unless bf16
        net.tap("pre", P, (size_t)nb * 230 * 230 * 4 * es);
        net.tap("stem_conv", B[0], (size_t)nb * 112 * 112 * 64 * es);
        net.chk(k_maxpool3s2(ctx, B[0], B[1], nb, 112, 112, 64, 55, 55, act, st));
        net.tap("stem", B[1], (size_t)nb * 55 * 55 * 64 * es);
    };
    // blocks [b_begin, b_end) of stage li on nb frames; the last of them writes to `last_out` when given.
    //
    // Where the stride lives.  The reference strides the FIRST block of stages 2-4: its conv1 and its downsample convolution
    // are 1x1, stride 2, no padding (video.py:12-19,140-149), so of the previous stage's output only the positions (2 oy, 2 ox)
    // are ever read.  Here the LAST block of stages 1-3 is evaluated at those positions alone (sub = 2: conv2 becomes a
    // stride-2 3x3 over the full-resolution conv1 output, conv3 adds the residual row of the even position) and writes the
    // compact [nb, oh, oh, 4 planes] tensor; the next stage's first block then reads it at stride 1.  Same values at every
    // position anyone reads, 9.6 % fewer products per frame (0.74 of 7.67 GFLOP) and a quarter of that block's trunk bytes.
    auto run_stage = [&](int li, int nb, void*& X, void*& T1, void*& T2, void*& OUT, int& h, int& cin, void* last_out, int b_begin,
                         int b_end) {
        const int planes = kStages[li][0], blocks = kStages[li][1];
        const bool chain = net.x3 && li < 2;  // stages 1-2 in x3 mode: non-first blocks run as the fused chain (fused.hip)
        for (int b = b_begin; b < b_end; ++b) {
            const int stride = 1;  // kStages[li][2] was applied by the previous stage's last block (sub below)
            const int sub = (b == blocks - 1 && li < 3) ? kStages[li + 1][2] : 1;
            const std::string p = "l" + std::to_string(li + 1) + "." + std::to_string(b) + ".";
            const int oh = (h - 1) / sub + 1;
            void* dst = (b == b_end - 1 && last_out) ? last_out : OUT;
            if (net.x3 && li == 2 && b >= 1) {
                // stage 3: conv2 stays a conv_gemm launch; conv3 + residual and the next block's conv1 share one (fused.hip)
                if (b == 1)
                    net.gemm(conv2d_desc(nb, h, h, cin, 1, 1, 1, 0, planes, 1), p + "c1.w", net.F(p + "c1.s"), net.F(p + "c1.b"),
                             X, nullptr, T1, act, act);
                net.gemm(conv2d_desc(nb, h, h, planes, 3, 3, sub, 1, planes, 1), p + "c2.w", net.F(p + "c2.s"), net.F(p + "c2.b"), T1,
                         nullptr, T2, act, act);
                if (b + 1 < blocks && (long)nb * h * h <= kTailPairRows) {
                    // few positions: the same two contractions (same folded weights, same arithmetic: bit-identical) as two
                    // launches of the skinny / tiled forms -- the fused kernel's 128-row blocks leave the chip empty here
                    const std::string pn = "l" + std::to_string(li + 1) + "." + std::to_string(b + 1) + ".";
                    net.gemm(conv2d_desc(nb, h, h, planes, 1, 1, 1, 0, planes * 4, 1), p + "c3.wf", nullptr, net.F(p + "c3.b"), T2, X,
                             dst, act, act);
                    net.gemm(conv2d_desc(nb, h, h, planes * 4, 1, 1, 1, 0, planes, 1), pn + "c1.wf", nullptr, net.F(pn + "c1.b"), dst,
                             nullptr, T1, act, act);
                } else if (b + 1 < blocks) {
                    const std::string pn = "l" + std::to_string(li + 1) + "." + std::to_string(b + 1) + ".";
                    const Tensor *w3 = net.T(p + "c3.wf"), *w1n = net.T(pn + "c1.wf");
                    if (net.err != AVCER_OK) return;
                    if (!w3->x3 || !w1n->x3) {
                        net.err = set_err(ctx, AVCER_ESTATE, "%s: split tail weights not prepared", p.c_str());
                        return;
                    }
                    net.chk(launch_bneck_tail(ctx, planes, (long)nb * h * h, T2, X, dst, T1, w3->x3, net.F(p + "c3.b"), w1n->x3,
                                              net.F(pn + "c1.b"), st));
                } else {
                    avcer_conv_desc d3 = conv2d_desc(nb, oh, oh, planes, 1, 1, 1, 0, planes * 4, 1);
                    d3.r_sub = sub; d3.r_h = h; d3.r_w = h;
                    net.gemm(d3, p + "c3.w", net.F(p + "c3.s"), net.F(p + "c3.b"), T2, X, dst, act, act);
                }
                if (dst == OUT) std::swap(X, OUT);
                else X = dst;
                h = oh;
                continue;
            }
            // li == 0: the first block too (stride 1, 64 input channels: the downsample operand rides in registers)
            if (chain && (b >= 1 || li == 0)) {
                const bool first = b == 0;
                // T1 of this block: written by the previous chain launch, or by a standalone conv1 at the head of a chain
                if (first || (b == 1 && li != 0))
                    net.gemm(conv2d_desc(nb, h, h, cin, 1, 1, 1, 0, planes, 1), p + "c1.w", net.F(p + "c1.s"), net.F(p + "c1.b"),
                             X, nullptr, T1, act, act);
                const bool next = b + 1 < blocks;
                const std::string pn = "l" + std::to_string(li + 1) + "." + std::to_string(b + 1) + ".";
                const Tensor *w2 = net.T(p + "c2.wf"), *w3 = net.T(first ? p + "c3d.w" : p + "c3.wf"),
                             *w1n = next ? net.T(pn + "c1.wf") : nullptr;
                if (net.err != AVCER_OK) return;
                if (!w2->x3 || !w3->x3 || (next && !w1n->x3)) {
                    net.err = set_err(ctx, AVCER_ESTATE, "%s: split chain weights not prepared", p.c_str());
                    return;
                }
                net.chk(launch_bneck(ctx, planes, nb, h, h, T1, X, first ? cin : 0, sub, dst, next ? T2 : nullptr, w2->x3,
                                     net.F(p + "c2.b"), w3->x3, net.F(first ? p + "c3d.b" : p + "c3.b"), next ? w1n->x3 : nullptr,
                                     next ? net.F(pn + "c1.b") : nullptr, st, w2->x3f));
                std::swap(T1, T2);  // the next block's T1 was written into T2
                if (dst == OUT) std::swap(X, OUT);
                else X = dst;
                cin = planes * 4;
                h = oh;
                if (first) {
                    net.tap("l1b0_c1", T2, (size_t)nb * h * h * planes * es);  // after the swap T2 holds this block's conv1 output
                    net.tap("l1b0", X, (size_t)nb * h * h * cin * es);
                }
                continue;
            }
            net.gemm(conv2d_desc(nb, h, h, cin, 1, 1, stride, 0, planes, 1), p + "c1.w", net.F(p + "c1.s"),
                     net.F(p + "c1.b"), X, nullptr, T1, act, act);
            net.gemm(conv2d_desc(nb, h, h, planes, 3, 3, sub, 1, planes, 1), p + "c2.w", net.F(p + "c2.s"),
                     net.F(p + "c2.b"), T1, nullptr, T2, act, act);
            if (b == 0) {
                // conv3 + downsample fused: K = [T2 (planes) | X at stride (cin)], BN scales folded into the weights
                avcer_conv_desc d = conv2d_desc(nb, oh, oh, planes, 1, 1, 1, 0, planes * 4, 1);
                d.x2_cin = cin; d.x2_stride = stride;
                d.x2_stride_b = (int64_t)h * h * cin; d.x2_stride_h = (int64_t)h * cin; d.x2_stride_w = cin;
                net.gemm(d, p + "c3d.w", nullptr, net.F(p + "c3d.b"), T2, nullptr, dst, act, act, X);
            } else {
                avcer_conv_desc d3 = conv2d_desc(nb, oh, oh, planes, 1, 1, 1, 0, planes * 4, 1);
                d3.r_sub = sub; d3.r_h = h; d3.r_w = h;
                net.gemm(d3, p + "c3.w", net.F(p + "c3.s"), net.F(p + "c3.b"), T2, X, dst, act, act);
            }
            if (dst == OUT) std::swap(X, OUT);
            else X = dst;
            h = oh;
            cin = planes * 4;
            if (li == 0 && b == 0) {
                net.tap("l1b0_c1", T1, (size_t)nb * h * h * planes * es);
                net.tap("l1b0_c2", T2, (size_t)nb * h * h * planes * es);
                net.tap("l1b0", X, (size_t)nb * h * h * cin * es);
            }
        }
        if (b_end == blocks) net.tap(("layer" + std::to_string(li + 1)).c_str(), X, (size_t)nb * h * h * cin * es);
    };
    for (int s0 = 0; s0 < n; s0 += NS) {
        const int nb = std::min(NS, n - s0);
        for (int c0 = 0; c0 < nb; c0 += NB) {  // front passes: stem, stage 1, first block of stage 2 -> bbuf[0]
            const int cn = std::min(NB, nb - c0);
            void *X, *T1 = buf[2], *T2 = buf[3], *OUT = buf[0];
            int h = 55, cin = 64;
            run_stem(s0 + c0, cn, buf);
            X = buf[1];
            run_stage(0, cn, X, T1, T2, OUT, h, cin, nullptr, 0, kStages[0][1]);
            run_stage(1, cn, X, T1, T2, OUT, h, cin, (char*)bbuf[0] + (size_t)c0 * 28 * 28 * 512 * es, 0, 1);
        }
        void *X = bbuf[0], *T1 = bbuf[2], *T2 = bbuf[3], *OUT = bbuf[1];
        int h = 28, cin = 512;
        run_stage(1, nb, X, T1, T2, OUT, h, cin, nullptr, 1, kStages[1][1]);
        for (int li = 2; li < 4; ++li) run_stage(li, nb, X, T1, T2, OUT, h, cin, nullptr, 0, kStages[li][1]);
        // x3 mode: the pooled features once more as sp32 pairs, fc1's operand (the split fc1 did on the fly before)
        net.chk(k_avgpool_hw(ctx, X, pooled, net.x3 ? pooled_sp : nullptr, nb, h * h, 2048, act, st));
        net.tap("avgpool", pooled, (size_t)nb * 2048 * 4);
        float* fo = feats ? feats + (size_t)s0 * 512 : feat_ws;
        {
            // fc1 in the mode's own arithmetic.  Rounds 2-3 ran it on the f32 MFMA in the x3 mode as well, because a bf16-pair
            // fc1 put 40 % on top of the pooled features' error; the fp16-pair contraction carries 7e-8 and costs a quarter
            // of the time of the f32 MFMA on this launch-latency-bound layer (16 tiles at batch 256: 114 -> ~30 us).
            avcer_conv_desc fd = linear_desc(nb, 2048, 512, 0);
            fd.tile_n = 64;  // 128 x 64 tiles: twice the blocks of a grid that does not fill the chip anyway
            if (net.x3) net.gemm(fd, "fc1.w", nullptr, net.F("fc1.b"), pooled_sp, nullptr, fo, 2, 0);
            else net.gemm(fd, "fc1.w", nullptr, net.F("fc1.b"), pooled, nullptr, fo, 0, 0);
        }
        if (logits || probs)
            net.chk(k_small_linear(ctx, fo, net.F("fc2.w"), net.F("fc2.b"), logits ? logits + (size_t)s0 * 7 : nullptr,
                                   probs ? probs + (size_t)s0 * 7 : nullptr, nb, 512, 7, 1, st));
        if (net.err != AVCER_OK) return net.err;
    }
    return net.err;
}

extern "C" int avcer_static_forward(avcer_ctx* ctx, const uint8_t* frames, int n, int in_h, int in_w, int mode,
                                    float* logits, float* probs, float* feats, avcer_stream_t stream) {
    if (!frames) return ctx ? set_err(ctx, AVCER_EINVAL, "static_forward: null frames") : AVCER_EINVAL;
    return static_forward_impl(ctx, frames, nullptr, n, in_h, in_w, mode, logits, probs, feats, stream);
}

extern "C" int avcer_static_forward_nchw(avcer_ctx* ctx, const float* x, int n, int mode, float* logits, float* probs,
                                         float* feats, avcer_stream_t stream) {
    if (!x) return ctx ? set_err(ctx, AVCER_EINVAL, "static_forward_nchw: null input") : AVCER_EINVAL;
    return static_forward_impl(ctx, nullptr, x, n, 224, 224, mode, logits, probs, feats, stream);
}

// ------------------------------------------------------------------------------------------------ RetinaFace-R50 (row f4)
// ref: data/face_detection/ibug/face_detection/retina_face/retina_face.py:46-115 (RetinaFace, test phase),
// retina_face_net.py:42-101 (SSH, FPN), torchvision ResNet-50 as the body (children conv1..layer4, return_layers
// layer2/3/4), retina_face_predictor.py:59-65 (mean subtraction).  Inference BatchNorm (eps 1e-5) is folded by
// avcer_amd/packing.py; the three 1x1 heads of a pyramid level are one GEMM with 32 (+32 padding) output channels.
static int face_forward_impl(avcer_ctx* ctx, const uint8_t* frames, int n, int H, int W, int rgb, int mode, float* loc,
                             float* conf, float* landms, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!ctx->face.loaded) return set_err(ctx, AVCER_ESTATE, "face detector weights not loaded");
    if (!frames || !loc || !conf || !landms || n <= 0 || H < 32 || W < 32)
        return set_err(ctx, AVCER_EINVAL, "face_forward: bad arguments (frames of at least 32x32)");
    if (mode < AVCER_MODE_FP32 || mode > AVCER_MODE_F16X3) return set_err(ctx, AVCER_EINVAL, "face_forward: mode %d", mode);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int bf = mode == AVCER_MODE_BF16;
    const int act = bf ? 1 : (mode == AVCER_MODE_F16X3 ? 2 : 0);
    if (bf) TRY(ensure_all_bf16(ctx, ctx->face, st));
    if (mode == AVCER_MODE_F16X3) TRY(ensure_all_x3(ctx, ctx->face, st));
    const size_t es = bf ? 2 : 4;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;       // conv1 7x7/2, padding 3
    const int PH = 2 * (OH - 1) + 8, PW = 2 * (OW - 1) + 8;      // zero-bordered image the 8x(8x4)-tap stem reads
    const int MH = (OH - 1) / 2 + 1, MW = (OW - 1) / 2 + 1;      // max-pool 3x3/2, padding 1
    int fh[5], fw[5];                                             // output extent of layer1..4 (index 1..4)
    fh[1] = MH; fw[1] = MW;
    for (int li = 2; li <= 4; ++li) { fh[li] = (fh[li - 1] - 1) / 2 + 1; fw[li] = (fw[li - 1] - 1) / 2 + 1; }
    const int P = 2 * (fh[2] * fw[2] + fh[3] * fw[3] + fh[4] * fw[4]);
    // frames per pass: every gathered operand must stay below the 4 GiB buffer-descriptor range
    const size_t per_frame = std::max({(size_t)PH * PW * 4, (size_t)OH * OW * 64, (size_t)MH * MW * 256}) * 4;
    const int NB = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, (size_t)0xF0000000u / per_frame));
    // the largest activation: the stem output or layer1's output (equal for even extents, the latter larger for odd ones)
    const size_t big = (size_t)NB * std::max((size_t)OH * OW * 64, (size_t)MH * MW * 256);
    size_t feat_el[5] = {0, 0, 0, 0, 0};
    for (int li = 2; li <= 4; ++li) feat_el[li] = (size_t)NB * fh[li] * fw[li] * kStages[li - 1][0] * 4;
    const size_t lvl = (size_t)NB * fh[2] * fw[2];                // positions of the finest pyramid level
    size_t total = (size_t)NB * PH * PW * 4 * es + 4 * (big * es + 256) + 4096;
    for (int li = 2; li <= 4; ++li) total += feat_el[li] * es + 256;
    total += 5 * (lvl * 256 * es + 256) + 2 * (lvl * 64 * es + 256) + lvl * 64 * 4 + 256;
    void* wsp = nullptr;
    TRY(ws_reserve(ctx, 3, total, &wsp));
    Arena ar(wsp, ctx->ws[3].cap);
    void* Pimg = ar.get((size_t)NB * PH * PW * 4 * es);
    void* buf[4];
    for (auto& b : buf) b = ar.get(big * es);
    void* feat[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int li = 2; li <= 4; ++li) feat[li] = ar.get(feat_el[li] * es);
    void* pyr[3];      // FPN lateral outputs, finest first
    for (auto& b : pyr) b = ar.get(lvl * 256 * es);
    void* mrg = ar.get(lvl * 256 * es);   // merged level (output of merge1 / merge2)
    void* S = ar.get(lvl * 256 * es);     // SSH output
    void* t51 = ar.get(lvl * 64 * es);
    void* t72 = ar.get(lvl * 64 * es);
    float* hd = (float*)ar.get(lvl * 64 * 4);
    if (!hd) return set_err(ctx, AVCER_ENOMEM, "face workspace arithmetic");

    Net net{ctx, ctx->face, bf, st, mode == AVCER_MODE_F16X3};
    for (int s0 = 0; s0 < n; s0 += NB) {
        const int nb = std::min(NB, n - s0);
        net.chk(k_face_pre(ctx, frames + (size_t)s0 * H * W * 3, nb, H, W, PH, PW, rgb, Pimg, bf, st));
        {
            avcer_conv_desc d;
            memset(&d, 0, sizeof(d));
            d.batch = nb; d.in_h = PH; d.in_w = PW; d.out_h = OH; d.out_w = OW;
            d.cin = 32; d.kh = 8; d.kw = 1;
            d.stride_h = 2; d.stride_w = 2; d.dil_h = d.dil_w = 1;
            d.x_stride_b = (int64_t)PH * PW * 4; d.x_stride_h = (int64_t)PW * 4; d.x_stride_w = 4;
            d.n = 64; d.y_ld = 64; d.r_ld = 64; d.act = 1;
            net.gemm(d, "stem.w", net.F("stem.s"), net.F("stem.b"), Pimg, nullptr, buf[0], bf, act);
        }
        net.chk(k_maxpool3s2p1(ctx, buf[0], buf[1], nb, OH, OW, 64, MH, MW, act, st));
        net.tap("face_pool", buf[1], (size_t)nb * MH * MW * 64 * es);
        void *X = buf[1], *T1 = buf[2], *T2 = buf[3];
        int h = MH, w = MW, cin = 64;
        for (int li = 1; li <= 4; ++li) {
            const int planes = kStages[li - 1][0], blocks = kStages[li - 1][1];
            for (int b = 0; b < blocks; ++b) {
                const int stride = b == 0 ? kStages[li - 1][2] : 1;   // torchvision: the stride sits on the 3x3 convolution
                const std::string p = "l" + std::to_string(li) + "." + std::to_string(b) + ".";
                const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
                // block output: the ping-pong buffer the input does not occupy, or the kept feature map of the stage
                void* dst = (b == blocks - 1 && li >= 2) ? feat[li] : (X == buf[0] ? buf[1] : buf[0]);
                avcer_conv_desc d1 = conv2d_desc(nb, h, h, cin, 1, 1, 1, 0, planes, 1);
                d1.in_w = w; d1.out_w = w; d1.x_stride_b = (int64_t)h * w * cin; d1.x_stride_h = (int64_t)w * cin;
                net.gemm(d1, p + "c1.w", net.F(p + "c1.s"), net.F(p + "c1.b"), X, nullptr, T1, act, act);
                avcer_conv_desc d2 = conv2d_desc(nb, h, h, planes, 3, 3, stride, 1, planes, 1);
                d2.in_w = w; d2.out_w = ow; d2.x_stride_b = (int64_t)h * w * planes; d2.x_stride_h = (int64_t)w * planes;
                net.gemm(d2, p + "c2.w", net.F(p + "c2.s"), net.F(p + "c2.b"), T1, nullptr, T2, act, act);
                avcer_conv_desc d3 = conv2d_desc(nb, oh, oh, planes, 1, 1, 1, 0, planes * 4, 1);
                d3.in_w = ow; d3.out_w = ow; d3.x_stride_b = (int64_t)oh * ow * planes; d3.x_stride_h = (int64_t)ow * planes;
                if (b == 0) {
                    d3.x2_cin = cin; d3.x2_stride = stride;
                    d3.x2_stride_b = (int64_t)h * w * cin; d3.x2_stride_h = (int64_t)w * cin; d3.x2_stride_w = cin;
                    net.gemm(d3, p + "c3d.w", nullptr, net.F(p + "c3d.b"), T2, nullptr, dst, act, act, X);
                } else {
                    net.gemm(d3, p + "c3.w", net.F(p + "c3.s"), net.F(p + "c3.b"), T2, X, dst, act, act);
                }
                X = dst;
                net.tap(("face_blk" + std::to_string(li) + "_" + std::to_string(b)).c_str(), X, (size_t)nb * oh * ow * planes * 4 * es);
                if (li == 1 && b == 0) {
                    net.tap("face_l1b0_c1", T1, (size_t)nb * oh * ow * planes * es);
                    net.tap("face_l1b0_c2", T2, (size_t)nb * oh * ow * planes * es);
                    net.tap("face_l1b0", X, (size_t)nb * oh * ow * planes * 4 * es);
                }
                h = oh; w = ow; cin = planes * 4;
            }
            if (li >= 2) net.tap(("face_body" + std::to_string(li - 1)).c_str(), X, (size_t)nb * h * w * cin * es);
            else net.tap("face_layer1", X, (size_t)nb * h * w * cin * es);
        }
        // FPN (retina_face_net.py:87-101): laterals, then top-down nearest-upsample + add + 3x3 merge
        for (int i = 0; i < 3; ++i) {
            const int li = i + 2, c = kStages[li - 1][0] * 4;
            avcer_conv_desc d = conv2d_desc(nb, fh[li], fh[li], c, 1, 1, 1, 0, 256, 1);
            d.in_w = fw[li]; d.out_w = fw[li]; d.x_stride_b = (int64_t)fh[li] * fw[li] * c; d.x_stride_h = (int64_t)fw[li] * c;
            const std::string p = "fpn.o" + std::to_string(i + 1) + ".";
            net.gemm(d, p + "w", net.F(p + "s"), net.F(p + "b"), feat[li], nullptr, pyr[i], act, act);
        }
        auto conv3 = [&](const std::string& p, const void* x, int hh, int ww, int c, void* y, int nout, int y_ld, int y_coff,
                         int relu, int okind) {
            avcer_conv_desc d = conv2d_desc(nb, hh, hh, c, 3, 3, 1, 1, nout, relu);
            d.in_w = ww; d.out_w = ww; d.x_stride_b = (int64_t)hh * ww * c; d.x_stride_h = (int64_t)ww * c;
            d.y_ld = y_ld; d.y_coff = y_coff;
            net.gemm(d, p + "w", net.F(p + "s"), net.F(p + "b"), x, nullptr, y, act, okind);
        };
        void* lvl_in[3] = {nullptr, nullptr, pyr[2]};
        net.tap("face_lat1", pyr[0], (size_t)nb * fh[2] * fw[2] * 256 * es);
        net.tap("face_lat2", pyr[1], (size_t)nb * fh[3] * fw[3] * 256 * es);
        net.tap("face_lat3", pyr[2], (size_t)nb * fh[4] * fw[4] * 256 * es);
        net.chk(k_upsample_add(ctx, pyr[1], pyr[2], nb, fh[3], fw[3], fh[4], fw[4], 256, act, st));
        void* m2 = T1;   // the ping-pong buffers of the body are free again
        net.tap("face_sum2", pyr[1], (size_t)nb * fh[3] * fw[3] * 256 * es);
        conv3("fpn.m2.", pyr[1], fh[3], fw[3], 256, m2, 256, 256, 0, 1, act);
        net.tap("face_fpn2", m2, (size_t)nb * fh[3] * fw[3] * 256 * es);
        net.chk(k_upsample_add(ctx, pyr[0], m2, nb, fh[2], fw[2], fh[3], fw[3], 256, act, st));
        conv3("fpn.m1.", pyr[0], fh[2], fw[2], 256, mrg, 256, 256, 0, 1, act);
        lvl_in[0] = mrg; lvl_in[1] = m2;
        net.tap("face_fpn1", mrg, (size_t)nb * fh[2] * fw[2] * 256 * es);
        // SSH + heads per level (retina_face_net.py:59-73, retina_face.py:101-113)
        int row0 = 0;
        for (int i = 0; i < 3; ++i) {
            const int hh = fh[i + 2], ww = fw[i + 2];
            const std::string p = "ssh" + std::to_string(i + 1) + ".";
            conv3(p + "c3.", lvl_in[i], hh, ww, 256, S, 128, 256, 0, 1, act);       // relu(cat(...)) = per-branch relu
            conv3(p + "c51.", lvl_in[i], hh, ww, 256, t51, 64, 64, 0, 1, act);
            conv3(p + "c52.", t51, hh, ww, 64, S, 64, 256, 128, 1, act);
            conv3(p + "c72.", t51, hh, ww, 64, t72, 64, 64, 0, 1, act);
            conv3(p + "c73.", t72, hh, ww, 64, S, 64, 256, 192, 1, act);
            if (i == 0) net.tap("face_ssh1", S, (size_t)nb * hh * ww * 256 * es);
            const std::string hp = "head" + std::to_string(i) + ".";
            net.gemm(linear_desc((long)nb * hh * ww, 256, 64, 0), hp + "w", nullptr, net.F(hp + "b"), S, nullptr, hd, act, 0);
            net.chk(k_face_head(ctx, hd, 64, nb, hh * ww, row0, P, loc + (size_t)s0 * P * 4, conf + (size_t)s0 * P * 2,
                                landms + (size_t)s0 * P * 10, st));
            row0 += hh * ww * 2;
        }
        if (net.err != AVCER_OK) return net.err;
    }
    return net.err;
}

extern "C" int avcer_load_face(avcer_ctx* ctx, const void* blob, size_t nbytes) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return load_blob(ctx, ctx->face, blob, nbytes);
}

extern "C" int avcer_face_num_priors(int h, int w) {
    if (h < 1 || w < 1) return 0;
    int fh = ((h - 1) / 2 + 1 - 1) / 2 + 1, fw = ((w - 1) / 2 + 1 - 1) / 2 + 1, p = 0;
    for (int li = 2; li <= 4; ++li) { fh = (fh - 1) / 2 + 1; fw = (fw - 1) / 2 + 1; p += 2 * fh * fw; }
    return p;
}

extern "C" int avcer_face_forward(avcer_ctx* ctx, const uint8_t* frames, int n, int h, int w, int rgb, int mode, float* loc,
                                  float* conf, float* landms, avcer_stream_t stream) {
    return face_forward_impl(ctx, frames, n, h, w, rgb ? 1 : 0, mode, loc, conf, landms, stream);
}

extern "C" int avcer_gather_windows(avcer_ctx* ctx, const float* feats, const int32_t* idx, int nwin, float* out,
                                    avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!feats || !idx || !out || nwin <= 0) return set_err(ctx, AVCER_EINVAL, "gather_windows: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_gather_windows(ctx, feats, idx, nwin, out, (hipStream_t)stream);
}

extern "C" int avcer_audio_chunks(avcer_ctx* ctx, const float* wav, const int32_t* starts, const int32_t* ends, int n,
                                  int window, int mode, float* out, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!wav || !starts || !ends || !out || n <= 0 || window <= 0 || mode < 0 || mode > 2)
        return set_err(ctx, AVCER_EINVAL, "audio_chunks: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_audio_chunks(ctx, wav, starts, ends, n, window, mode, out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ dynamic LSTM
// ref: architectures/video.py:169-185.  Input projections of all 10 steps are one GEMM per layer; the recurrent
// part is one [n,H]x[H,4H] GEMM + one cell kernel per step (h_0 = c_0 = 0, so step 0 needs no GEMM).
static int dynamic_forward_impl(avcer_ctx* ctx, const float* windows, int n, int mode, float* logits, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!ctx->dyn.loaded) return set_err(ctx, AVCER_ESTATE, "dynamic weights not loaded");
    if (!windows || !logits || n <= 0) return set_err(ctx, AVCER_EINVAL, "dynamic_forward: bad arguments");
    if (mode < AVCER_MODE_FP32 || mode > AVCER_MODE_F16X3) return set_err(ctx, AVCER_EINVAL, "dynamic_forward: mode %d", mode);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the recurrence keeps f32 state in every mode; AVCER_MODE_F16X3 runs its ten dependent GEMMs per layer on the
    // split-fp16 MFMA (a 128-row tile of the f32 MFMA costs 5x the cycles, and these launches are pure latency)
    const int x3 = mode == AVCER_MODE_F16X3;
    if (x3) TRY(ensure_all_x3(ctx, ctx->dyn, st));
    constexpr int T = 10, I = 512, H1 = 512, H2 = 256;
    const size_t total = ((size_t)n * T * 4 * H1 + (size_t)n * 4 * H1 + (size_t)n * T * H1 * 2 + (size_t)n * H1 +
                          (size_t)n * T * 4 * H2 + (size_t)n * H2 * 3) * 4 + 10 * 256;
    void* wsp = nullptr;
    TRY(ws_reserve(ctx, 1, total, &wsp));
    Arena ar(wsp, ctx->ws[1].cap);
    float* xp1 = (float*)ar.get((size_t)n * T * 4 * H1 * 4);
    float* hp = (float*)ar.get((size_t)n * 4 * H1 * 4);
    float* h1 = (float*)ar.get((size_t)n * T * H1 * 4);
    float* c1 = (float*)ar.get((size_t)n * H1 * 4);
    float* xp2 = (float*)ar.get((size_t)n * T * 4 * H2 * 4);
    float* h2 = (float*)ar.get((size_t)n * H2 * 4);
    float* c2 = (float*)ar.get((size_t)n * H2 * 4);
    // x3 mode: h of both layers once more as sp32 pairs, written by the cell kernel -- the operand of the next step's recurrent
    // contraction (and of layer 2's input projection), which then runs on the sp32 forms: at a window or a few hundred per call
    // that is the skinny one (9 us a step where the 128-row tile of the f32-operand form took 22)
    void* h1s = ar.get((size_t)n * T * H1 * 4);
    void* h2s = ar.get((size_t)n * H2 * 4);
    if (!h2s) return set_err(ctx, AVCER_ENOMEM, "dynamic workspace arithmetic");
    const int ak = x3 ? 2 : 0;
    Net net{ctx, ctx->dyn, 0, st, x3};
    net.gemm(linear_desc((long)n * T, I, 4 * H1, 0), "lstm1.wih.w", nullptr, net.F("lstm1.b"), windows, nullptr, xp1, 0, 0);
    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            avcer_conv_desc d = linear_desc(n, H1, 4 * H1, 0);
            d.x_stride_b = (int64_t)T * H1;  // rows of h_{t-1} inside the [n, T, H1] sequence buffer
            const void* hprev = x3 ? (const void*)((const char*)h1s + (size_t)(t - 1) * H1 * 4) : (const void*)(h1 + (size_t)(t - 1) * H1);
            net.gemm(d, "lstm1.whh.w", nullptr, nullptr, hprev, nullptr, hp, ak, 0);
        }
        net.chk(k_lstm_cell(ctx, xp1 + (size_t)t * 4 * H1, (int64_t)T * 4 * H1, hp, c1, h1 + (size_t)t * H1,
                            x3 ? (char*)h1s + (size_t)t * H1 * 4 : nullptr, (int64_t)T * H1, n, H1, t == 0, st));
    }
    net.gemm(linear_desc((long)n * T, H1, 4 * H2, 0), "lstm2.wih.w", nullptr, net.F("lstm2.b"), x3 ? h1s : (const void*)h1, nullptr,
             xp2, ak, 0);
    for (int t = 0; t < T; ++t) {
        if (t > 0) net.gemm(linear_desc(n, H2, 4 * H2, 0), "lstm2.whh.w", nullptr, nullptr, x3 ? h2s : (const void*)h2, nullptr, hp, ak, 0);
        net.chk(k_lstm_cell(ctx, xp2 + (size_t)t * 4 * H2, (int64_t)T * 4 * H2, hp, c2, h2, x3 ? h2s : nullptr, H2, n, H2, t == 0, st));
    }
    net.chk(k_small_linear(ctx, h2, net.F("fc.w"), net.F("fc.b"), logits, nullptr, n, H2, 7, 0, st));
    return net.err;
}

extern "C" int avcer_dynamic_forward(avcer_ctx* ctx, const float* windows, int n, float* logits, avcer_stream_t stream) {
    return dynamic_forward_impl(ctx, windows, n, AVCER_MODE_FP32, logits, stream);
}

extern "C" int avcer_dynamic_forward_mode(avcer_ctx* ctx, const float* windows, int n, int mode, float* logits,
                                          avcer_stream_t stream) {
    return dynamic_forward_impl(ctx, windows, n, mode, logits, stream);
}

// ------------------------------------------------------------------------------------------------ audio model
// ref: architectures/audio_8_cl.py:131-190; transformers 4.36.2 Wav2Vec2Model with feat_extract_norm="layer",
// do_stable_layer_norm=True (third party); architectures/attention_layers.py:221-267.
// Residual streams stay f32 in every mode; MFMA operand tensors (LN outputs, attention context, GELU'd FFN hidden,
// conv-extractor activations) are f32 / bf16 / sp32 pairs according to the mode.
extern "C" int avcer_audio_forward(avcer_ctx* ctx, const float* wav, int n, int t, int normalize, int mode,
                                   float* logits, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!ctx->aud.loaded) return set_err(ctx, AVCER_ESTATE, "audio weights not loaded");
    if (!wav || !logits || n <= 0) return set_err(ctx, AVCER_EINVAL, "audio_forward: bad arguments");
    if (mode < AVCER_MODE_FP32 || mode > AVCER_MODE_F16X3) return set_err(ctx, AVCER_EINVAL, "audio_forward: mode %d", mode);
    static const int ck[7] = {10, 3, 3, 3, 3, 2, 2}, cs[7] = {5, 2, 2, 2, 2, 2, 2};
    int len[8];
    len[0] = t;
    for (int i = 0; i < 7; ++i) {
        if (len[i] < ck[i]) return set_err(ctx, AVCER_EINVAL, "audio_forward: %d samples is too short", t);
        len[i + 1] = (len[i] - ck[i]) / cs[i] + 1;
    }
    const int S = len[7];
    if (S > 256) return set_err(ctx, AVCER_EINVAL, "audio_forward: %d tokens > 256 (window longer than ~5 s)", S);
    const int L1 = (S - 2 * 4 - 1) / 3 + 1;  // Conv1d k5 s3 dil2
    const int L2 = L1 / 5;                   // MaxPool1d(5)
    const int L3 = L2 - 2;                   // Conv1d k3
    if (S < 9 || L2 < 3) return set_err(ctx, AVCER_EINVAL, "audio_forward: %d tokens is too short for the head", S);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int bf = mode == AVCER_MODE_BF16;
    const int act = bf ? 1 : (mode == AVCER_MODE_F16X3 ? 2 : 0);  // storage of MFMA operand tensors: f32 / bf16 / sp32
    const int plain = bf ? 1 : 0;                                   // storage of GEMM outputs read by LN / attention
    if (bf) TRY(ensure_all_bf16(ctx, ctx->aud, st));
    if (mode == AVCER_MODE_F16X3) TRY(ensure_all_x3(ctx, ctx->aud, st));
    const size_t es = bf ? 2 : 4;
    const int C = 512, E = 1024, FF = 4096;
    const int NB = std::min(n, 128);
    const size_t rows = (size_t)NB * S;
    const size_t e0 = (size_t)NB * len[1] * C * es, e1 = (size_t)NB * len[2] * C * es;
    const size_t total = (size_t)NB * t * 4 + e0 + 2 * e1 + rows * E * 4 * 4 + rows * E * es * 2 + rows * 3 * E * es +
                         rows * FF * es + rows * E * es + (size_t)NB * (L1 + 2 * L2 + L3 + 1) * E * 4 + 34 * 256;
    void* wsp = nullptr;
    TRY(ws_reserve(ctx, 2, total, &wsp));
    Arena ar(wsp, ctx->ws[2].cap);
    float* wn = (float*)ar.get((size_t)NB * t * 4);
    void* EA = ar.get(e0);
    void* EB = ar.get(e1);
    void* TMP = ar.get(e1);
    float* Hf = (float*)ar.get(rows * E * 4);   // residual stream
    float* Of = (float*)ar.get(rows * E * 4);
    float* Xf = (float*)ar.get(rows * E * 4);
    float* Yf = (float*)ar.get(rows * E * 4);
    void* Xb = ar.get(rows * E * es);  // operand-typed copies (bf16 or sp32) of f32 tensors
    void* Yb = ar.get(rows * E * es);
    void* QKV = ar.get(rows * 3 * E * es);
    void* FFB = ar.get(rows * FF * es);
    void* ATT = ar.get(rows * E * es);
    float* c1o = (float*)ar.get((size_t)NB * L1 * E * 4);
    float* mp = (float*)ar.get((size_t)NB * L2 * E * 4);
    void* mps = ar.get((size_t)NB * L2 * E * 4);  // x3 mode: the pooled head activations as sp32 pairs (td4's operand)
    float* c2o = (float*)ar.get((size_t)NB * L3 * E * 4);
    float* pooled = (float*)ar.get((size_t)NB * E * 4);
    if (!pooled) return set_err(ctx, AVCER_ENOMEM, "audio workspace arithmetic");
    const int ncls = ctx->aud_classes;

    Net net{ctx, ctx->aud, bf, st, mode == AVCER_MODE_F16X3};
    // LN whose output is an MFMA operand: f32 in the f32 mode, else a bf16 / sp32 tensor
    // GELU: the library erff in the f32 mode; the short Abramowitz-Stegun erf (gemm_dev.h gelu_fast, same 5e-7 bound as
    // the exact form's own f32 rounding) where the contractions around it are bf16 or split-fp16
    const int gelu = act ? 3 : 2;
    auto ln_act = [&](const void* x, int x_kind, const std::string& p, void* y, long r, int c, int fn) {
        net.chk(k_layernorm(ctx, x, nullptr, net.F(p + ".g"), net.F(p + ".b"), act ? nullptr : y, act ? y : nullptr, r, c,
                            1e-5f, fn, x_kind, act, st));
    };
    for (int s0 = 0; s0 < n; s0 += NB) {
        const int nb = std::min(NB, n - s0);
        const long r = (long)nb * S;
        const float* x0 = wav + (size_t)s0 * t;
        if (normalize) {
            net.chk(k_wav_normalize(ctx, x0, wn, nb, t, st));
            x0 = wn;
        }
        // ---- conv feature extractor: 7 x (Conv1d -> LN(512) -> GELU)
        net.chk(k_conv0_ln_gelu(ctx, x0, net.F("fe0.w"), net.F("fe0.cb"), net.F("fe0.ln.g"), net.F("fe0.ln.b"), EA, nb, t,
                                len[1], act, st));
        net.tap("norm", x0, (size_t)nb * t * 4);
        net.tap("conv0", EA, (size_t)nb * len[1] * C * es);
        void* cur = EA;
        void* nxt = EB;
        for (int i = 1; i < 7; ++i) {
            const std::string p = "fe" + std::to_string(i);
            net.gemm(conv1d_desc(nb, len[i], C, ck[i], cs[i], 0, 1, len[i + 1], C, 0), p + ".w", nullptr, net.F(p + ".cb"),
                     cur, nullptr, TMP, act, plain);
            ln_act(TMP, plain, p + ".ln", nxt, (long)nb * len[i + 1], C, gelu);
            std::swap(cur, nxt);
            if (i == 1) nxt = EA;  // EA (largest) is free once layer 1 has consumed it
        }
        net.tap("extract", cur, (size_t)r * C * es);
        // ---- feature projection: LN(512) -> Linear 512 -> 1024
        ln_act(cur, act, "fp.ln", TMP, r, C, 0);
        net.gemm(linear_desc(r, C, E, 0), "fp.w", nullptr, net.F("fp.b"), TMP, nullptr, Hf, act, 0);
        net.tap("proj", Hf, (size_t)r * E * 4);
        // ---- positional conv embedding (k=128, groups=16, pad 64, drop last, GELU) added to the stream
        const void* pin = Hf;
        if (act == 1) {
            net.chk(k_f32_to_bf16(ctx, Hf, (bf16_t*)Xb, (size_t)r * E, st));
            pin = Xb;
        } else if (act == 2) {
            net.chk(k_split_weights(ctx, Hf, (bf16_t*)Xb, (size_t)r * E, st));  // same sp32 layout as split weights
            pin = Xb;
        }
        {  // 16 groups of 64 channels in ONE launch (grid.y = group)
            avcer_conv_desc d = conv1d_desc(nb, S, 64, 128, 1, 64, 1, S, 64, gelu);
            d.x_stride_b = (int64_t)S * E; d.x_stride_h = E; d.x_stride_w = E;
            d.y_ld = E; d.r_ld = E;
            d.res_after_act = 1;
            d.groups = 16;
            net.gemm(d, "pos.w", nullptr, net.F("pos.b"), pin, Hf, Of, act, 0);
        }
        float* h = Of;
        net.tap("posconv", h, (size_t)r * E * 4);
        // ---- 12 pre-LN encoder layers
        for (int l = 0; l < 12; ++l) {
            const std::string p = "enc" + std::to_string(l);
            ln_act(h, 0, p + ".ln1", TMP, r, E, 0);
            net.gemm(linear_desc(r, E, 3 * E, 0), p + ".qkv.w", nullptr, net.F(p + ".qkv.b"), TMP, nullptr, QKV, act, plain);
            net.chk(k_attention(ctx, QKV, ATT, nb, S, 16, 64, 0.125f, plain, act, st));
            net.gemm(linear_desc(r, E, E, 0), p + ".o.w", nullptr, net.F(p + ".o.b"), ATT, h, h, act, 0);
            ln_act(h, 0, p + ".ln2", TMP, r, E, 0);
            net.gemm(linear_desc(r, E, FF, gelu), p + ".ff1.w", nullptr, net.F(p + ".ff1.b"), TMP, nullptr, FFB, act, act);
            net.gemm(linear_desc(r, FF, E, 0), p + ".ff2.w", nullptr, net.F(p + ".ff2.b"), FFB, h, h, act, 0);
            net.tap(("layer" + std::to_string(l)).c_str(), h, (size_t)r * E * 4);
        }
        net.chk(k_layernorm(ctx, h, nullptr, net.F("enc.ln.g"), net.F("enc.ln.b"), Xf, nullptr, r, E, 1e-5f, 0, 0, 0, st));
        net.tap("w2v", Xf, (size_t)r * E * 4);
        // ---- two first-party TransformerLayers (32 x 32 and 16 x 64 heads)
        float* xin = Xf;
        for (int l = 1; l <= 2; ++l) {
            const std::string p = "tl" + std::to_string(l);
            const int heads = l == 1 ? 32 : 16, dh = E / heads;
            // x + PE is both the projections' operand (typed copy) and the residual (f32)
            net.chk(k_add_pe(ctx, xin, net.F("pe"), Yf, act ? Yb : nullptr, nb, S, E, act, st));
            net.gemm(linear_desc(r, E, 3 * E, 0), p + ".qkv.w", nullptr, nullptr, act ? Yb : (void*)Yf, nullptr, QKV, act, plain);
            net.chk(k_attention(ctx, QKV, ATT, nb, S, heads, dh, 1.0f / sqrtf((float)dh), plain, act, st));
            net.gemm(linear_desc(r, E, E, 0), p + ".o.w", nullptr, nullptr, ATT, Yf, Hf, act, 0);
            net.chk(k_layernorm(ctx, Hf, nullptr, net.F(p + ".ln1.g"), net.F(p + ".ln1.b"), Yf, act ? Yb : nullptr, r, E, 1e-5f,
                                0, 0, act, st));
            net.gemm(linear_desc(r, E, E, 1), p + ".ff1.w", nullptr, net.F(p + ".ff1.b"), act ? Yb : (void*)Yf, nullptr, FFB,
                     act, act);
            net.gemm(linear_desc(r, E, E, 0), p + ".ff2.w", nullptr, net.F(p + ".ff2.b"), FFB, Yf, Hf, act, 0);
            net.chk(k_layernorm(ctx, Hf, nullptr, net.F(p + ".ln2.g"), net.F(p + ".ln2.b"), Xf, act ? Xb : nullptr, r, E, 1e-5f,
                                0, 0, act, st));
            xin = Xf;
            net.tap(p.c_str(), Xf, (size_t)r * E * 4);
        }
        // ---- head: Conv1d k5 s3 dil2 + BN -> MaxPool(5) -> ReLU -> Conv1d k3 + BN -> mean -> ReLU -> Linear
        net.gemm(conv1d_desc(nb, S, E, 5, 3, 0, 2, L1, E, 0), "td0.w", net.F("td0.s"), net.F("td0.b"),
                 act ? Xb : (void*)Xf, nullptr, c1o, act, 0);
        net.chk(k_maxpool1d_relu(ctx, c1o, mp, net.x3 ? mps : nullptr, nb, L1, L2, E, 5, st));
        net.gemm(conv1d_desc(nb, L2, E, 3, 1, 0, 1, L3, E, 0), "td4.w", net.F("td4.s"), net.F("td4.b"), net.x3 ? mps : (const void*)mp,
                 nullptr, c2o, net.x3 ? 2 : 0, 0);
        net.tap("td0", c1o, (size_t)nb * L1 * E * 4);
        net.tap("mp", mp, (size_t)nb * L2 * E * 4);
        net.tap("td4", c2o, (size_t)nb * L3 * E * 4);
        net.chk(k_mean_time_relu(ctx, c2o, pooled, nb, L3, E, st));
        net.tap("pooled", pooled, (size_t)nb * E * 4);
        net.chk(k_small_linear(ctx, pooled, net.F("fd.w"), net.F("fd.b"), logits + (size_t)s0 * ncls, nullptr, nb, E, ncls, 0, st));
        if (net.err != AVCER_OK) return net.err;
    }
    return net.err;
}

// ------------------------------------------------------------------------------------------------ fusion
extern "C" int avcer_audio_frame_mean(avcer_ctx* ctx, const float* win_logits, const int32_t* frame_lo,
                                      const int32_t* frame_hi, int n_win, int c, int n_frames, float* out,
                                      int32_t* count, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!win_logits || !frame_lo || !frame_hi || !out || n_win < 0 || c <= 0 || n_frames <= 0)
        return set_err(ctx, AVCER_EINVAL, "audio_frame_mean: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_frame_mean(ctx, win_logits, frame_lo, frame_hi, n_win, c, n_frames, out, count, (hipStream_t)stream);
}

extern "C" int avcer_face_decode(avcer_ctx* ctx, const float* loc, const float* conf, const float* landms,
                                 const float* priors, int n_priors, int im_h, int im_w, float var0, float var1,
                                 float* dets, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!loc || !conf || !landms || !priors || !dets || n_priors <= 0 || im_h <= 0 || im_w <= 0)
        return set_err(ctx, AVCER_EINVAL, "face_decode: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_face_decode(ctx, loc, conf, landms, priors, n_priors, im_h, im_w, var0, var1, dets, (hipStream_t)stream);
}

extern "C" int avcer_face_nms(avcer_ctx* ctx, const float* dets, int n_frames, int n_priors, float conf_thresh, float nms_thresh,
                              int nms_top_k, int top_k, float threshold, float* out, int32_t* out_n, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!dets || !out || !out_n || n_frames <= 0 || n_priors <= 0 || nms_top_k <= 0 || top_k <= 0 || top_k > 1024)
        return set_err(ctx, AVCER_EINVAL, "face_nms: bad arguments (top_k <= 1024)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* wsp = nullptr;
    TRY(ws_reserve(ctx, 4, ((size_t)n_frames * nms_top_k + n_frames) * 4 + 256, &wsp));
    int32_t* order = (int32_t*)wsp;
    int32_t* count = order + (size_t)n_frames * nms_top_k;
    return k_face_nms(ctx, dets, n_frames, n_priors, conf_thresh, nms_thresh, nms_top_k, top_k, threshold, order, count, out, out_n,
                      (hipStream_t)stream);
}

extern "C" int avcer_crop_tiles(avcer_ctx* ctx, const uint8_t* frames, int n_frames, int h, int w, const int32_t* rects,
                                int n, int swap_rb, uint8_t* tiles, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!frames || !rects || !tiles || n_frames <= 0 || h <= 0 || w <= 0 || n <= 0)
        return set_err(ctx, AVCER_EINVAL, "crop_tiles: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_crop_tiles(ctx, frames, n_frames, h, w, rects, n, swap_rb ? 1 : 0, tiles, (hipStream_t)stream);
}

extern "C" int avcer_fuse(avcer_ctx* ctx, const float* stat, const float* dyn_logits, const float* aud_mean, int n,
                          int n_aud, int aud_c, const double* w1, const double* w2, int ce_weights_type, int ce_mask,
                          double* comp_prob, int32_t* comp_argmax, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!stat || !dyn_logits || !aud_mean || !comp_prob || !comp_argmax || n <= 0 || n_aud <= 0 || aud_c < 7)
        return set_err(ctx, AVCER_EINVAL, "fuse: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double w[21];
    if (w1)
        for (int m = 0; m < 3; ++m)
            for (int k = 0; k < 7; ++k) w[m * 7 + k] = w1[m * 7 + k] * (w2 ? w2[m] : 1.0);  // run.py:109-111
    return k_fuse(ctx, stat, dyn_logits, aud_mean, n, std::min(n_aud, n), aud_c, w1 ? w : nullptr, w1 != nullptr,
                  ce_weights_type, ce_mask, comp_prob, comp_argmax, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ kernel-level entry
extern "C" int avcer_conv_gemm(avcer_ctx* ctx, const avcer_conv_desc* d, int dtype, const void* x, const void* w,
                               const float* scale, const float* bias, const void* residual, void* y,
                               avcer_stream_t stream) {
    if (!ctx || !d) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_conv_gemm(ctx, *d, dtype, x, w, scale, bias, residual, y, (hipStream_t)stream);
}

// Frames per internal pass of the static CNN (default 1024; layers 3/4 need >= 512 frames to fill 256 CUs twice).
extern "C" int avcer_set_static_batch(avcer_ctx* ctx, int frames) {
    if (!ctx) return AVCER_EINVAL;
    if (frames < 1 || frames > 1024) return set_err(ctx, AVCER_EINVAL, "static batch %d outside [1,1024]", frames);
    ctx->static_batch = frames;
    return AVCER_OK;
}

extern "C" int avcer_profile_enable(avcer_ctx* ctx, int on) {
    if (!ctx) return AVCER_EINVAL;
    ctx->prof = on != 0;
    ctx->prof_used = 0;
    for (int f = 0; f < 8; ++f) { ctx->fam_launches[f] = 0; ctx->fam_flops[f] = 0.0; ctx->fam_bytes[f] = 0.0; }
    return AVCER_OK;
}

// The same events by kernel family (AVCER_FAM_*): summed HIP-event milliseconds, launches, algorithmic FLOPs and compulsory
// HBM bytes of the launches recorded since avcer_profile_enable / the last read; arrays of n_fam entries; synchronises.
extern "C" int avcer_profile_read_families(avcer_ctx* ctx, int n_fam, double* ms, int64_t* launches, double* flops, double* bytes) {
    if (!ctx || n_fam < 1 || n_fam > 8 || !ms || !launches || !flops || !bytes) return ctx ? set_err(ctx, AVCER_EINVAL, "profile_read_families: bad arguments") : AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int f = 0; f < n_fam; ++f) { ms[f] = 0.0; launches[f] = 0; flops[f] = 0.0; bytes[f] = 0.0; }
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->prof_ev[i + 1]));
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
        const int f = ctx->prof_fam[i / 2];
        if (f < n_fam) ms[f] += t;
    }
    for (int f = 0; f < n_fam && f < 8; ++f) {
        launches[f] = ctx->fam_launches[f]; flops[f] = ctx->fam_flops[f]; bytes[f] = ctx->fam_bytes[f];
        ctx->fam_launches[f] = 0; ctx->fam_flops[f] = 0.0; ctx->fam_bytes[f] = 0.0;
    }
    ctx->prof_used = 0;
    return AVCER_OK;
}

// Sum of the HIP-event durations of the conv_gemm launches recorded since avcer_profile_enable; synchronises.
extern "C" int avcer_profile_read(avcer_ctx* ctx, double* total_ms, int64_t* launches) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double ms = 0.0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->prof_ev[i + 1]));
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int64_t)(ctx->prof_used / 2);
    ctx->prof_used = 0;
    return AVCER_OK;
}

extern "C" int avcer_debug_tap(avcer_ctx* ctx, const char* name, void* dst_dev, size_t bytes) {
    if (!ctx || !name) return AVCER_EINVAL;
    ctx->tap_name = name;
    ctx->tap_dst = dst_dev;
    ctx->tap_bytes = bytes;
    ctx->tap_copied = -1;
    return AVCER_OK;
}

extern "C" int64_t avcer_debug_tap_copied(const avcer_ctx* ctx) { return ctx ? ctx->tap_copied : -1; }

extern "C" int avcer_conv_gemm_dual(avcer_ctx* ctx, const avcer_conv_desc* d, int dtype, const void* x, const void* x2,
                                    const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                    avcer_stream_t stream) {
    if (!ctx || !d || !x2) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_conv_gemm(ctx, *d, dtype, x, w, scale, bias, residual, y, (hipStream_t)stream, x2);
}

extern "C" int avcer_bneck_chain(avcer_ctx* ctx, int planes, int nb, int h, int w, const void* t1, const void* x, int ds_cin,
                                 int out_step, void* out, void* t1n, const void* w2, const float* b2, const void* w3, const float* b3,
                                 const void* w1n, const float* b1n, const void* w2_frags, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (nb <= 0 || h <= 0 || w <= 0) return set_err(ctx, AVCER_EINVAL, "bneck_chain: bad geometry");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_bneck(ctx, planes, nb, h, w, t1, x, ds_cin, out_step, out, t1n, w2, b2, w3, b3, w1n, b1n, (hipStream_t)stream, w2_frags);
}

extern "C" int avcer_stem_pool(avcer_ctx* ctx, const void* planes_hi_lo, size_t plane_bytes, const void* w, const float* scale,
                               const float* bias, void* y, int n, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (plane_bytes != (size_t)n * 230 * 230 * 4 * 2) return set_err(ctx, AVCER_EINVAL, "stem_pool: plane_bytes != n*230*230*4*2");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_stem_pool(ctx, planes_hi_lo, plane_bytes, w, scale, bias, y, n, (hipStream_t)stream);
}

extern "C" int avcer_stem_pool_u8(avcer_ctx* ctx, const uint8_t* frames, int n, int in_h, int in_w, const void* w, const float* scale,
                                  const float* shifts9, void* y, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_stem_pool_u8(ctx, frames, in_h, in_w, w, scale, shifts9, y, n, (hipStream_t)stream);
}

extern "C" int avcer_split_weights(avcer_ctx* ctx, const float* w, void* out, size_t numel, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!w || !out || numel % 32) return set_err(ctx, AVCER_EINVAL, "split_weights: numel must be a multiple of 32");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_split_weights(ctx, w, (bf16_t*)out, numel, (hipStream_t)stream);
}

extern "C" int avcer_split_weight_rows(avcer_ctx* ctx, const float* w, void* out, int n, int k, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!w || !out || n <= 0 || k <= 0) return set_err(ctx, AVCER_EINVAL, "split_weight_rows: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_split_weight_rows(ctx, w, (bf16_t*)out, n, k, (hipStream_t)stream);
}

extern "C" int avcer_attention(avcer_ctx* ctx, const void* qkv, void* out, int n, int s, int heads, int head_dim, float scale,
                               int in_kind, int out_kind, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!qkv || !out || n <= 0 || heads <= 0) return set_err(ctx, AVCER_EINVAL, "attention: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_attention(ctx, qkv, out, n, s, heads, head_dim, scale, in_kind, out_kind, (hipStream_t)stream);
}

extern "C" int avcer_weight_frags(avcer_ctx* ctx, const void* rows, void* out, int n, int k, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    if (!rows || !out || n <= 0 || k <= 0) return set_err(ctx, AVCER_EINVAL, "weight_frags: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return k_weight_frags(ctx, (const bf16_t*)rows, (bf16_t*)out, n, k, (hipStream_t)stream);
}

extern "C" int avcer_measure_ceilings(avcer_ctx* ctx, double* mfma_bf16_tflops, double* hbm_copy_tbs, avcer_stream_t stream) {
    if (!ctx) return AVCER_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return measure_ceilings(ctx, mfma_bf16_tflops, hbm_copy_tbs, (hipStream_t)stream);
}

extern "C" int avcer_gemm_stats(avcer_ctx* ctx, int64_t* launches, double* flops, int reset) {
    if (!ctx) return AVCER_EINVAL;
    if (launches) *launches = ctx->gemm_launches;
    if (flops) *flops = ctx->gemm_flops;
    if (reset) {
        ctx->gemm_launches = 0;
        ctx->gemm_flops = 0.0;
    }
    return AVCER_OK;
}
