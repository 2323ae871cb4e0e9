// f16 MFMA probe (tools only, not part of libavcer_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/f16_probe tools/f16_probe.hip && gpurun_out/f16_probe
// Answers, before the split type of the fast mode moves from bf16 to fp16:
//   1. does v_mfma_f32_16x16x32_f16 keep SUBNORMAL f16 A / B operands (lo halves of small values are subnormal)?
//   2. does the f32 -> f16 conversion of the epilogues produce subnormals (round to nearest even) instead of flushing?
//   3. the register-only issue rate of the f16 form next to the bf16 form on this GPU, and of the f32 mode's v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// D[16x16] = A[16x32] B[32x16]: lane l holds A row (l & 15), k = 8 (l >> 4) .. +7, and B column (l & 15), the same k
__global__ void mfma_once(const _Float16* a, const _Float16* b, float* d) {
    const int lane = threadIdx.x;
    f16x8_t af, bf;
    for (int j = 0; j < 8; ++j) {
        af[j] = a[(lane & 15) * 32 + 8 * (lane >> 4) + j];
        bf[j] = b[(lane & 15) * 32 + 8 * (lane >> 4) + j];
    }
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];  // d[i][j] = sum_k a[i][k] b[j][k]
}

__global__ void cvt_probe(const float* x, _Float16* h, _Float16* l, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const _Float16 hi = (_Float16)x[i];
    h[i] = hi;
    l[i] = (_Float16)(x[i] - (float)hi);
}

template <int F16>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters) {
    f32x4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f16x8_t ah, bh;
    bf16x8_t ab, bb;
    for (int j = 0; j < 8; ++j) {
        ah[j] = (_Float16)(0.001f * (threadIdx.x + j));
        bh[j] = (_Float16)(0.002f * (threadIdx.x + 2 * j));
        ab[j] = (__bf16)(0.001f * (threadIdx.x + j));
        bb[j] = (__bf16)(0.002f * (threadIdx.x + 2 * j));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (F16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// the f32 mode's instruction: v_mfma_f32_32x32x2_f32, 4 independent accumulators per wave (the kernel's 2 x 2 tiles)
__global__ void __launch_bounds__(256) rate_f32_kernel(float* out, int iters) {
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x + 3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) out[0] = s;
}

static float h2f(_Float16 h) { return (float)h; }

int main() {
    // ---- 1. subnormal operands
    std::vector<_Float16> a(16 * 32, (_Float16)0.f), b(16 * 32, (_Float16)0.f);
    const float sub = ldexpf(1.f, -20);  // f16 subnormal (min normal 2^-14, subnormal step 2^-24)
    // row 0 of A: one subnormal; row 1: normal 2^-10; B column 0: 1024; column 1: subnormal 2^-18 against A row 2 = 4096
    a[0 * 32 + 3] = (_Float16)sub;
    a[1 * 32 + 5] = (_Float16)ldexpf(1.f, -10);
    a[2 * 32 + 7] = (_Float16)4096.f;
    b[0 * 32 + 3] = (_Float16)1024.f;
    b[0 * 32 + 5] = (_Float16)1024.f;
    b[1 * 32 + 7] = (_Float16)ldexpf(3.f, -19);
    _Float16 *da, *db;
    float* dd;
    hipMalloc(&da, a.size() * 2);
    hipMalloc(&db, b.size() * 2);
    hipMalloc(&dd, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    mfma_once<<<1, 64>>>(da, db, dd);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 1024      : got %.9g  want %.9g  %s\n", d[0 * 16 + 0], sub * 1024.f, d[0] == sub * 1024.f ? "KEPT" : "FLUSHED");
    printf("normal    A (2^-10) x 1024      : got %.9g  want %.9g\n", d[1 * 16 + 0], 1.0f);
    printf("4096 x subnormal B (3 * 2^-19)  : got %.9g  want %.9g  %s\n", d[2 * 16 + 1], 4096.f * ldexpf(3.f, -19),
           d[2 * 16 + 1] == 4096.f * ldexpf(3.f, -19) ? "KEPT" : "FLUSHED");
    // ---- 2. conversions
    const float xs[8] = {1e-3f, 3.3e-5f, 1.7e-6f, 5e-8f, 2.9e-8f, 0.1f, 70000.f, -1e-5f};
    float* dx;
    _Float16 *dh, *dl;
    hipMalloc(&dx, 32);
    hipMalloc(&dh, 16);
    hipMalloc(&dl, 16);
    hipMemcpy(dx, xs, 32, hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(dx, dh, dl, 8);
    _Float16 hh[8], hl[8];
    hipMemcpy(hh, dh, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hl, dl, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) {
        const _Float16 rh = (_Float16)xs[i];
        const _Float16 rl = (_Float16)(xs[i] - (float)rh);
        printf("cvt %-12g hi %.9g lo %.9g | host hi %.9g lo %.9g | hi+lo-x %.3g %s\n", xs[i], h2f(hh[i]), h2f(hl[i]), h2f(rh), h2f(rl),
               (double)h2f(hh[i]) + (double)h2f(hl[i]) - (double)xs[i],
               (memcmp(&rh, &hh[i], 2) == 0 && memcmp(&rl, &hl[i], 2) == 0) ? "same" : "DIFFERENT");
    }
    // ---- 3. rate
    float* dout;
    hipMalloc(&dout, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    for (int f16 = 0; f16 < 2; ++f16) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (f16) rate_kernel<1><<<blocks, 256>>>(dout, iters);
            else rate_kernel<0><<<blocks, 256>>>(dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 16 * 16 * 32 * 8.0 * iters * 4.0 * blocks;
            printf("%s 16x16x32 register-only: %.1f TFLOP/s (%.2f ms)\n", f16 ? "f16 " : "bf16", flop / ms * 1e-9, ms);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        const int it32 = 10000, waves_per_simd = 2;
        hipEventRecord(e0);
        rate_f32_kernel<<<256 * waves_per_simd, 256>>>(dout, it32);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * 2 * 4.0 * it32 * 4.0 * 256 * waves_per_simd;
        printf("f32  32x32x2 register-only (2 waves per SIMD, 4 accumulators): %.1f TFLOP/s (%.2f ms)\n", flop / ms * 1e-9, ms);
    }
    return 0;
}
