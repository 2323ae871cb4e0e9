// Device-side helpers shared by the MFMA kernels of libavcer_hip.so (gemm.hip, bneck.hip): LDS-DMA, the searched
// LDS swizzle, bf16 / sp32 conversions.  gfx950 only.
#pragma once

#include "common.h"
#include "split_dev.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

constexpr int ROWB = 128;  // bytes per tile row per K-step

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
constexpr unsigned OOB = 0xFFFFFF00u;  // voffset that always fails the buffer bounds check -> load returns 0

// 16 bytes per lane, global -> LDS (wave-uniform LDS base + lane * 16); zeros when voff fails the bounds check
template <typename Rsrc>
__device__ __forceinline__ void dma16(Rsrc rs, char* lds_wave_base, unsigned voff, unsigned soff = 0u) {
#if defined(__HIP_DEVICE_COMPILE__)
    // soff: wave-uniform byte offset (SGPR operand of the instruction), added to the per-lane voff by the hardware
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
#endif
}

// XOR swizzle of the 16-byte chunks of a 128-byte tile row.  The key f((row>>1)&7) with f = (0,1,4,5,6,7,2,3) was
// found by exhaustive search: it makes every ds_read_b128 fragment pattern used below (bf16 16x16x32, f32 32x32x2 and
// the two-chunk f32 row read of the split-fp16 mode) conflict-free under the MI355X 16-lane-group banking.
__device__ __forceinline__ int swz_key(int row) { return (int)((0x32765410u >> (((row >> 1) & 7) * 4)) & 7u); }
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ swz_key(row)) << 4); }

// (contraction off in the activations and the epilogues that call them: whether their last product fuses with the residual add
// behind it must not depend on how a kernel's epilogue happens to be shaped -- the forms of one contraction are bit-identical)
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// The same function with erf from Abramowitz & Stegun 7.1.26 (1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p |z|)):
// 14 vector instructions, no branch, against ~45 for the library erff (two divergent branches, a two-step exp).  Against
// float64, |gelu_fast - gelu| <= 4.7e-7 over [-8, 8] -- the same bound the exact form has from rounding its own f32 result
// (4.5e-7) --, with up to 5e-7 absolute on erf itself near 0.  Used where the arithmetic around it is the split-fp16 or
// bf16 one (4.5e-6 relative per contraction); the f32 mode keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
#pragma clang fp contract(off)
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

// ReLU that keeps a NaN a NaN like torch (any sign, any payload: the comparison is false for it), in TWO vector instructions
// (v_cmp_lt + v_cndmask) -- the (v > 0 ? v : (v != v ? v : 0)) form costs four, and the fused bottleneck kernels run this
// on every element they store.  -0.0 stays -0.0, which no consumer can tell from +0.0.
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// sp32 storage: per aligned group of 32 channels, 32 hi bf16 then 32 lo bf16 (value = hi + lo; 4 bytes per element)
__device__ __forceinline__ long sp32_byte(long e) { return ((e & ~31L) << 2) + ((e & 31L) << 1); }

}  // namespace
