// The split operand type of the parity-grade fast mode ("x3": AVCER_MODE_F16X3), shared by every translation unit.
//
// A value x is carried as a pair (hi, lo) of 16-bit floats, x = hi + lo, and a product as ah.wh + ah.wl + al.wh on the
// 16-bit MFMA with f32 accumulation.  Rounds 1-3 split into bf16 (8 + 8 significand bits: 2^-17 per operand, 4.5e-6 per
// contraction, which put sharp softmax heads at the 1e-4 probability gate).  Round 4 splits into IEEE fp16 (11 + 11 bits):
// v_mfma_f32_16x16x32_f16 issues at the bf16 form's rate on gfx950, the storage is the same 4 bytes per element, and the
// representation error falls to 2^-22 per operand -- 7e-8 per contraction (tools/x3_error_probe.py), f32-grade.
//
// What fp16 costs is RANGE, not rate:
//   * activations are stored unscaled: |x| must stay below 65504 (larger values become +-inf, and inf - inf = NaN in the lo
//     half: the overflow reaches the output as NaN, never as a wrong finite number).  Below 2^-3 the lo half is subnormal
//     in fp16 -- kept by the conversion and by the MFMA (tools/f16_probe.hip on the GPU: subnormal A / B operands are not
//     flushed) -- so a small element carries an ABSOLUTE error of at most 2^-25, f32-grade against an O(1) tensor;
//   * weights are O(1/sqrt(K)) and would sit in that subnormal band: every weight matrix is multiplied by ONE power of
//     two that puts its largest magnitude into [2^14, 2^15) before the split (split_weight_rows_kernel), and the inverse
//     -- also a power of two, so nothing rounds differently -- travels with the weights in a trailer behind the split
//     data (AVCER_SPLIT_TRAILER bytes: float [0] = the multiplier the consumer applies to its accumulators).
//
// AVCER_SPLIT_BF16 (lab builds only, tools/build_lab.sh) restores the round-3 bf16 split for A/B measurements.
#pragma once

#include <cstdint>

constexpr int AVCER_SPLIT_TRAILER_BYTES = 256;  // == AVCER_SPLIT_TRAILER of include/avcer_hip.h

#if defined(AVCER_SPLIT_BF16)
typedef __bf16 spe_t;
#else
typedef _Float16 spe_t;
#endif
typedef __attribute__((ext_vector_type(8))) spe_t spx8_t;
typedef __attribute__((ext_vector_type(4))) float sp_f32x4_t;

// A value about to be split must be ONE f32 number for both halves.  HIP contracts floating-point expressions by default,
// and on gfx950 hipcc folds a producing multiply INTO the conversion (v_fma_mixlo_f16: f16(a * b) with a single rounding) --
// separately for each use.  Round 4 caught it in the attention kernel's output: hi was stored as f16(f32(o * inv)) while lo
// was computed against f16(o * inv); on the rare values whose f32 rounding lands on an fp16 tie the two disagree by one hi
// ulp, and hi + lo is then wrong by 2^-11 relative (tools/x3_audio_stage_error.py: 6e-6 rms in the attention output, three
// times the f32 mode's error at the logits).  Passing the value through an empty asm makes it an opaque register.
__device__ __forceinline__ float sp_value(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// f32 -> the 16 bits of the split element type (round to nearest even) and back
__device__ __forceinline__ uint16_t f2sp(float f) { return __builtin_bit_cast(uint16_t, (spe_t)f); }
__device__ __forceinline__ float sp2f(uint16_t b) { return (float)__builtin_bit_cast(spe_t, b); }

__device__ __forceinline__ sp_f32x4_t mfma_sp(const spx8_t a, const spx8_t b, const sp_f32x4_t c) {
#if defined(AVCER_SPLIT_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// ---- the range contract's run-time signal (round 5; include/avcer_hip.h avcer_x3_overflow_count)
// An activation of magnitude >= 65520 rounds to +-inf in its fp16 hi half and reaches the output as NaN -- which is also what
// the reference legitimately returns for an empty audio window, so NaN alone does not tell a caller that the contract broke.
// Every site that splits an f32 ACTIVATION takes the maximum magnitude of the values it splits (one v_max3_f32 per two
// values; NaN operands do not move it: the max instructions return the other operand), compares it with the threshold and
// ORs the wave's ballot into a wave-uniform mask (scalar registers: a per-thread running maximum cost the fused kernels 20
// vector registers and a resident block).  Once, at the end of the kernel, a wave whose mask is not empty adds its lane
// count to the context's device counter.  Only FINITE values count (an infinite input was counted where it became
// infinite), so a NaN audio window leaves the counter at 0.
constexpr float AVCER_SP_OVERFLOW = 65520.f;  // smallest magnitude that rounds to +-inf in fp16 (round to nearest even)
typedef unsigned long long sp_flags_t;
__device__ __forceinline__ float sp_max2(float amax, float a, float b) {
    return __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(a)), __builtin_fabsf(b));
}
__device__ __forceinline__ bool sp_out_of_range(float amax) { return amax >= AVCER_SP_OVERFLOW && amax < __builtin_inff(); }
// amax = the largest magnitude of the values this lane has just split
__device__ __forceinline__ void sp_flag(sp_flags_t& flags, float amax) {
#if defined(__HIP_DEVICE_COMPILE__)
    flags |= __ballot(sp_out_of_range(amax));
#endif
}
__device__ __forceinline__ void sp_commit(unsigned* ovf, sp_flags_t flags) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ovf != nullptr && flags != 0) {  // wave-uniform; the lane id is recomputed here (mbcnt) instead of being kept alive
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (lane == (unsigned)__builtin_ctzll(__ballot(1))) atomicAdd(ovf, (unsigned)__builtin_popcountll(flags));
    }
#endif
}
// the HBM-bound element-wise kernels test and count on the spot (a rare divergent branch, no state)
__device__ __forceinline__ void sp_count_now(unsigned* ovf, float amax) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ovf != nullptr && sp_out_of_range(amax)) atomicAdd(ovf, 1u);
#endif
}

// the accumulator multiplier stored behind a split weight matrix of `bytes` bytes (see above)
__device__ __forceinline__ float split_wmul(const char* w, size_t bytes) { return *reinterpret_cast<const float*>(w + bytes); }
