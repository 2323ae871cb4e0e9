// Does gfx950 record IEEE exception status of VALU conversions in the wave's sticky TRAPSTS.EXCP bits with traps disabled?
// (tools only)   hipcc --offload-arch=gfx950 -O3 -o /tmp/excp_probe tools/excp_probe.hip && /tmp/excp_probe
//
// If it does, "a finite f32 became +-inf in an fp16 conversion" -- the one way the x3 mode's range contract (|x| < 65504,
// csrc/split_dev.h) can be broken -- costs NOTHING to detect inside the kernels: one s_getreg_b32 at the end of a wave
// instead of a compare per split value.  Each block runs one case and stores hwreg(HW_REG_TRAPSTS, 0, 9) before / after:
// bit 0 invalid, 1 input denormal, 2 divide by zero, 3 OVERFLOW, 4 underflow, 5 inexact, 6 integer divide by zero.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((ext_vector_type(2))) _Float16 h2;

__global__ void probe(const float* in, unsigned* out, float* sink) {
    const unsigned before = __builtin_amdgcn_s_getreg((8 << 11) | 3);
    const float x = in[blockIdx.x * 64 + threadIdx.x];
    const float y = in[blockIdx.x * 64 + ((threadIdx.x + 1) & 63)];
    float r = 0.f;
    switch (blockIdx.x) {
        case 0: r = (float)(_Float16)x; break;                                   // in-range conversion (inexact only)
        case 1: r = (float)(_Float16)x; break;                                   // 70000 -> +inf
        case 2: { h2 p; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y)); r = (float)p[0] + (float)p[1]; } break;  // packed form, -1e6
        case 3: r = x * y; break;                                                // f32 overflow
        case 4: r = x - y; break;                                                // inf - inf: invalid
        case 5: r = (float)(_Float16)x; break;                                   // an inf input: exact, no overflow
        case 6: r = (float)(_Float16)x; break;                                   // a NaN input
        case 7: {                                                                // the library's split of an in-range value
            const _Float16 h = (_Float16)x;
            r = (float)h + (float)(_Float16)(x - (float)h);
        } break;
        case 8: {                                                                // the library's split of an overflowing value
            const _Float16 h = (_Float16)x;
            r = (float)h + (float)(_Float16)(x - (float)h);
        } break;
        default: break;
    }
    sink[blockIdx.x * 64 + threadIdx.x] = r;
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned after = __builtin_amdgcn_s_getreg((8 << 11) | 3);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = before;
        out[2 * blockIdx.x + 1] = after;
    }
}

int main() {
    const int NB = 9;
    float h[NB * 64];
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    const float v[NB] = {1.2345678f, 70000.f, -1e6f, 3e38f, inf, inf, nan, 123.456789f, 65520.f};
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < 64; ++i) h[b * 64 + i] = v[b];
    float *din, *dsink;
    unsigned* dout;
    if (hipMalloc(&din, sizeof(h)) != hipSuccess || hipMalloc(&dsink, sizeof(h)) != hipSuccess || hipMalloc(&dout, NB * 8) != hipSuccess) return 1;
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<NB, 64>>>(din, dout, dsink);
    unsigned o[NB * 2];
    float s[NB * 64];
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    (void)hipMemcpy(s, dsink, sizeof(s), hipMemcpyDeviceToHost);
    const char* name[NB] = {"cvt f16 in range", "cvt f16 of 70000", "packed cvt of -1e6", "f32 3e38 * 3e38", "inf - inf", "cvt f16 of inf",
                            "cvt f16 of NaN", "split 123.456789", "split 65520"};
    for (int b = 0; b < NB; ++b)
        printf("%-20s result %-12g  EXCP before 0x%03x  after 0x%03x  overflow bit %u\n", name[b], s[b * 64], o[2 * b], o[2 * b + 1],
               (o[2 * b + 1] >> 3) & 1);
    return 0;
}
