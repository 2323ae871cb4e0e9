// Access-shape probe for the trunk streams of the fused bottleneck kernels (tools only, not part of libavcer_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_shapes tools/copy_shapes.hip && /tmp/copy_shapes
//
// bneck_kernel reads the residual and writes the block output as MFMA fragments: a wave instruction covers 16 rows
// (positions) x 64 bytes at a 1024-byte row stride (sp32 rows of 256 channels; the hi half of a 32-channel group, the lo half
// 64 bytes further on by a second instruction), two groups of residual in flight per wave, 2-3 blocks of 4 waves per CU.
// tools/copy_sweep.hip timed that shape with SIXTEEN accesses in flight per thread (5.26 TB/s against 6.19 for one 16-byte
// access per thread) -- which confounds shape and depth.  Here both shapes run with the kernel's own depth and occupancy:
//   frag     16 rows x 64 B per instruction (the sp32 row layout the library uses)
//   blocked  1 KiB contiguous per instruction: the same bytes of a tile-blocked tensor [M/16][C/32][hi, lo][lane][16 B]
// LDSK = KiB of LDS a block declares (caps the blocks per CU like the kernel's tile buffers), EXTRA = also write a 256-byte
// row per position per 8 groups (the T1' stream).  Prints TB/s of bytes read + written.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u4;

// one wave = 32 positions (two 16-position tiles) x 8 groups of 32 channels, ring of two groups in flight
template <bool BLOCKED, int LDSK, bool EXTRA>
__global__ void __launch_bounds__(256) stream_kernel(const char* __restrict__ src, char* __restrict__ dst, char* __restrict__ t1n,
                                                     long tiles16) {
    __shared__ char pad[LDSK ? LDSK * 1024 : 16];
    if (LDSK && src == nullptr) pad[threadIdx.x] = 1;  // keeps the allocation
    const int lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long tile0 = wave * 2;
    if (tile0 + 1 >= tiles16) return;
    long off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
        off[t] = BLOCKED ? (tile0 + t) * 16384 + lane * 16 : ((tile0 + t) * 16 + l15) * 1024 + g * 16;
    constexpr int GS = BLOCKED ? 2048 : 128, HL = BLOCKED ? 1024 : 64;  // byte step per group / from hi to lo
    u4 h[2][2], l[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            h[r][t] = *reinterpret_cast<const u4*>(src + off[t] + r * GS);
            l[r][t] = *reinterpret_cast<const u4*>(src + off[t] + r * GS + HL);
        }
    u4 accx = {0, 0, 0, 0};
#pragma unroll
    for (int G = 0; G < 8; ++G) {
        const int r = G & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u4 a = h[r][t], b = l[r][t];
            accx ^= a ^ b;
            a.x += 1;  // not a pure copy: the store data depend on the load
            *reinterpret_cast<u4*>(dst + off[t] + G * GS) = a;
            *reinterpret_cast<u4*>(dst + off[t] + G * GS + HL) = b;
            if (G + 2 < 8) {
                h[r][t] = *reinterpret_cast<const u4*>(src + off[t] + (G + 2) * GS);
                l[r][t] = *reinterpret_cast<const u4*>(src + off[t] + (G + 2) * GS + HL);
            }
        }
        asm volatile("" ::: "memory");
    }
    if (EXTRA) {  // T1': 64 channels per position = two groups of the same shape (rows of 256 bytes)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                char* yp = t1n + ((tile0 + t) * 16 + l15) * 256 + q * 128 + g * 16;
                *reinterpret_cast<u4*>(yp) = accx;
                *reinterpret_cast<u4*>(yp + 64) = accx;
            }
    }
}

template <typename F>
void bench(const std::string& name, F launch, double bytes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    launch();
    std::vector<float> ms;
    for (int r = 0; r < 9; ++r) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float t;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-58s %8.3f ms  %6.3f TB/s  (best %6.3f)\n", name.c_str(), ms[4], bytes / (ms[4] * 1e-3) / 1e12, bytes / (ms[0] * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const long rows = 3L << 20;  // 3 Mi positions x 1 KiB = 3 GiB per tensor: one 1024-frame stage-1 trunk tensor
    const size_t bytes = (size_t)rows * 1024;
    char *a, *b, *c;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, (size_t)rows * 256) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    const long tiles = rows / 16;
    const unsigned grid = (unsigned)(tiles / 2 / 4);
#define RUN(BL, LDSK, EX)                                                                                           \
    bench((BL ? "blocked (1 KiB / instr)" : "frag (16 rows x 64 B / instr)") + std::string("  lds " #LDSK " KiB  extra " #EX), \
          [&] { stream_kernel<BL, LDSK, EX><<<grid, 256>>>(a, b, c, tiles); }, 2.0 * bytes + (EX ? rows * 256.0 : 0.0))
    for (int rep = 0; rep < 2; ++rep) {
        RUN(false, 1, false);
        RUN(true, 1, false);
        RUN(false, 50, false);
        RUN(true, 50, false);
        RUN(false, 76, false);
        RUN(true, 76, false);
        RUN(false, 50, true);
        RUN(true, 50, true);
    }
    (void)hipFree(a);
    (void)hipFree(b);
    (void)hipFree(c);
    return 0;
}
