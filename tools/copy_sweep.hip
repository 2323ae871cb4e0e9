// Streaming-copy sweep for avcer_measure_ceilings' HBM ceiling (tools only, not part of libavcer_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_sweep tools/copy_sweep.hip && /tmp/copy_sweep
// Variants: loads in flight per thread (U), plain vs non-temporal accesses, one-shot grid vs persistent grid-stride grid,
// block size.  Prints TB/s (bytes read + bytes written) for a 1 GiB -> 1 GiB copy, median of 7 timed launches each.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int U, bool NT>
__global__ void copy_once(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
    const size_t t = (size_t)gridDim.x * blockDim.x, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * t) : src[i + k * t];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (NT) __builtin_nontemporal_store(v[k], dst + i + k * t);
        else dst[i + k * t] = v[k];
    }
}

// contiguous per-block chunks: block b copies [b * U * blockDim, (b + 1) * U * blockDim)
template <int U, bool NT>
__global__ void copy_chunk(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    u4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + base + k * blockDim.x) : src[base + k * blockDim.x];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (NT) __builtin_nontemporal_store(v[k], dst + base + k * blockDim.x);
        else dst[base + k * blockDim.x] = v[k];
    }
}

template <int U, bool NT>
__global__ void copy_persistent(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
    const size_t t = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * t < n16; i += U * t) {
        u4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * t) : src[i + k * t];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + k * t);
            else dst[i + k * t] = v[k];
        }
    }
}

// The access shape of the fused bottleneck kernels' residual loads and output stores: a wave covers 16 rows of 1024 B, lane
// l = (row l & 15, 16-byte piece l >> 4) -- four lanes give 64 contiguous bytes of a row, the second access the other half of
// the 128-byte line.  GROUPS = how many consecutive 128-byte groups of its rows a wave walks (8 = a whole 1024-byte row).
template <int GROUPS>
__global__ void copy_fragment(const char* __restrict__ src, char* __restrict__ dst, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t row = wave * 16 + (lane & 15);
    if (row >= rows) return;
    const size_t off = row * 1024 + (size_t)(lane >> 4) * 16;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const u4 h = *reinterpret_cast<const u4*>(src + off + g * 128), l = *reinterpret_cast<const u4*>(src + off + g * 128 + 64);
        *reinterpret_cast<u4*>(dst + off + g * 128) = h;
        *reinterpret_cast<u4*>(dst + off + g * 128 + 64) = l;
    }
}
// the same bytes with a wave covering 8 rows x 128 B per access (lane l = row l >> 3, piece l & 7): whole lines per row
template <int GROUPS>
__global__ void copy_lines(const char* __restrict__ src, char* __restrict__ dst, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const size_t row = wave * 16 + half * 8 + (lane >> 3);
        if (row >= rows) return;
        const size_t off = row * 1024 + (size_t)(lane & 7) * 16;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) *reinterpret_cast<u4*>(dst + off + g * 128) = *reinterpret_cast<const u4*>(src + off + g * 128);
    }
}

template <typename F>
double bench(const char* name, F launch, size_t bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    launch();
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double tbs = 2.0 * bytes / (ms[3] * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %6.3f TB/s  (best %6.3f)\n", name, ms[3], tbs, 2.0 * bytes / (ms[0] * 1e-3) / 1e12);
    fflush(stdout);
    return tbs;
}

int main() {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    char *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    const u4* s = (const u4*)a;
    u4* d = (u4*)b;
#define ONCE(U, NT, BS) bench("once U=" #U " nt=" #NT " block=" #BS, [&] { copy_once<U, NT><<<(unsigned)(n16 / (U * BS)), BS>>>(s, d, n16); }, bytes)
#define CHUNK(U, NT, BS) bench("chunk U=" #U " nt=" #NT " block=" #BS, [&] { copy_chunk<U, NT><<<(unsigned)(n16 / (U * BS)), BS>>>(s, d, n16); }, bytes)
#define PERS(U, NT, BS, G) bench("persistent U=" #U " nt=" #NT " block=" #BS " grid=" #G, [&] { copy_persistent<U, NT><<<G, BS>>>(s, d, n16); }, bytes)
    ONCE(4, false, 256);   // what avcer_measure_ceilings ran in round 2
    ONCE(4, true, 256);
    ONCE(8, false, 256);
    ONCE(8, true, 256);
    ONCE(2, false, 256);
    ONCE(1, false, 256);
    ONCE(4, false, 512);
    ONCE(4, false, 1024);
    CHUNK(4, false, 256);
    CHUNK(4, true, 256);
    CHUNK(8, false, 256);
    CHUNK(8, true, 256);
    CHUNK(16, false, 256);
    CHUNK(4, false, 1024);
    PERS(4, false, 256, 2048);
    PERS(4, true, 256, 2048);
    PERS(8, false, 256, 2048);
    PERS(4, false, 256, 4096);
    PERS(4, false, 512, 1024);
    PERS(4, false, 1024, 512);
    PERS(2, false, 256, 8192);
    {
        const size_t rows = bytes / 1024;
        const unsigned grid = (unsigned)(rows / 16 / 4);  // 4 waves per block, 16 rows per wave
        bench("fragment shape 16 rows x 64 B, 8 groups", [&] { copy_fragment<8><<<grid, 256>>>(a, b, rows); }, bytes);
        bench("line shape 8 rows x 128 B, 8 groups", [&] { copy_lines<8><<<grid, 256>>>(a, b, rows); }, bytes);
    }
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    bench("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
    hipFree(a);
    hipFree(b);
    return 0;
}
