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
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    bench("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
    hipFree(a);
    hipFree(b);
    return 0;
}
