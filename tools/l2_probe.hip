// What does the L2 -> CU path deliver over the whole chip when every wave streams L2-RESIDENT data?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_probe tools/l2_probe.hip && /tmp/l2_probe
//
// Round 5 met the same ceiling from two sides: the skinny contraction (hundreds of single-wave blocks streaming L2-resident
// activations and weights) stops at 11-12 TB/s, and the memory side of the planes-64 bottleneck chain (HBM trunk + tap re-reads +
// weight re-streams) moves 11 TB/s into the CUs.  This probe measures the path by itself: every block walks the same buffer
// (far smaller than an XCD's 4 MiB L2, far larger than a CU's 32 KiB L1) with 16-byte loads, eight in flight per lane, at
// 1 ... 8 waves per CU-slot, and -- for comparison -- a buffer that only HBM holds.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) stream_read(const uint4* __restrict__ buf, size_t n_vec, int iters, unsigned* sink) {
    const size_t stride = (size_t)blockDim.x * 8;
    size_t base = ((size_t)blockIdx.x * 977 * stride + threadIdx.x) % n_vec;  // blocks start at different places of the same buffer
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t i = base + (size_t)j * blockDim.x;
            if (i >= n_vec) i -= n_vec;
            v[j] = buf[i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        base += stride;
        if (base >= n_vec) base -= n_vec;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t big = (size_t)1 << 30;
    uint4* buf;
    unsigned* sink;
    if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, big);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs\n", cus);
    struct Cfg { size_t bytes; const char* what; };
    const Cfg bufs[] = {{(size_t)256 << 10, "256 KiB (L2-resident in every XCD)"}, {(size_t)1 << 20, "1 MiB (L2-resident)"},
                        {(size_t)2 << 20, "2 MiB (L2-resident)"}, {(size_t)64 << 20, "64 MiB (memory-side cache)"}, {big, "1 GiB (HBM)"}};
    for (const Cfg& c : bufs)
        for (int waves_per_cu : {1, 2, 4, 8, 16}) {
            // blocks of 64 * w threads so that `waves_per_cu` waves share a CU: one block per CU for w <= 4, then 2 / 4 blocks of 256
            const int threads = waves_per_cu <= 4 ? 64 * waves_per_cu : 256;
            const int blocks = cus * (waves_per_cu <= 4 ? 1 : waves_per_cu / 4);
            const size_t n_vec = c.bytes / 16;
            const int iters = 4000;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            stream_read<<<blocks, threads>>>(buf, n_vec, 200, sink);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            stream_read<<<blocks, threads>>>(buf, n_vec, iters, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)blocks * threads * 8 * 16 * iters;
            printf("%-38s %2d waves per CU (%4d blocks x %3d threads): %7.2f TB/s into the CUs (%6.1f GB/s per CU)\n", c.what, waves_per_cu, blocks,
                   threads, bytes / ms / 1e9, bytes / ms / 1e6 / cus);
            fflush(stdout);
        }
    return 0;
}
