// How promptly does the hardware workgroup dispatcher refill a freed block slot of a CU?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_probe tools/dispatch_probe.hip && /tmp/dispatch_probe
//
// The spatial-tile form of the stage-1 bottleneck chain (bneck_kernel<..., T11>) shortened a block's residency by a quarter
// and did not get faster: a census of its blocks (HW_ID + s_memrealtime at entry and exit, tools/clock_lab.py) found a freed
// slot empty for 13-16 us on average where the older, slower form's slots are refilled within 3 us.  This probe separates the
// dispatcher from the kernel: blocks of the same footprint (256 threads, 46 KiB of LDS, ~168 VGPRs: three per CU) that only
// SLEEP -- for a fixed time, or for a time that varies from block to block like the real kernel's (p10 / p90 = 1 : 1.8) --
// and, optionally, hold memory traffic in flight while they do.  Per configuration: blocks resident per CU (time-weighted)
// and the delay between a block's exit and the next block's entry on the same CU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>

constexpr int LDS_BYTES = 46592 - 3072;  // + the hog spill area hipcc adds = the kernel's 46 592 bytes

__global__ void __launch_bounds__(256, 3) sleeper(unsigned long long* stamps, int base_ticks, int jitter_pct, const float* src, float* dst,
                                                  int traffic) {
    __shared__ char lds[LDS_BYTES];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // hold ~160 VGPRs alive so that the register file admits three blocks per CU, like the real kernel
    float hog[150];
#pragma unroll
    for (int i = 0; i < 150; ++i) hog[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = (char)threadIdx.x;
    unsigned h = blockIdx.x * 2654435761u;
    h ^= h >> 15;
    const int span = base_ticks * jitter_pct / 100;  // uniform in [base - span, base + span]
    const unsigned long long want = (unsigned long long)(base_ticks - span + (span ? (int)(h % (unsigned)(2 * span + 1)) : 0));
    float acc = 0.f;
    size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) {
        if (traffic) {  // a stream of 16-byte loads / stores per lane while waiting (the streaming phase's kind of traffic)
            const float4 v = *reinterpret_cast<const float4*>(src + off);
            *reinterpret_cast<float4*>(dst + off) = v;
            acc += v.x;
            off = (off + 256 * 4 * 4099) & ((1u << 28) - 1);
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
#pragma unroll
        for (int i = 0; i < 150; ++i) asm volatile("" : "+v"(hog[i]));
    }
    float s = acc;
#pragma unroll
    for (int i = 0; i < 150; ++i) s += hog[i];
    if (s == 123.456f) dst[0] = s + lds[(threadIdx.x * 7) & 255];
    if (threadIdx.x == 0) {
        stamps[3 * blockIdx.x] = t0;
        stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                     ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

int main() {
    const int blocks = 25600;
    unsigned long long* d;
    float *src, *dst;
    if (hipMalloc(&d, blocks * 24) != hipSuccess || hipMalloc(&src, 1u << 30) != hipSuccess || hipMalloc(&dst, 1u << 30) != hipSuccess) return 1;
    (void)hipMemset(src, 1, 1u << 30);
    std::vector<unsigned long long> h(blocks * 3);
    struct Cfg { int ticks, jitter, traffic; const char* name; };
    const Cfg cfgs[] = {{4500, 0, 0, "45 us, fixed, sleeping"},      {4500, 30, 0, "45 us +-30 %, sleeping"},
                        {4500, 0, 1, "45 us, fixed, streaming"},     {4500, 30, 1, "45 us +-30 %, streaming"},
                        {6000, 8, 0, "60 us +-8 %, sleeping"},       {6000, 8, 1, "60 us +-8 %, streaming"},
                        {1500, 30, 0, "15 us +-30 %, sleeping"}};
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            sleeper<<<blocks, 256>>>(d, c.ticks, c.jitter, src, dst, c.traffic);
            (void)hipDeviceSynchronize();
        }
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        sleeper<<<blocks, 256>>>(d, c.ticks, c.jitter, src, dst, c.traffic);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), d, blocks * 24, hipMemcpyDeviceToHost);
        std::map<std::tuple<int, int, int, int>, std::vector<std::pair<unsigned long long, int>>> cu;
        double dur = 0;
        for (int b = 0; b < blocks; ++b) {
            const unsigned long long hw = h[3 * b + 2];
            const unsigned id = (unsigned)hw, xcc = (unsigned)(hw >> 32) & 0xf;
            auto key = std::make_tuple((int)xcc, (int)((id >> 13) & 7), (int)((id >> 12) & 1), (int)((id >> 8) & 15));
            cu[key].push_back({h[3 * b], 1});
            cu[key].push_back({h[3 * b + 1], -1});
            dur += (double)(h[3 * b + 1] - h[3 * b]);
        }
        double wsum = 0, tot = 0;
        std::vector<double> refill;
        for (auto& kv : cu) {
            auto& ev = kv.second;
            std::sort(ev.begin(), ev.end());
            int cur = 0;
            unsigned long long last = ev[0].first;
            std::vector<unsigned long long> freed;
            size_t fi = 0;
            for (auto& e : ev) {
                wsum += (double)cur * (double)(e.first - last);
                tot += (double)(e.first - last);
                cur += e.second;
                last = e.first;
                if (e.second < 0) freed.push_back(e.first);
                else if (fi < freed.size()) refill.push_back((double)(e.first - freed[fi++]) * 0.01);
            }
        }
        std::sort(refill.begin(), refill.end());
        double mean = 0;
        for (double r : refill) mean += r;
        printf("%-28s kernel %7.3f ms  mean block %5.1f us  CUs %3zu  resident blocks per CU %.2f  refill delay us: p50 %6.2f  p90 %6.2f  mean %6.2f\n",
               c.name, ms, dur / blocks * 0.01, cu.size(), wsum / tot, refill.empty() ? 0.0 : refill[refill.size() / 2],
               refill.empty() ? 0.0 : refill[refill.size() * 9 / 10], refill.empty() ? 0.0 : mean / refill.size());
        fflush(stdout);
    }
    return 0;
}
