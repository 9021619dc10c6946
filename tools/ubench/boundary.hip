// Micro-benchmark: what makes a dependent kernel boundary (same stream) expensive on gfx950?  Every launch stamps its first
// start / last end on the 100-MHz counter; the GAP = next launch's first start - this launch's last end.
//   hipcc --offload-arch=gfx950 -O3 -o boundary tools/ubench/boundary.hip && ./boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args { unsigned long long* st; int launch; int spin_us; const float* rd; float* wr; int rd_elems; int wr_elems; int big_code; };

__global__ __launch_bounds__(512) void k(Args a) {
    extern __shared__ float lds[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) atomicMin(&a.st[2 * a.launch], t0);
    float acc = 0.f;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gid; i < a.rd_elems; i += gsz) acc += a.rd[i];
    for (int i = gid; i < a.wr_elems; i += gsz) a.wr[i] = acc + (float)i;
    if (a.spin_us > 0) {
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 12345.f) lds[threadIdx.x] = acc;
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) atomicMax(&a.st[2 * a.launch + 1], __builtin_amdgcn_s_memrealtime());
}

int main() {
    const int NL = 40;
    unsigned long long* st; float *rd, *wr;
    hipMalloc(&st, NL * 16); hipMalloc(&rd, 256 << 20); hipMalloc(&wr, 256 << 20);
    hipMemset(rd, 0, 256 << 20);
    struct Case { const char* name; int wgs; size_t lds; int spin; int rd_mb; int wr_mb; };
    const Case cases[] = {
        {"empty", 256, 0, 0, 0, 0},
        {"spin 20 us", 256, 0, 20, 0, 0},
        {"spin 20 us, 150 KB LDS", 256, 150 * 1024, 20, 0, 0},
        {"spin 20 us, 1 workgroup", 1, 0, 20, 0, 0},
        {"spin 20 us + read 64 MB", 256, 0, 20, 64, 0},
        {"spin 20 us + write 1 MB", 256, 0, 20, 0, 1},
        {"spin 20 us + write 16 MB", 256, 0, 20, 0, 16},
        {"spin 20 us + write 128 MB", 256, 0, 20, 0, 128},
        {"spin 20 us + read 64 + write 64 MB, 150 KB LDS", 256, 150 * 1024, 20, 64, 64},
    };
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (const Case& c : cases) {
        std::vector<unsigned long long> h(2 * NL);
        for (int i = 0; i < NL; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        hipMemcpy(st, h.data(), NL * 16, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int i = 0; i < NL; ++i) {
            Args a{st, i, c.spin, rd, wr, c.rd_mb << 18, c.wr_mb << 18, 0};
            hipLaunchKernelGGL(k, dim3(c.wgs), dim3(512), c.lds, 0, a);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, NL * 16, hipMemcpyDeviceToHost);
        std::vector<double> gaps, spans;
        for (int i = 10; i + 1 < NL; ++i) { gaps.push_back((double)(h[2 * (i + 1)] - h[2 * i + 1]) / 100.0); spans.push_back((double)(h[2 * i + 1] - h[2 * i]) / 100.0); }
        std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
        printf("%-52s period %7.2f us  in-kernel span (median) %7.2f us  GAP last end -> next first start: median %5.2f us (min %5.2f, max %5.2f)\n",
               c.name, ms * 1e3 / NL, spans[spans.size() / 2], gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
    return 0;
}
