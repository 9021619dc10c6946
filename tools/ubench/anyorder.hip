// Micro-benchmark: can a chain of DEPENDENT launches on one stream drop the in-order barrier (hipExtAnyOrderLaunch) and carry
// the dependency itself -- a per-launch "done" counter the producer's workgroups bump (agent-scope release) and the consumer's
// workgroups wait on (agent-scope acquire) AFTER their own prologue -- so that the next launch's dispatch + prologue run under
// the tail of the previous one?  Every launch stamps first start / last end (100-MHz counter); every workgroup of launch k
// reads a chunk a DIFFERENT workgroup (another XCD) of launch k-1 wrote and checks it, so a visibility hole shows as a count.
//   hipcc --offload-arch=gfx950 -O3 -o anyorder tools/ubench/anyorder.hip && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args {
    unsigned long long* st;      // [2 * launches] first start / last end
    unsigned* done;              // [launches + 1] workgroups of launch k that have finished
    unsigned* bad;               // [2] mismatching elements / wait timeouts
    const unsigned* rd; unsigned* wr;
    int launch, chunk, spin_us, prologue_us, use_flags, coh_st, coh_ld;
};

__global__ __launch_bounds__(256) void k(Args a) {
    extern __shared__ unsigned lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) atomicMin(&a.st[2 * a.launch], t0);
    // prologue: independent of the producer (weights -> LDS in the real kernels)
    if (a.prologue_us > 0) {
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.prologue_us * 100ull) __builtin_amdgcn_s_sleep(4);
        lds[threadIdx.x] = a.launch;
    }
    if (a.use_flags && a.launch > 0) {
        if (threadIdx.x == 0) {
            int it = 0;
            while (__hip_atomic_load(&a.done[a.launch - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                __builtin_amdgcn_s_sleep(2);
                if (++it > (1 << 22)) { atomicAdd(&a.bad[1], 1u); break; }     // bounded: never hang the box
            }
        }
        __syncthreads();
        if (!a.coh_ld) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave: agent-scope acquire (buffer_inv sc1)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // body: read what workgroup (b + 1) of the previous launch wrote, check it, write + 1 into this workgroup's chunk
    const int src = (blockIdx.x + 1) % gridDim.x;
    unsigned nbad = 0;
    for (int i = threadIdx.x; i < a.chunk; i += blockDim.x) {
        const unsigned* rp = &a.rd[(size_t)src * a.chunk + i];
        const unsigned v = a.coh_ld ? __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *rp;   // sc1: coherent at the device
        if (v != (unsigned)a.launch) ++nbad;
        unsigned* wp = &a.wr[(size_t)blockIdx.x * a.chunk + i];
        if (a.coh_st) __hip_atomic_store(wp, (unsigned)a.launch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
        else *wp = (unsigned)a.launch + 1u;
    }
    if (nbad) atomicAdd(&a.bad[0], nbad);
    if (a.spin_us > 0)
        while (__builtin_amdgcn_s_memrealtime() - t1 < (unsigned long long)a.spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are acknowledged ...
    __syncthreads();                                  // ... and so is every wave's
    if (threadIdx.x == 0) {
        if (a.use_flags && !a.coh_st) __hip_atomic_fetch_add(&a.done[a.launch], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (a.use_flags && a.coh_st) __hip_atomic_fetch_add(&a.done[a.launch], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(&a.st[2 * a.launch + 1], __builtin_amdgcn_s_memrealtime());
    }
}

int main() {
    const int NL = 40, WG = 256;
    unsigned long long* st; unsigned *done, *bad, *b0, *b1;
    hipMalloc(&st, NL * 16); hipMalloc(&done, (NL + 1) * 4); hipMalloc(&bad, 8);
    hipMalloc(&b0, 256 << 20); hipMalloc(&b1, 256 << 20);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    struct Case { const char* name; int anyorder, flags, spin, pro, mb, cst, cld; };
    const Case cases[] = {
        {"in-order launches, 1 MB", 0, 0, 20, 0, 1},
        {"in-order launches, 64 MB", 0, 0, 20, 0, 64},
        {"in-order launches + flags (sanity), 64 MB", 0, 1, 20, 0, 64},
        {"ANY-ORDER + flags, 1 MB", 1, 1, 20, 0, 1},
        {"ANY-ORDER + flags, 64 MB", 1, 1, 20, 0, 64},
        {"in-order, 3 us prologue, 64 MB", 0, 0, 20, 3, 64},
        {"ANY-ORDER + flags, 3 us prologue, 64 MB", 1, 1, 20, 3, 64},
        {"ANY-ORDER + flags, no spin, 64 MB", 1, 1, 0, 0, 64},
        {"in-order, no spin, 64 MB", 0, 0, 0, 0, 64},
        {"ANY-ORDER without flags (expect mismatches if honoured), 64 MB", 1, 0, 20, 0, 64},
        {"in-order launches, 1 MB (again, warm)", 0, 0, 20, 0, 1},
        {"ANY-ORDER + flags, sc1 stores (no wbl2), 1 MB", 1, 1, 20, 0, 1, 1, 0},
        {"ANY-ORDER + flags, sc1 loads (no inv), 1 MB", 1, 1, 20, 0, 1, 0, 1},
        {"ANY-ORDER + flags, sc1 stores + sc1 loads, 1 MB", 1, 1, 20, 0, 1, 1, 1},
        {"ANY-ORDER + flags, sc1 stores (no wbl2), 64 MB", 1, 1, 20, 0, 64, 1, 0},
        {"ANY-ORDER + flags, sc1 loads (no inv), 64 MB", 1, 1, 20, 0, 64, 0, 1},
        {"ANY-ORDER + flags, sc1 stores + sc1 loads, 64 MB", 1, 1, 20, 0, 64, 1, 1},
        {"in-order, sc1 stores + sc1 loads, 64 MB", 0, 0, 20, 0, 64, 1, 1},
        {"ANY-ORDER + flags, sc1 stores + sc1 loads, 3 us prologue, 1 MB", 1, 1, 20, 3, 1, 1, 1},
        {"in-order, 3 us prologue, 1 MB", 0, 0, 20, 3, 1, 0, 0},
    };
    for (const Case& c : cases) {
        std::vector<unsigned long long> h(2 * NL);
        for (int i = 0; i < NL; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        hipMemcpy(st, h.data(), NL * 16, hipMemcpyHostToDevice);
        hipMemset(done, 0, (NL + 1) * 4); hipMemset(bad, 0, 8);
        hipMemset(b0, 0, 256 << 20); hipMemset(b1, 0xff, 256 << 20);
        hipDeviceSynchronize();
        const int chunk = (c.mb << 18) / WG;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        for (int i = 0; i < NL; ++i) {
            Args a{st, done, bad, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, i, chunk, c.spin, c.pro, c.flags, c.cst, c.cld};
            hipExtLaunchKernelGGL(k, dim3(WG), dim3(256), 4096, s, nullptr, nullptr, c.anyorder ? hipExtAnyOrderLaunch : 0, a);
        }
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(h.data(), st, NL * 16, hipMemcpyDeviceToHost);
        std::vector<double> gaps, spans;
        for (int i = 10; i + 1 < NL; ++i) {
            gaps.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) / 100.0);
            spans.push_back((double)(h[2 * i + 1] - h[2 * i]) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
        printf("%-64s period %7.2f us  span %7.2f us  next first start - last end: median %6.2f us (min %6.2f, max %6.2f)  mismatches %u  wait timeouts %u\n",
               c.name, ms * 1e3 / NL, spans[spans.size() / 2], gaps[gaps.size() / 2], gaps.front(), gaps.back(), hb[0], hb[1]);
        fflush(stdout);
    }
    return 0;
}
