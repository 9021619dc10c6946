// Micro-benchmark: how fast can every CU pull an L2-RESIDENT operand stream (the weights of a fused launch: every workgroup
// streams the same ~2 MB) -- through LDS-DMA (`buffer_load ... lds`, what bneck23_kernel's weight rings use) against plain
// 16-byte loads into VGPRs (a wave-private operand needs no LDS hop)?  256 (or 512) workgroups of 8 waves, each wave walking
// its own 1/8 slice of the buffer in 1-KB pieces, NF pieces in flight.
//   hipcc --offload-arch=gfx950 -O3 -o l2_feed tools/ubench/l2_feed.hip && ./l2_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

template <int MODE, int NF>   // MODE 0: LDS-DMA into a per-wave ring; 1: global -> VGPR
__global__ __launch_bounds__(512, 2) void feed(const unsigned char* __restrict__ buf, unsigned bytes, int pieces, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
    const unsigned slice = bytes / 8, base = wave * slice + lane * 16;
    unsigned off = 0;
    u32x4 acc = {0, 0, 0, 0};
    if constexpr (MODE == 0) {
        unsigned char* ring = smem + wave * (NF * 1024);
        for (int p = 0; p < pieces; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(ring + (p % NF) * 1024), 16, base + off, 0, 0, 0);
            off += 1024; if (off >= slice) off = 0;
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NF - 1) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<u32x4*>(ring + lane * 16);
    } else {
        u32x4 r[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) { r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + off), 0, 0); off += 1024; if (off >= slice) off = 0; }
        for (int p = NF; p < pieces; p += NF) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                acc ^= r[i];                                  // (consumes the oldest load: the compiler waits with a counted vmcnt)
                r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + off), 0, 0);
                off += 1024; if (off >= slice) off = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) acc ^= r[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int NF>
void run(const char* name, const unsigned char* buf, unsigned bytes, int wgs, unsigned* sink) {
    const int pieces = 4096;
    const size_t lds = MODE == 0 ? (size_t)8 * NF * 1024 : 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((feed<MODE, NF>), dim3(wgs), dim3(512), lds, 0, buf, bytes, pieces, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tb = (double)wgs * 8 * pieces * 1024 / (ms * 1e-3) / 1e12;
    printf("%-34s %4d WGs  NF %2d: %7.3f ms  %6.2f TB/s  (%5.1f B/clk/CU at 2.0 GHz over %d CUs)\n", name, wgs, NF, ms, tb,
           tb * 1e12 / (wgs < 256 ? wgs : 256) / 2.0e9, wgs < 256 ? wgs : 256);
}

int main() {
    for (unsigned mb : {2u, 16u}) {
        const unsigned bytes = mb << 20;
        unsigned char* buf; unsigned* sink;
        hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
        printf("operand buffer %u MB (%s)\n", mb, mb <= 4 ? "fits every XCD's 4-MB L2" : "Infinity-Cache resident");
        for (int wgs : {64, 256, 512}) {
            run<0, 3>("LDS-DMA (buffer_load ... lds)", buf, bytes, wgs, sink);
            run<0, 8>("LDS-DMA (buffer_load ... lds)", buf, bytes, wgs, sink);
            run<1, 4>("global -> VGPR (buffer_load_b128)", buf, bytes, wgs, sink);
            run<1, 8>("global -> VGPR (buffer_load_b128)", buf, bytes, wgs, sink);
            run<1, 16>("global -> VGPR (buffer_load_b128)", buf, bytes, wgs, sink);
        }
        hipFree(buf); hipFree(sink);
    }
    return 0;
}
