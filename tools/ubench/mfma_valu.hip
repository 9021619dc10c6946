// Micro-benchmark: do VALU instructions of one wave issue beside the MFMAs of ANOTHER wave of the same SIMD on gfx950?
// One workgroup of 8 waves (two per SIMD: waves w and w + 4).  Waves 0-3 run a chain of v_mfma_f32_32x32x16_bf16 (four
// independent accumulators), waves 4-7 a chain of fp32 VALU adds / max / shifts (the fused bottleneck's epilogue mix).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu tools/ubench/mfma_valu.hip && ./mfma_valu
// Prints shader clocks for: MFMA waves alone, VALU waves alone, both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    if (mfma_wave && (mode & 1)) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x & 3); b[r] = (__bf16)1.0f; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        t1 = __builtin_amdgcn_s_memtime();
        sink[threadIdx.x] = s;
    } else if (!mfma_wave && (mode & 2)) {
        float v[8];
        unsigned u = threadIdx.x * 2654435761u;
        for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {      // 4 VALU per element: add, shift (unpack), add, max
                    const float idv = __uint_as_float((u + i) << 16);
                    v[i] = fmaxf(v[i] + 0.5f + idv, 0.f);
                    asm volatile("" : "+v"(v[i]));
                }
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        t1 = __builtin_amdgcn_s_memtime();
        sink[threadIdx.x] = s;
    }
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 8 * sizeof(*out)); hipMalloc(&sink, 512 * sizeof(float));
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        unsigned long long h[8];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, mode, iters, out, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s): MFMA wave 0: %llu clk (%.1f clk / MFMA), VALU wave 4: %llu clk (%.2f clk / VALU op)\n", mode,
               mode == 1 ? "MFMA waves alone" : mode == 2 ? "VALU waves alone" : "both", h[0], (double)h[0] / (iters * 4.0), h[4],
               (double)h[4] / (iters * 4.0 * 8 * 4));
    }
    return 0;
}
