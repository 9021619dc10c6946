// Fused pair of 1x1 convolutions across a Bottleneck boundary of CLIP-RN50 layer1 (bf16 MFMA, gfx950):
//
//   y = relu( a0 . w0^T + b0  [+ a1 . w1^T + b1]  [+ res] )      [M, 256]   conv3+bn3 (+ downsample conv+bn) + identity + ReLU
//   z = relu( y . w2^T + b2 )                                     [M, N2]    the NEXT block's conv1+bn1+ReLU (N2 = 64 | 128)
//
// Replaces, per boundary, two (three with the downsample) launches of `conv_igemm` and the HBM round trip of
// y between them ([U] openai/CLIP clip/model.py Bottleneck.forward: `out = relu(bn3(conv3(out)) + identity)`
// followed by the next block's `relu(bn1(conv1(x)))`; reached from
// primitive_probing/generate_data/thor_image_features.py:109).  At 56x56 these 1x1 convs are bandwidth-bound
// (K = 64: 0.5 flop/B), so the win is bytes: y is written once and never re-read by conv1, and with the
// downsample fused the 256-channel identity tensor of block 0 is never materialised at all.
//
// Design (one workgroup of 4 waves per CU, persistent; everything wave-private after the prologue):
//   * all weights stay resident in LDS for the lifetime of the workgroup (w0|w1: 32-64 KB, w2: 32-64 KB);
//   * a wave owns a 32-pixel tile.  Pixel operands (a0, a1) are loaded straight into MFMA operand layout,
//     the residual tile as coalesced 16-B row chunks; both are PREFETCHED one tile ahead into registers
//     (1 wave/SIMD => 512 VGPRs), so ~20 KB per wave are always in flight and no barrier is ever needed;
//   * swapped MFMA operands (D[channel][pixel]): a lane owns one pixel and 4 consecutive channels per 4
//     accumulator registers.  After bias/residual/ReLU/bf16 rounding, 8 consecutive accumulator registers
//     ARE the k-slots a lane must supply for one K-step of the second GEMM -- y never leaves registers on its
//     way into conv1; the matching K permutation is applied to w2 once, while it is staged into LDS;
//   * y and z leave through a small per-wave LDS staging image so every global access is a 16-B row chunk.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: plain loads/stores, SROA-friendly
constexpr int PX = 32;         // pixels per wave tile
constexpr int NY = 256;        // channels of y
constexpr int KA = 64;         // channels of a0 / a1
constexpr int SP = 272;        // staging pitch in bytes of a 128-channel half row (+16: 2-way instead of 32-way conflicts)
constexpr int STG = PX * SP;   // staging bytes per wave

struct PairArgs {
    const uint16_t *a0, *a1, *w0, *w1, *w2, *res;
    const float *b0, *b1, *b2;
    uint16_t *y, *z;
    int ntiles;
    uint16_t* yp;   // PL: AvgPool2d(2) of y, [B, H/2, W/2, 256]
    int H, W;       // PL: frame geometry (tiles are 4 x 8 pixel blocks = 8 pooling windows)
};

__device__ __forceinline__ float pq_xor1(float v) {   // lane ^ 1 within a quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float pq_xor2(float v) {   // lane ^ 2 within a quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

__device__ __forceinline__ int pw_off(int row, int chunk) {   // 128-B rows, XOR swizzle on the 16-B chunk
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Software-pipelined LDS fragment stream for kernels that run ONE wave per SIMD: left to the compiler every MFMA gets its
// own `ds_read_b128 -> s_waitcnt lgkmcnt(0)` in front (it sinks operand reads to their use), i.e. the full LDS latency per
// MFMA with nothing to hide it.  Here the reads are inline asm, LEAD steps ahead of the MFMA that consumes them, and the
// waits carry the fragment register so the MFMA cannot be scheduled above its wait (LDS returns data in order:
// "lgkmcnt(n)" = all but the newest n have arrived; compiler-generated LDS traffic in between only makes the waits
// conservative).  addr(i): LDS byte address (VGPR) and immediate offset of step i; mma(i, frag) consumes it.
template <int I, int LEAD, class AddrFn>
__device__ __forceinline__ void lsm_issue(u32x4 (&ring)[LEAD], AddrFn& addr) {
    const auto a = addr(std::integral_constant<int, I>{});
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[I % LEAD]) : "v"(a.first), "n"(decltype(a.second)::value));
}
template <int I, int NSTEP, int LEAD, class AddrFn, class MmaFn>
__device__ __forceinline__ void lsm_step(u32x4 (&ring)[LEAD], AddrFn& addr, MmaFn& mma) {
    if constexpr (I < NSTEP) {
        constexpr int after = (LEAD - 1 < NSTEP - 1 - I) ? LEAD - 1 : NSTEP - 1 - I;   // reads issued after read I
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[I % LEAD]) : "n"(after));
        mma(std::integral_constant<int, I>{}, ring[I % LEAD]);
        if constexpr (I + LEAD < NSTEP) lsm_issue<I + LEAD, LEAD>(ring, addr);
        lsm_step<I + 1, NSTEP, LEAD>(ring, addr, mma);
    }
}
template <int I, int N, int LEAD, class AddrFn>
__device__ __forceinline__ void lsm_prologue(u32x4 (&ring)[LEAD], AddrFn& addr) {
    if constexpr (I < N) {
        lsm_issue<I, LEAD>(ring, addr);
        lsm_prologue<I + 1, N, LEAD>(ring, addr);
    }
}
template <int NSTEP, int LEAD, class AddrFn, class MmaFn>
__device__ __forceinline__ void lds_stream_mfma(AddrFn addr, MmaFn mma) {
    u32x4 ring[LEAD];
    lsm_prologue<0, (LEAD < NSTEP ? LEAD : NSTEP), LEAD>(ring, addr);
    lsm_step<0, NSTEP, LEAD>(ring, addr, mma);
}

// PL: the tile is a 4 x 8 block of pixels in quad order (row r of the tile = window (r >> 2), corner r & 3), so a 2 x 2
// pooling window is one lane quad and the kernel also emits AvgPool2d(2)(y) -- the input of the next layer's
// downsample path -- instead of a separate pooling pass over y.
template <bool TWO, bool RES, int N2, bool PL>
__global__ __launch_bounds__(256, 1) void conv1x1_pair_kernel(PairArgs p) {
    static_assert(!PL || N2 == 128, "pooled variant: the layer-1 -> layer-2 boundary");
    constexpr int W0_BYTES = (TWO ? 2 : 1) * NY * 128;
    constexpr int W2_BYTES = 4 * N2 * 128;
    constexpr int FN2 = N2 / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* sW0 = sm;
    unsigned char* sW2 = sm + W0_BYTES;
    float* sBy = reinterpret_cast<float*>(sm + W0_BYTES + W2_BYTES);
    float* sBz = sBy + NY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* stg = sm + W0_BYTES + W2_BYTES + (NY + N2) * 4 + wave * STG;
    const int px = lane & 31, h = lane >> 5;
    // LDS byte address of the weight images and this lane's per-K-step fragment offsets (pw_off(32 j + px, 2 ks + h) minus
    // the 32 j rows, which go into the instruction's immediate offset: the swizzle term only depends on px)
    const unsigned w_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)sW0;
    unsigned wk_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wk_off[ks] = (unsigned)pw_off(px, 2 * ks + h);
    // tile row r -> pixel offset from the tile's first pixel
    auto roff = [&](int r) -> int {
        if (!PL) return r;
        return (2 * (r >> 4) + ((r >> 1) & 1)) * p.W + 2 * ((r >> 2) & 3) + (r & 1);
    };
    const int roff_px = roff(px);
    int rrow[8];                               // rows touched by the coalesced 16-lanes-per-row accesses
#pragma unroll
    for (int j = 0; j < 8; ++j) rrow[j] = roff(4 * j + (lane >> 4));
    auto tile_base = [&](int tile) -> long {   // first pixel of the tile (wave-uniform)
        if (!PL) return (long)tile * PX;
        const int t = __builtin_amdgcn_readfirstlane(tile);
        const int tpr = p.W / 8, tpi = (p.H / 4) * tpr;
        const int b = t / tpi, rem = t - b * tpi, ty = rem / tpr, tx = rem - ty * tpr;
        return ((long)b * p.H + 4 * ty) * p.W + 8 * tx;
    };

    // ---- prologue: weights -> LDS (once per workgroup) ----
    ec_stage_all<(TWO ? 2 : 1) * NY * 8, 256, uint4>(
        tid,
        [&](int idx) {
            const int kt = idx / (NY * 8), r = (idx / 8) % NY, c = idx & 7;
            return *reinterpret_cast<const uint4*>((kt == 0 ? p.w0 : p.w1) + r * KA + c * 8);
        },
        [&](int idx, const uint4& v) {
            const int kt = idx / (NY * 8), r = (idx / 8) % NY, c = idx & 7;
            *reinterpret_cast<uint4*>(sW0 + kt * (NY * 128) + pw_off(r, c)) = v;
        });
    // w2 with the K permutation of the register-chained operand: K-step s = (j, gp), half hh, element e
    //   <->  channel 32 j + 16 gp + 8 (e >> 2) + 4 hh + (e & 3)
    ec_stage_all<N2 * 32, 256, uint4>(
        tid,
        [&](int idx) {
            const int n = idx >> 5, q = idx & 31;
            const int s = q >> 1, hh = q & 1;
            const int c0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * hh;
            const uint2 lo = *reinterpret_cast<const uint2*>(p.w2 + n * NY + c0);
            const uint2 hi = *reinterpret_cast<const uint2*>(p.w2 + n * NY + c0 + 8);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        },
        [&](int idx, const uint4& v) {
            const int n = idx >> 5, q = idx & 31;
            const int s = q >> 1, hh = q & 1;
            *reinterpret_cast<uint4*>(sW2 + (s >> 2) * (N2 * 128) + pw_off(n, 2 * (s & 3) + hh)) = v;
        });
    for (int i = tid; i < NY; i += 256) sBy[i] = p.b0[i] + (TWO ? p.b1[i] : 0.f);
    for (int i = tid; i < N2; i += 256) sBz[i] = p.b2[i];
    __syncthreads();

    const int gw = blockIdx.x * 4 + wave, GW = gridDim.x * 4;
    int t = gw;
    if (t >= p.ntiles) return;

    // ---- register prefetch of the next tile ----
    u32x4 a0n[4], a1n[TWO ? 4 : 1], rn[RES ? 16 : 1];
    auto prefetch = [&](int tile) {
        const long m0 = tile_base(tile);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a0n[ks] = *reinterpret_cast<const u32x4*>(p.a0 + (m0 + roff_px) * KA + (2 * ks + h) * 8);
            if (TWO) a1n[ks] = *reinterpret_cast<const u32x4*>(p.a1 + (m0 + roff_px) * KA + (2 * ks + h) * 8);
        }
        if constexpr (RES) {   // fold over compile-time indices: keeps the array in registers (a runtime-indexed loop lands in scratch)
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((rn[I] = *reinterpret_cast<const u32x4*>(p.res + (m0 + rrow[I & 7]) * NY + (I >> 3) * 128 +
                                                          (((I & 7) * 64 + lane) & 15) * 8)), ...);
            }(std::make_integer_sequence<int, 16>{});
        }
    };
    prefetch(t);

    for (;;) {
        const long m0 = tile_base(t);
        u32x4 a0c[4], a1c[TWO ? 4 : 1], rc[RES ? 16 : 1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { a0c[ks] = a0n[ks]; if (TWO) a1c[ks] = a1n[ks]; }
        if constexpr (RES) {
            [&]<int... I>(std::integer_sequence<int, I...>) { ((rc[I] = rn[I]), ...); }(std::make_integer_sequence<int, 16>{});
        }
        const int tn = t + GW;
        const bool more = tn < p.ntiles;
        if (more) prefetch(tn);

        // ---- GEMM 1: acc[j] = D[channel 32j..][pixel] ----
        f32x16_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        {   // step i = (ks, operand 0 | 1, j): weight fragment W[32 j + px][16 ks + 8 h ..] of sW0 (+ the second matrix)
            constexpr int PER_KS = (TWO ? 16 : 8);
            lds_stream_mfma<4 * PER_KS, 8>(
                [&](auto ic) {
                    constexpr int i = decltype(ic)::value, ks = i / PER_KS, r = i % PER_KS, j = r & 7, second = r >> 3;
                    return std::pair<unsigned, std::integral_constant<int, second * (NY * 128) + j * 4096>>{w_lds + wk_off[ks], {}};
                },
                [&](auto ic, const u32x4& wf) {
                    constexpr int i = decltype(ic)::value, ks = i / PER_KS, r = i % PER_KS, j = r & 7, second = r >> 3;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                                     __builtin_bit_cast(bf16x8_t, second ? a1c[TWO ? ks : 0] : a0c[ks]),
                                                                     acc[j], 0, 0, 0);
                });
        }

        // ---- epilogue 1 (two 128-channel halves) + the packed operand of GEMM 2 ----
        uint4 P[8][2];
        long pool_base = 0;                       // PL: first pooled pixel of the tile (8 pooled pixels = a 2 x 4 block)
        if constexpr (PL) {
            const int tt = __builtin_amdgcn_readfirstlane(t);
            const int tpr = p.W / 8, tpi = (p.H / 4) * tpr;
            const int b = tt / tpi, rem = tt - b * tpi, ty = rem / tpr, tx = rem - ty * tpr;
            pool_base = ((long)b * (p.H / 2) + 2 * ty) * (p.W / 2) + 4 * tx;
        }
        auto half_epilogue = [&](auto hfc) {
            constexpr int hf = decltype(hfc)::value;   // compile-time: keeps rc[] / acc[] / P[] in registers
            if constexpr (RES) {
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    ((*reinterpret_cast<u32x4*>(stg + ((I * 64 + lane) >> 4) * SP + ((I * 64 + lane) & 15) * 16) = rc[hf * 8 + I]), ...);
                }(std::make_integer_sequence<int, 8>{});
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = hf * 4 + jj;
                unsigned pk[8];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lc = 32 * jj + 8 * g + 4 * h;                       // channel within the half
                    const float4 bv = *reinterpret_cast<const float4*>(sBy + hf * 128 + lc);
                    float v0 = acc[j][4 * g + 0] + bv.x, v1 = acc[j][4 * g + 1] + bv.y;
                    float v2 = acc[j][4 * g + 2] + bv.z, v3 = acc[j][4 * g + 3] + bv.w;
                    uint2* slot = reinterpret_cast<uint2*>(stg + px * SP + lc * 2);
                    if (RES) {
                        const uint2 rr = *slot;
                        v0 += ec_lo(rr.x); v1 += ec_hi(rr.x); v2 += ec_lo(rr.y); v3 += ec_hi(rr.y);
                    }
                    v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                    uint2 o;
                    o.x = ec_pack2(v0, v1);
                    o.y = ec_pack2(v2, v3);
                    *slot = o;
                    pk[2 * g] = o.x; pk[2 * g + 1] = o.y;
                }
                P[j][0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                P[j][1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
            if constexpr (PL) {
                // AvgPool2d(2)(y) from the staged half: window w = tile rows 4w..4w+3 (quad order); a lane takes two
                // (window, 8-channel chunk) items: mean of the ROUNDED bf16 values, summed as (r0 + r1) + (r2 + r3) like
                // avgpool2_kernel, straight to global memory (was: two DPP adds per value in the epilogue loop + a
                // staging pass -- ~4x the VALU work of this)
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2) {
                    const int item = it2 * 64 + lane, w = item >> 4, c = item & 15;
                    const uint4 r0 = *reinterpret_cast<const uint4*>(stg + (4 * w + 0) * SP + c * 16);
                    const uint4 r1 = *reinterpret_cast<const uint4*>(stg + (4 * w + 1) * SP + c * 16);
                    const uint4 r2 = *reinterpret_cast<const uint4*>(stg + (4 * w + 2) * SP + c * 16);
                    const uint4 r3 = *reinterpret_cast<const uint4*>(stg + (4 * w + 3) * SP + c * 16);
                    auto avg2 = [](unsigned a, unsigned b, unsigned cc, unsigned d) {
                        const float lo = 0.25f * ((ec_lo(a) + ec_lo(b)) + (ec_lo(cc) + ec_lo(d)));
                        const float hi = 0.25f * ((ec_hi(a) + ec_hi(b)) + (ec_hi(cc) + ec_hi(d)));
                        return ec_pack2(lo, hi);
                    };
                    u32x4 po;
                    po[0] = avg2(r0.x, r1.x, r2.x, r3.x); po[1] = avg2(r0.y, r1.y, r2.y, r3.y);
                    po[2] = avg2(r0.z, r1.z, r2.z, r3.z); po[3] = avg2(r0.w, r1.w, r2.w, r3.w);
                    *reinterpret_cast<u32x4*>(p.yp + (pool_base + (w >> 2) * (p.W / 2) + (w & 3)) * NY + hf * 128 + c * 8) = po;
                }
            }
            if (!PL || p.y) {   // (layer-1 -> layer-2 boundary: nobody reads the full-resolution y -- its consumers are z and y_pooled)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + (idx >> 4) * SP + (idx & 15) * 16);
                __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p.y + (m0 + rrow[i]) * NY + hf * 128 + (idx & 15) * 8));
            }
            }
        };
        half_epilogue(std::integral_constant<int, 0>{});
        half_epilogue(std::integral_constant<int, 1>{});

        // ---- GEMM 2, pixel operand straight from registers: z[n][pixel] ----
        f32x16_t acc2[FN2];
#pragma unroll
        for (int n = 0; n < FN2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
        lds_stream_mfma<16 * FN2, 8>(
            [&](auto ic) {
                constexpr int i = decltype(ic)::value, st = i / FN2, n = i % FN2;
                return std::pair<unsigned, std::integral_constant<int, (st >> 2) * (N2 * 128) + n * 4096>>{w_lds + (unsigned)W0_BYTES + wk_off[st & 3], {}};
            },
            [&](auto ic, const u32x4& wf) {
                constexpr int i = decltype(ic)::value, st = i / FN2, n = i % FN2;
                acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                                  __builtin_bit_cast(bf16x8_t, P[st >> 1][st & 1]), acc2[n], 0, 0, 0);
            });
        // ---- epilogue 2 ----
        constexpr int ZP = N2 * 2 + 16;            // staging pitch of a z row
        constexpr int ZC = N2 / 8;                 // 16-B chunks per z row
#pragma unroll
        for (int n = 0; n < FN2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 32 * n + 8 * g + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>(sBz + lc);
                uint2 o;
                o.x = ec_pack2(fmaxf(acc2[n][4 * g + 0] + bv.x, 0.f), fmaxf(acc2[n][4 * g + 1] + bv.y, 0.f));
                o.y = ec_pack2(fmaxf(acc2[n][4 * g + 2] + bv.z, 0.f), fmaxf(acc2[n][4 * g + 3] + bv.w, 0.f));
                *reinterpret_cast<uint2*>(stg + px * ZP + lc * 2) = o;
            }
#pragma unroll
        for (int i = 0; i < PX * ZC / 64; ++i) {
            const int idx = i * 64 + lane;
            const uint4 v = *reinterpret_cast<const uint4*>(stg + (idx / ZC) * ZP + (idx % ZC) * 16);
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p.z + (m0 + (PL ? rrow[i] : idx / ZC)) * N2 + (idx % ZC) * 8));
        }
        if (!more) break;
        t = tn;
    }
}

template <bool TWO, bool RES, int N2, bool PL = false>
int launch_pair(const PairArgs& p, hipStream_t s) {
    constexpr size_t lds = (size_t)(TWO ? 2 : 1) * NY * 128 + 4 * N2 * 128 + (NY + N2) * 4 + 4 * STG;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv1x1_pair_kernel<TWO, RES, N2, PL>;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int wgs = (p.ntiles + 3) / 4 < 256 ? (p.ntiles + 3) / 4 : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}


// -----------------------------------------------------------------------------------------------------------------
// Layer-2 boundary (a0 [M,128] . w0[512,128]^T + res -> y [M,512];  z = relu(y . w2[128,512]^T + b2) [M,128]).
// The two weight matrices are 128 KB each -- they cannot both sit in LDS -- so here they live in REGISTERS: the four
// waves of a workgroup split the OUTPUT CHANNELS (wave w owns y channels [128w, 128w+128) and z channels
// [32w, 32w+32)), each holding its 32-KB slices of w0 and w2 as 2 x 128 VGPRs of ready-made MFMA fragments
// (1 wave / SIMD, 512-register budget).  A workgroup walks 32-pixel tiles; the y tile (32 x 512 bf16) is exchanged
// between the waves through a double-buffered LDS image (one barrier per tile), pixel operands and the residual
// slice of the next tile are prefetched in registers.  Bytes per pixel: 256 + 1024 in, 1024 + 256 out -- y is never
// re-read from HBM by the next block's conv1.
// -----------------------------------------------------------------------------------------------------------------
constexpr int K2 = 128, NY2 = 512, NZ2 = 128;
constexpr int YP2 = NY2 * 2 + 16;           // y-tile row pitch: 260 dwords == 4 (mod 64): 16-lane read groups conflict-free
constexpr int SP2 = 256 + 16;               // per-wave staging pitch (128-channel slice rows)

struct Pair2Args {
    const uint16_t *a0, *w0, *w2, *res;
    const float *b0, *b2;
    uint16_t *y, *z;
    int ntiles;
};

__global__ __launch_bounds__(256, 1) void conv1x1_pair512_kernel(Pair2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* ytile = sm;                                        // [2][32][YP2]
    float* sBy = reinterpret_cast<float*>(sm + 2 * PX * YP2);         // [512]
    float* sBz = sBy + NY2;                                           // [128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* stg = sm + 2 * PX * YP2 + (NY2 + NZ2) * 4 + wave * (PX * SP2);
    unsigned char* atile = sm + 2 * PX * YP2 + (NY2 + NZ2) * 4 + 4 * (PX * SP2);   // [2][32][SP2]: the pixel operand tile
    const int px = lane & 31, h = lane >> 5;

    for (int i = tid; i < NY2; i += 256) sBy[i] = p.b0[i];
    for (int i = tid; i < NZ2; i += 256) sBz[i] = p.b2[i];

    // ---- this wave's weight slices as MFMA fragments, in registers for the lifetime of the workgroup ----
    u32x4 w0f[4][8];      // y channels 128 w + 32 j + (lane & 31), K-step ks: k = 16 ks + 8 h ..
    u32x4 w2f[32];        // z channel   32 w + (lane & 31),        K-step s : k = 16 s + 8 h ..
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((w0f[I >> 3][I & 7] = *reinterpret_cast<const u32x4*>(p.w0 + (long)(128 * wave + 32 * (I >> 3) + px) * K2 + 16 * (I & 7) + 8 * h)), ...);
    }(std::make_integer_sequence<int, 32>{});
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((w2f[I] = *reinterpret_cast<const u32x4*>(p.w2 + (long)(32 * wave + px) * NY2 + 16 * I + 8 * h)), ...);
    }(std::make_integer_sequence<int, 32>{});
    __syncthreads();

    int t = blockIdx.x;
    if (t >= p.ntiles) return;      // (uniform per workgroup)
    const int G = gridDim.x;

    // The pixel operand tile (32 x 128 bf16 = 8 KB) is the same for all four waves: each wave fetches a quarter of it
    // (2 coalesced loads per lane), the quarters meet in LDS behind the barrier of the PREVIOUS tile.
    u32x4 an[2], rn[8];
    auto prefetch = [&](int tile) {
        const long m0 = (long)tile * PX;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((an[I] = *reinterpret_cast<const u32x4*>(p.a0 + (m0 + 8 * wave + 4 * I + (lane >> 4)) * K2 + (lane & 15) * 8)), ...);
        }(std::make_integer_sequence<int, 2>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((rn[I] = *reinterpret_cast<const u32x4*>(p.res + (m0 + ((I * 64 + lane) >> 4)) * NY2 + 128 * wave + ((I * 64 + lane) & 15) * 8)), ...);
        }(std::make_integer_sequence<int, 8>{});
    };
    auto publish_a = [&](int buf) {      // this wave's 8 rows of the prefetched operand tile -> LDS
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((*reinterpret_cast<u32x4*>(atile + buf * (PX * SP2) + (8 * wave + 4 * I + (lane >> 4)) * SP2 + (lane & 15) * 16) = an[I]), ...);
        }(std::make_integer_sequence<int, 2>{});
    };
    prefetch(t);
    publish_a(0);
    __syncthreads();

    for (int it = 0;; ++it) {
        const long m0 = (long)t * PX;
        unsigned char* yt = ytile + (it & 1) * (PX * YP2);
        u32x4 rc[8];
        [&]<int... I>(std::integer_sequence<int, I...>) { ((rc[I] = rn[I]), ...); }(std::make_integer_sequence<int, 8>{});
        const unsigned char* at = atile + (it & 1) * (PX * SP2);
        const int tn = t + G;
        const bool more = tn < p.ntiles;
        if (more) prefetch(tn);

        // ---- GEMM 1: this wave's 128 y channels of the 32 pixels ----
        f32x16_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        {   // pixel-operand fragments as a software-pipelined LDS stream (lds_stream_mfma): one read per K-step, 4 MFMAs on it
            const unsigned a_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) const unsigned char*)at +
                                   (unsigned)(px * SP2 + 16 * h);
            lds_stream_mfma<8, 6>(
                [&](auto ic) { return std::pair<unsigned, std::integral_constant<int, 32 * decltype(ic)::value>>{a_lds, {}}; },
                [&](auto ic, const u32x4& avr) {
                    constexpr int ks = decltype(ic)::value;
                    const bf16x8_t av = __builtin_bit_cast(bf16x8_t, avr);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w0f[j][ks]), av, acc[j], 0, 0, 0);
                });
        }

        // ---- epilogue 1: residual slice -> staging; bias + residual + ReLU -> bf16 -> staging AND shared y tile ----
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((*reinterpret_cast<u32x4*>(stg + ((I * 64 + lane) >> 4) * SP2 + ((I * 64 + lane) & 15) * 16) = rc[I]), ...);
        }(std::make_integer_sequence<int, 8>{});
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 32 * j + 8 * g + 4 * h;                    // channel within the wave's slice
                const float4 bv = *reinterpret_cast<const float4*>(sBy + 128 * wave + lc);
                uint2* slot = reinterpret_cast<uint2*>(stg + px * SP2 + lc * 2);
                const uint2 rr = *slot;
                const float v0 = fmaxf(acc[j][4 * g + 0] + bv.x + ec_lo(rr.x), 0.f), v1 = fmaxf(acc[j][4 * g + 1] + bv.y + ec_hi(rr.x), 0.f);
                const float v2 = fmaxf(acc[j][4 * g + 2] + bv.z + ec_lo(rr.y), 0.f), v3 = fmaxf(acc[j][4 * g + 3] + bv.w + ec_hi(rr.y), 0.f);
                uint2 o;
                o.x = ec_pack2(v0, v1);
                o.y = ec_pack2(v2, v3);
                *slot = o;
                *reinterpret_cast<uint2*>(yt + px * YP2 + (128 * wave + lc) * 2) = o;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (idx >> 4) * SP2 + (idx & 15) * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.y + (m0 + (idx >> 4)) * NY2 + 128 * wave + (idx & 15) * 8));
        }
        if (more) publish_a((it + 1) & 1);   // next tile's operand quarter (its last readers were GEMM 1 of tile it-1)
        __syncthreads();            // the whole y tile is in LDS (and the other buffer's readers of tile it-1 are done)

        // ---- GEMM 2: this wave's 32 z channels from the shared y tile ----
        f32x16_t acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        {
            const unsigned y_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) const unsigned char*)yt +
                                   (unsigned)(px * YP2 + 16 * h);
            lds_stream_mfma<32, 8>(
                [&](auto ic) { return std::pair<unsigned, std::integral_constant<int, 32 * decltype(ic)::value>>{y_lds, {}}; },
                [&](auto ic, const u32x4& yv) {
                    constexpr int I = decltype(ic)::value;
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w2f[I]), __builtin_bit_cast(bf16x8_t, yv), acc2, 0, 0, 0);
                });
        }
        // ---- epilogue 2: 32 pixels x 32 channels -> 64-B row pieces ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int lc = 8 * g + 4 * h;
            const float4 bv = *reinterpret_cast<const float4*>(sBz + 32 * wave + lc);
            uint2 o;
            o.x = ec_pack2(fmaxf(acc2[4 * g + 0] + bv.x, 0.f), fmaxf(acc2[4 * g + 1] + bv.y, 0.f));
            o.y = ec_pack2(fmaxf(acc2[4 * g + 2] + bv.z, 0.f), fmaxf(acc2[4 * g + 3] + bv.w, 0.f));
            *reinterpret_cast<uint2*>(stg + px * 80 + lc * 2) = o;          // 64-B rows, pitch 80
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane;                                    // 32 rows x 4 chunks
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (idx >> 2) * 80 + (idx & 3) * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.z + (m0 + (idx >> 2)) * NZ2 + 32 * wave + (idx & 3) * 8));
        }
        if (!more) break;
        t = tn;
    }
}

int launch_pair512(const Pair2Args& p, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)PX * YP2 + (NY2 + NZ2) * 4 + 4 * (size_t)PX * SP2 + 2 * (size_t)PX * SP2;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_pair512_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int wgs = p.ntiles < 256 ? p.ntiles : 256;
    hipLaunchKernelGGL(conv1x1_pair512_kernel, dim3((unsigned)wgs), dim3(256), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}


// -----------------------------------------------------------------------------------------------------------------
// Single bandwidth-bound 1x1 convolution with the WEIGHTS IN REGISTERS (same scheme as the layer-2 boundary kernel,
// without the second GEMM):  y = act(a . w^T + b (+ res)),  a [M, K], w [N, K], y [M, N].
// The four waves of the persistent workgroup split the N output channels; each keeps its [N/4, K] weight slice as
// MFMA fragments in <= 256 VGPRs, the 32-pixel operand tile is fetched once per workgroup (a quarter per wave) into a
// double-buffered LDS image, one barrier per tile.  Used where K*N*2 B <= 256 KB and the tiled kernel is far from the
// HBM roof: the layer-2 downsample conv (256 -> 512 @28x28), layer-2's last conv3 (128 -> 512 + residual) and
// layer-3's first conv1 (512 -> 256 @28x28).
// -----------------------------------------------------------------------------------------------------------------
struct RegwArgs {
    const uint16_t *a, *w, *res;
    const float* b;
    uint16_t* y;
    int ntiles;
    int ldy;      // row stride of y / res in elements: N * (number of channel groups); blockIdx.y = channel group
    uint16_t* yp; // PL: AvgPool2d(2)(y), pooled pixel q at yp + q * ldp (a column block of a wider tensor)
    int ldp;
    int H, W;     // PL: frame geometry
};

// PL (round 6): the tile's 32 rows are EIGHT 2 x 2 POOLING WINDOWS in quad order -- tile row r = window 8 t + (r >> 2) of the
// launch (windows numbered frame-major, then row-major over the H/2 x W/2 pooled map), corner r & 3 -- so that the kernel also
// emits AvgPool2d(2)(y), the pooled block input of the next layer's downsample path ([U] clip/model.py Bottleneck.downsample:
// AvgPool2d(stride) in front of the 1x1 conv), as 8 contiguous pooled rows per tile instead of a separate pass over y
// (avgpool2_kernel: 98 MB read per 128 frames).  A 1x1 conv is per pixel: the order of the rows changes no value; the pooled
// values are the mean of the ROUNDED bf16 outputs, summed ((c0 + c1) + c2) + c3 as avgpool2_kernel does.  A window index is
// wave-uniform per load / store instruction, so it is decoded on the scalar unit.
template <int K, int N, bool RES, bool RELU, bool PL = false>
__global__ __launch_bounds__(256, 1) void conv1x1_regw_kernel(RegwArgs p) {
    static_assert(!PL || (K == 128 && N == 512 && RES), "pooled variant: layer 2's last conv3");
    constexpr int NPW = N / 4, FJ = NPW / 32, KS = K / 16;
    constexpr int AP = K * 2 + 16;            // operand-tile row pitch (K*2 B is a multiple of 256 B: +16 staggers banks)
    constexpr int OPW = NPW * 2 + 16;         // staging pitch of the wave's output slice
    constexpr int CPR = K / 8;                // 16-B chunks per operand row
    constexpr int AL = 8 * CPR / 64;          // operand loads per lane (8 rows per wave)
    constexpr int OC = NPW / 8;               // 16-B chunks per output-slice row
    constexpr int OL = PX * OC / 64;          // coalesced stores / residual loads per lane
    static_assert(FJ >= 1 && FJ * KS * 4 <= 256, "weight slice must fit 256 VGPRs");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* atile = sm;                                   // [2][32][AP]
    float* sB = reinterpret_cast<float*>(sm + 2 * PX * AP);      // [N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* stg = sm + 2 * PX * AP + N * 4 + wave * (PX * OPW);
    const int px = lane & 31, h = lane >> 5;
    // wider layers are cut into channel groups of N (grid y): a workgroup owns ONE group for all of its pixel tiles
    const int c0 = blockIdx.y * N;
    const uint16_t* wg = p.w + (long)c0 * K;
    const uint16_t* resg = RES ? p.res + c0 : nullptr;
    uint16_t* yg = p.y + c0;
    const int ld = p.ldy;
    for (int i = tid; i < N; i += 256) sB[i] = p.b ? p.b[c0 + i] : 0.f;

    u32x4 wf[FJ][KS];
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((wf[I / KS][I % KS] = *reinterpret_cast<const u32x4*>(wg + (long)(NPW * wave + 32 * (I / KS) + px) * K + 16 * (I % KS) + 8 * h)), ...);
    }(std::make_integer_sequence<int, FJ * KS>{});

    int t = blockIdx.x;
    if (t >= p.ntiles) return;
    const int G = gridDim.x;
    u32x4 an[AL], rn[RES ? OL : 1];
    // PL: first pixel (corner 0) of pooling window q, on the scalar unit; a lane's corner is lane >> 4 in every row-chunk
    // enumeration below (16 chunks per row: row = 4 I + (lane >> 4), i.e. window I of the tile / of the wave's 8 operand rows)
    const int Wq = PL ? p.W / 2 : 1, QPF = PL ? (p.H / 2) * Wq : 1;
    const int coff = PL ? ((lane >> 5) & 1) * p.W + ((lane >> 4) & 1) : 0;
    auto qpix = [&](int q_) -> long {
        const int q = __builtin_amdgcn_readfirstlane(q_);
        const int f = q / QPF, rem = q - f * QPF, qr = rem / Wq, qc = rem - qr * Wq;
        return ((long)f * p.H + 2 * qr) * p.W + 2 * qc;
    };
    long opix[PL ? OL : 1];   // PL: this lane's pixel per output / residual row-chunk of the tile in the prefetch registers
    auto prefetch = [&](int tile) {
        const long m0 = (long)tile * PX;
        if constexpr (PL) {
            static_assert(!PL || (CPR == 16 && OC == 16 && AL == 2 && OL == 8), "row = 4 I + (lane >> 4)");
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((an[I] = *reinterpret_cast<const u32x4*>(p.a + (qpix(8 * tile + 2 * wave + I) + coff) * K + (lane & 15) * 8)), ...);
            }(std::make_integer_sequence<int, AL>{});
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((opix[I] = qpix(8 * tile + I) + coff,
                  rn[I] = *reinterpret_cast<const u32x4*>(resg + opix[I] * ld + NPW * wave + (lane & 15) * 8)), ...);
            }(std::make_integer_sequence<int, OL>{});
            return;
        }
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((an[I] = *reinterpret_cast<const u32x4*>(p.a + (m0 + 8 * wave + (I * 64 + lane) / CPR) * K + ((I * 64 + lane) % CPR) * 8)), ...);
        }(std::make_integer_sequence<int, AL>{});
        if constexpr (RES) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((rn[I] = *reinterpret_cast<const u32x4*>(resg + (m0 + (I * 64 + lane) / OC) * ld + NPW * wave + ((I * 64 + lane) % OC) * 8)), ...);
            }(std::make_integer_sequence<int, OL>{});
        }
    };
    auto publish_a = [&](int buf) {
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((*reinterpret_cast<u32x4*>(atile + buf * (PX * AP) + (8 * wave + (I * 64 + lane) / CPR) * AP + ((I * 64 + lane) % CPR) * 16) = an[I]), ...);
        }(std::make_integer_sequence<int, AL>{});
    };
    prefetch(t);
    publish_a(0);
    __syncthreads();

    for (int it = 0;; ++it) {
        const long m0 = (long)t * PX;
        const unsigned char* at = atile + (it & 1) * (PX * AP);
        u32x4 rc[RES ? OL : 1];
        long opc[PL ? OL : 1];
        if constexpr (RES) {
            [&]<int... I>(std::integer_sequence<int, I...>) { ((rc[I] = rn[I]), ...); }(std::make_integer_sequence<int, OL>{});
        }
        if constexpr (PL) {
            [&]<int... I>(std::integer_sequence<int, I...>) { ((opc[I] = opix[I]), ...); }(std::make_integer_sequence<int, OL>{});
        }
        const int tn = t + G;
        const bool more = tn < p.ntiles;
        if (more) prefetch(tn);

        f32x16_t acc[FJ];
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        {   // pixel-operand fragments as a software-pipelined LDS stream (see lds_stream_mfma): K-step I = 16 channels
            const unsigned a_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) const unsigned char*)at +
                                   (unsigned)(px * AP + 16 * h);
            lds_stream_mfma<KS, (KS < 6 ? KS : 6)>(
                [&](auto ic) { return std::pair<unsigned, std::integral_constant<int, 32 * decltype(ic)::value>>{a_lds, {}}; },
                [&](auto ic, const u32x4& avr) {
                    constexpr int I = decltype(ic)::value;
                    const bf16x8_t av = __builtin_bit_cast(bf16x8_t, avr);
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[j][I]), av, acc[j], 0, 0, 0);
                });
        }

        if constexpr (RES) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((*reinterpret_cast<u32x4*>(stg + ((I * 64 + lane) / OC) * OPW + ((I * 64 + lane) % OC) * 16) = rc[I]), ...);
            }(std::make_integer_sequence<int, OL>{});
        }
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 32 * j + 8 * g + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>(sB + NPW * wave + lc);
                uint2* slot = reinterpret_cast<uint2*>(stg + px * OPW + lc * 2);
                float v0 = acc[j][4 * g + 0] + bv.x, v1 = acc[j][4 * g + 1] + bv.y;
                float v2 = acc[j][4 * g + 2] + bv.z, v3 = acc[j][4 * g + 3] + bv.w;
                if constexpr (RES) {
                    const uint2 rr = *slot;
                    v0 += ec_lo(rr.x); v1 += ec_hi(rr.x); v2 += ec_lo(rr.y); v3 += ec_hi(rr.y);
                }
                if constexpr (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                uint2 o;
                o.x = ec_pack2(v0, v1);
                o.y = ec_pack2(v2, v3);
                *slot = o;
            }
#pragma unroll
        for (int i = 0; i < OL; ++i) {
            const int idx = i * 64 + lane;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (idx / OC) * OPW + (idx % OC) * 16);
            if constexpr (PL) *reinterpret_cast<u32x4*>(yg + opc[i] * ld + NPW * wave + (idx % OC) * 8) = v;
            else
            *reinterpret_cast<u32x4*>(yg + (m0 + idx / OC) * ld + NPW * wave + (idx % OC) * 8) = v;
        }
        if constexpr (PL) {   // window w = tile rows 4 w .. 4 w + 3; a lane takes two (window, 8-channel chunk) items of the wave's slice
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2) {
                const int item = it2 * 64 + lane, w = item >> 4, c = item & 15;
                const uint4 r0 = *reinterpret_cast<const uint4*>(stg + (4 * w + 0) * OPW + c * 16);
                const uint4 r1 = *reinterpret_cast<const uint4*>(stg + (4 * w + 1) * OPW + c * 16);
                const uint4 r2 = *reinterpret_cast<const uint4*>(stg + (4 * w + 2) * OPW + c * 16);
                const uint4 r3 = *reinterpret_cast<const uint4*>(stg + (4 * w + 3) * OPW + c * 16);
                auto avg2 = [](unsigned a, unsigned b, unsigned cc, unsigned d) {
                    const float lo = 0.25f * (ec_lo(a) + ec_lo(b) + ec_lo(cc) + ec_lo(d));
                    const float hi = 0.25f * (ec_hi(a) + ec_hi(b) + ec_hi(cc) + ec_hi(d));
                    return ec_pack2(lo, hi);
                };
                u32x4 po;
                po[0] = avg2(r0.x, r1.x, r2.x, r3.x); po[1] = avg2(r0.y, r1.y, r2.y, r3.y);
                po[2] = avg2(r0.z, r1.z, r2.z, r3.z); po[3] = avg2(r0.w, r1.w, r2.w, r3.w);
                *reinterpret_cast<u32x4*>(p.yp + ((long)t * 8 + w) * p.ldp + c0 + NPW * wave + c * 8) = po;
            }
        }
        if (!more) break;
        publish_a((it + 1) & 1);     // last readers of that buffer: the GEMM of tile it-1, before the previous barrier
        __syncthreads();
        t = tn;
    }
}

template <int K, int N, bool RES, bool RELU, bool PL = false>
int launch_regw(const RegwArgs& p, hipStream_t s, int groups = 1) {
    constexpr size_t lds = 2 * (size_t)PX * (K * 2 + 16) + N * 4 + 4 * (size_t)PX * (N / 4 * 2 + 16);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv1x1_regw_kernel<K, N, RES, RELU, PL>;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int wgs = 256 / groups;
    if (wgs > p.ntiles) wgs = p.ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs, (unsigned)groups), dim3(256), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

}  // namespace

extern "C" int ec_conv1x1_pair_bf16(const void* a0, const void* w0, const float* b0, const void* a1, const void* w1,
                                    const float* b1, const void* res, void* y, const void* w2, const float* b2, void* z,
                                    long M, int K0, int N, int N2, ec_stream_t stream) {
    if (!a0 || !w0 || !b0 || !y || !w2 || !b2 || !z) return EC_ERR_ARG;
    if ((a1 != nullptr) != (w1 != nullptr) || (a1 != nullptr) != (b1 != nullptr)) return EC_ERR_ARG;
    if (M <= 0 || (M % PX) != 0 || M / PX > 0x7fffffffL) return EC_ERR_SHAPE;
    if (K0 == K2 && N == NY2 && N2 == NZ2) {            // layer-2 geometry: weights in registers (residual form only)
        if (a1 || !res) return EC_ERR_SHAPE;
        Pair2Args q{(const uint16_t*)a0, (const uint16_t*)w0, (const uint16_t*)w2, (const uint16_t*)res, b0, b2, (uint16_t*)y,
                    (uint16_t*)z, (int)(M / PX)};
        return launch_pair512(q, (hipStream_t)stream);
    }
    if (K0 != KA || N != NY || (N2 != 64 && N2 != 128)) return EC_ERR_SHAPE;
    PairArgs p{(const uint16_t*)a0, (const uint16_t*)a1, (const uint16_t*)w0, (const uint16_t*)w1, (const uint16_t*)w2,
               (const uint16_t*)res, b0, b1, b2, (uint16_t*)y, (uint16_t*)z, (int)(M / PX), nullptr, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const bool two = a1 != nullptr, hres = res != nullptr;
    if (N2 == 64) {
        if (two && hres) return launch_pair<true, true, 64>(p, s);
        if (two) return launch_pair<true, false, 64>(p, s);
        if (hres) return launch_pair<false, true, 64>(p, s);
        return launch_pair<false, false, 64>(p, s);
    }
    if (two) return EC_ERR_SHAPE;           // w0|w1 (64 KB) + w2 (64 KB) + staging exceeds the 160-KB LDS
    if (hres) return launch_pair<false, true, 128>(p, s);
    return launch_pair<false, false, 128>(p, s);
}

extern "C" int ec_conv1x1_pair_pool_bf16(const void* a0, const void* w0, const float* b0, const void* res, void* y,
                                         void* y_pooled, const void* w2, const float* b2, void* z, int B, int H, int W,
                                         int K0, int N, int N2, ec_stream_t stream) {
    if (!a0 || !w0 || !b0 || !res || !y_pooled || !w2 || !b2 || !z) return EC_ERR_ARG;   // y may be NULL: not stored
    if (K0 != KA || N != NY || N2 != 128 || B <= 0 || H <= 0 || W <= 0 || (H % 4) != 0 || (W % 8) != 0) return EC_ERR_SHAPE;
    const long tiles = (long)B * (H / 4) * (W / 8);
    if (tiles > 0x7fffffffL) return EC_ERR_SHAPE;
    PairArgs p{(const uint16_t*)a0, nullptr, (const uint16_t*)w0, nullptr, (const uint16_t*)w2, (const uint16_t*)res, b0,
               nullptr, b2, (uint16_t*)y, (uint16_t*)z, (int)tiles, (uint16_t*)y_pooled, H, W};
    return launch_pair<false, true, 128, true>(p, (hipStream_t)stream);
}

// Register-weight 1x1 conv for the shapes listed at conv1x1_regw_kernel; EC_ERR_SHAPE = not handled (the caller then
// uses conv_igemm).  Only worth it when the launch has enough 32-pixel tiles to keep 256 workgroups busy.
int ec_conv1x1_regw(const void* a, const void* w, const float* bias, const void* res, void* y, long M, int K, int N, int act,
                    hipStream_t s) {
    const bool on = ec_config().conv_regw != 0;
    if (!on || (M % PX) != 0 || M / PX > 0x7fffffffL) return EC_ERR_SHAPE;
    const long tiles = M / PX;
    RegwArgs p{(const uint16_t*)a, (const uint16_t*)w, (const uint16_t*)res, bias, (uint16_t*)y, (int)tiles, N, nullptr, 0, 0, 0};
    if (tiles >= 1024) {
        if (K == 256 && N == 512 && !res && act == EC_ACT_NONE) return launch_regw<256, 512, false, false>(p, s);
        if (K == 512 && N == 256 && !res && act == EC_ACT_RELU) return launch_regw<512, 256, false, true>(p, s);
        if (K == 128 && N == 512 && res && act == EC_ACT_RELU) return launch_regw<128, 512, true, true>(p, s);
    }
    return EC_ERR_SHAPE;
}

// ... and layer 2's last conv3 (128 -> 512 + identity + ReLU @28x28) emitting AvgPool2d(2) of its output as well (PL above):
// y_pooled + q * ld_pooled = pooled pixel q.  EC_ERR_SHAPE = not this shape / too few tiles / windows not a multiple of 8:
// the caller runs the conv and the pooling pass separately.
int ec_conv1x1_regw_pool(const void* a, const void* w, const float* bias, const void* res, void* y, void* y_pooled, int B, int H,
                         int W, int K, int N, int act, int ld_pooled, hipStream_t s) {
    if (!a || !w || !bias || !res || !y || !y_pooled) return EC_ERR_ARG;
    if (!ec_config().conv_regw || !ec_config().rn50_poolout) return EC_ERR_SHAPE;
    if (K != 128 || N != 512 || act != EC_ACT_RELU || B <= 0 || (H & 1) || (W & 1) || ld_pooled < N || (ld_pooled & 7) ||
        ((size_t)y_pooled & 15))
        return EC_ERR_SHAPE;
    const long windows = (long)B * (H / 2) * (W / 2);
    if ((windows % 8) != 0 || windows / 8 < 1024 || windows / 8 > 0x7fffffffL) return EC_ERR_SHAPE;
    RegwArgs p{(const uint16_t*)a, (const uint16_t*)w, (const uint16_t*)res, bias, (uint16_t*)y, (int)(windows / 8), N,
               (uint16_t*)y_pooled, ld_pooled, H, W};
    return launch_regw<128, 512, true, true, true>(p, s);
}
