// 3x3 convolution (pad 1, stride 1) + folded BN + ReLU [+ fused AvgPool2d(2)] for the NARROW early layers of
// CLIP-RN50 (Cin, Cout in {32, 64}: stem conv2 / conv3, layer-1 conv2), bf16 NHWC, gfx950 MFMA.
//
// Replaces the cuDNN conv + BatchNorm(eval) + ReLU (+ AvgPool2d) of [U] openai/CLIP clip/model.py
// ModifiedResNet.stem / Bottleneck.conv2 reached from primitive_probing/generate_data/thor_image_features.py:109.
//
// Why a second kernel: these layers have huge M (B*112*112 or B*56*56 rows), tiny N and K <= 576.  In the tiled
// `conv_igemm` kernel every workgroup re-stages the same <= 72 KB of weights per tile and pays two barriers per
// 64-wide K-tile for 4.5-9 K-tiles of work.  Here
//   * the whole weight tensor is staged into LDS ONCE per (persistent) workgroup, per-K-step rows of 32 B with an
//     XOR swizzle that makes the 16-lane ds_read_b128 groups conflict-free;
//   * each wave owns a 32-pixel tile and fetches its im2col operand straight from global memory in MFMA operand
//     layout (lane = pixel x 8-channel half): no LDS staging of activations, no barriers; padding taps point at a
//     zero page.  The 9x re-read of every input pixel is served by the L1/L2;
//   * 16 waves per CU (4 per SIMD, <= 128 VGPRs): one wave's loads fly under the others' MFMAs;
//   * swapped MFMA operands (D[channel][pixel]) -> 8-byte packed epilogue, 2x2 pooling as two DPP quad adds
//     (rows are ordered m = 4 q + (dy*2+dx) in POOL mode), output through a per-wave LDS image as 16-B row chunks.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int PX = 32;
__device__ u32x4 ec_zero_page3[4];   // 64 zero bytes: source of padding taps

struct N3Args {
    const uint16_t *in, *w;
    const float* bias;
    uint16_t* out;
    int H, W, ntiles;
    unsigned in_bytes = 0;   // extent of the input tensor (buffer-descriptor loads of conv3x3_rowsN_kernel)
    int dbg = 0;      // profiling only (EC_ROWS_DBG): 1 no global fetch, 2 no global stores, 4 no MFMA stream, 8 no epilogue staging, 16 no footprint writes, 32 no store pass
};

__device__ __forceinline__ float dppq_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dppq_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
// ReLU as ONE instruction: v_med3_f32(x, 0, +inf).  (fmaxf costs a canonicalising v_max first; an inline-asm
// v_max_f32 is NOT safe here -- the hazard recogniser does not see that it reads an MFMA result.)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
// 16-B unit of weight row n, half hh inside a K-step block of COUT x 32 B
__device__ __forceinline__ int wunit(int n, int hh) { return ((2 * n + hh) ^ ((n >> 3) & 1)) << 4; }

template <int CIN, int COUT, bool POOL, int NW>
__global__ __launch_bounds__(NW * 64, 1) void conv3x3_narrow_kernel(N3Args p) {
    constexpr int KSTEPS = 9 * CIN / 16;        // 16 input channels of one tap per MFMA K-step
    constexpr int CB = CIN / 16;                // K-steps per tap
    constexpr int FN = COUT / 32;
    constexpr int K = 9 * CIN;
    constexpr int W_BYTES = KSTEPS * COUT * 32;
    constexpr int OP = COUT * 2 + 16;           // staging pitch
    constexpr int STG = (POOL ? PX / 4 : PX) * OP;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* sB = reinterpret_cast<float*>(sm + W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* stg = sm + W_BYTES + COUT * 4 + wave * STG;
    const int px = lane & 31, h = lane >> 5;

    // ---- weights -> LDS once: w[n][k], k = (tap, ci);  K-step ks = k / 16, half = (k / 8) & 1 ----
    ec_stage_all<COUT * (K / 8), NW * 64, u32x4>(
        tid,
        [&](int idx) { return *reinterpret_cast<const u32x4*>(p.w + (long)(idx / (K / 8)) * K + (idx % (K / 8)) * 8); },
        [&](int idx, const u32x4& v) {
            const int n = idx / (K / 8), c = idx % (K / 8);
            *reinterpret_cast<u32x4*>(sm + (c >> 1) * (COUT * 32) + wunit(n, c & 1)) = v;
        });
    for (int i = tid; i < COUT; i += NW * 64) sB[i] = p.bias[i];
    __syncthreads();

    const int GW = gridDim.x * NW;
    const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(ec_zero_page3);
    const int HW = p.H * p.W;

    // XCD-aware: each XCD (own L2) walks a contiguous run of logical blocks, so the halo rows shared by neighbouring
    // bands of pixels are fetched into ONE L2 instead of eight
    const int lb = (int)ec_xcd_remap(blockIdx.x, gridDim.x);
    for (int t = lb * NW + wave; t < p.ntiles; t += GW) {
        // ---- this lane's pixel and its 9-bit tap validity mask ----
        const int m = t * PX + px;
        int y, x, pix;
        if (POOL) {
            const int q = m >> 2, s2 = m & 3;
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            const int b = q / (Hp * Wp);
            const int r2 = q - b * (Hp * Wp);
            const int yp = r2 / Wp;
            y = 2 * yp + (s2 >> 1);
            x = 2 * (r2 - yp * Wp) + (s2 & 1);
            pix = (b * p.H + y) * p.W + x;
        } else {
            const int b = m / HW;
            const int r2 = m - b * HW;
            y = r2 / p.W;
            x = r2 - y * p.W;
            pix = m;
        }
        const unsigned xm = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);
        const unsigned msk = (y > 0 ? xm : 0u) | (xm << 3) | (y < p.H - 1 ? (xm << 6) : 0u);
        const unsigned char* base = in_b + ((long)pix * CIN + h * 8) * 2;

        // ---- im2col operand straight into MFMA layout: K-step (tap, cb) = 16 B of pixel (y+dy, x+dx) ----
        // K is walked in phases of <= 18 K-steps (72 VGPRs of operands) so that 4 waves per SIMD stay resident;
        // within a phase every load is issued before the first MFMA waits on one.
        f32x16_t acc[FN];
#pragma unroll
        for (int n = 0; n < FN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        constexpr int PH = (KSTEPS + 17) / 18, PS = KSTEPS / PH;
        static_assert(PS * PH == KSTEPS, "phase split");
        auto phase = [&](auto phc) {
            constexpr int S0 = decltype(phc)::value * PS;
            u32x4 a[PS];
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((a[I] = *reinterpret_cast<const u32x4*>(
                      ((msk >> ((S0 + I) / CB)) & 1u)
                          ? base + (((((S0 + I) / CB) / 3 - 1) * p.W + (((S0 + I) / CB) % 3 - 1)) * CIN + ((S0 + I) % CB) * 16) * 2
                          : zp + h * 16)),
                 ...);
            }(std::make_integer_sequence<int, PS>{});
            __builtin_amdgcn_sched_barrier(0);
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     const bf16x8_t aop = __builtin_bit_cast(bf16x8_t, a[I]);
#pragma unroll
                     for (int n = 0; n < FN; ++n) {
                         const s16x8_t wf = *reinterpret_cast<const s16x8_t*>(sm + (S0 + I) * (COUT * 32) + wunit(32 * n + px, h));
                         acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), aop, acc[n], 0, 0, 0);
                     }
                 }()),
                 ...);
            }(std::make_integer_sequence<int, PS>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... P>(std::integer_sequence<int, P...>) { (phase(std::integral_constant<int, P>{}), ...); }
        (std::make_integer_sequence<int, PH>{});

        // ---- epilogue: bias + ReLU (+ 2x2 mean over the lane quad) -> bf16 -> staging -> 16-B row chunks ----
#pragma unroll
        for (int n = 0; n < FN; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 32 * n + 8 * g + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>(sB + lc);
                float v0 = fmaxf(acc[n][4 * g + 0] + bv.x, 0.f), v1 = fmaxf(acc[n][4 * g + 1] + bv.y, 0.f);
                float v2 = fmaxf(acc[n][4 * g + 2] + bv.z, 0.f), v3 = fmaxf(acc[n][4 * g + 3] + bv.w, 0.f);
                if (POOL) {
                    v0 += dppq_xor1(v0); v1 += dppq_xor1(v1); v2 += dppq_xor1(v2); v3 += dppq_xor1(v3);
                    v0 += dppq_xor2(v0); v1 += dppq_xor2(v1); v2 += dppq_xor2(v2); v3 += dppq_xor2(v3);
                    if ((lane & 3) == 0) {
                        uint2 o;
                        o.x = ec_pack2(0.25f * v0, 0.25f * v1);
                        o.y = ec_pack2(0.25f * v2, 0.25f * v3);
                        *reinterpret_cast<uint2*>(stg + (px >> 2) * OP + lc * 2) = o;
                    }
                } else {
                    uint2 o;
                    o.x = ec_pack2(v0, v1);
                    o.y = ec_pack2(v2, v3);
                    *reinterpret_cast<uint2*>(stg + px * OP + lc * 2) = o;
                }
            }
        constexpr int ORows = POOL ? PX / 4 : PX;
        constexpr int ZC = COUT / 8;                          // 16-B chunks per output row
        const long orow0 = (long)t * ORows;
#pragma unroll
        for (int i = 0; i < (ORows * ZC + 63) / 64; ++i) {
            const int idx = i * 64 + lane;
            if (ORows * ZC >= 64 || idx < ORows * ZC) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (idx / ZC) * OP + (idx % ZC) * 16);
                *reinterpret_cast<u32x4*>(p.out + (orow0 + idx / ZC) * COUT + (idx % ZC) * 8) = v;
            }
        }
    }
}


// -----------------------------------------------------------------------------------------------------------------
// Row-tile variant: the activations of a tile go through a WAVE-PRIVATE LDS image instead of being gathered per
// K-step from global memory.  A wave owns 28 pixels of one image row (POOL: a 2 x 14 block = 7 pooling windows), its
// input footprint is 3 rows x 30 pixels (4 x 16), i.e. 3-4 contiguous runs that are fetched with coalesced 16-B
// loads (6-12 load instructions per tile instead of 18-36 gathers that each touch 16-32 cache lines -- the L1-tag
// bound of the kernel above), prefetched one tile ahead into registers.  The 9 taps then read the image with
// ds_read_b128 at compile-time offsets; pixel pitches of 80 B / 144 B (+128 B row skew in POOL mode) make every
// 16-lane read group conflict-free.  28 of 32 MFMA rows are real (12.5 % padding): these layers are not MFMA-bound.
// -----------------------------------------------------------------------------------------------------------------
// (Keeping the weight fragments in registers instead of LDS -- 72-144 VGPRs for 32 input channels -- was measured:
// no gain; the kernel is VALU-issue bound, not LDS bound.)
template <int CIN, int COUT, bool POOL, int NW>
__global__ __launch_bounds__(NW * 64, 1) void conv3x3_rows_kernel(N3Args p) {
    constexpr int KSTEPS = 9 * CIN / 16, CB = CIN / 16, FN = COUT / 32, K = 9 * CIN;
    constexpr int CPP = CIN / 8;                              // 16-B chunks per pixel
    constexpr int PITCH = CIN * 2 + 16;                       // bytes per pixel in the LDS image
    constexpr int RW = POOL ? 16 : 30;                        // pixels per image row of the footprint
    constexpr int RPB = POOL ? RW * PITCH + 128 : RW * PITCH; // row pitch in bytes
    constexpr int NR = POOL ? 4 : 3;
    constexpr int REGION = NR * RPB + 4 * PITCH;              // + overrun of the 4 padding lanes
    constexpr int NCH = NR * RW * CPP;                        // 16-B chunks of the footprint
    constexpr int IT = (NCH + 63) / 64;
    constexpr int W_BYTES = KSTEPS * COUT * 32;
    constexpr int OP = COUT * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* sB = reinterpret_cast<float*>(sm + W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* img = sm + W_BYTES + COUT * 4 + wave * REGION;
    const int px = lane & 31, h = lane >> 5;

    ec_stage_all<COUT * (K / 8), NW * 64, u32x4>(
        tid,
        [&](int idx) { return *reinterpret_cast<const u32x4*>(p.w + (long)(idx / (K / 8)) * K + (idx % (K / 8)) * 8); },
        [&](int idx, const u32x4& v) {
            const int n = idx / (K / 8), c = idx % (K / 8);
            *reinterpret_cast<u32x4*>(sm + (c >> 1) * (COUT * 32) + wunit(n, c & 1)) = v;
        });
    for (int i = tid; i < COUT; i += NW * 64) sB[i] = p.bias[i];
    __syncthreads();

    const int GW = gridDim.x * NW;
    const int lb = (int)ec_xcd_remap(blockIdx.x, gridDim.x);
    const int nseg = POOL ? p.W / 14 : p.W / 28;
    const int nrow = POOL ? p.H / 2 : p.H;

    // per-lane footprint geometry (tile-invariant): row / pixel / sub-chunk of the chunks this lane fetches
    int f_r[IT], f_px[IT], f_lds[IT], f_goff[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int g = i * 64 + lane;
        const int r = g / (RW * CPP), rem = g - r * (RW * CPP);
        f_r[i] = (g < NCH) ? r : -100000;                     // invalid chunk: row test fails below
        f_px[i] = rem / CPP;
        f_lds[i] = r * RPB + (rem / CPP) * PITCH + (rem % CPP) * 16;
        f_goff[i] = ((r * p.W + rem / CPP) * CIN + (rem % CPP) * 8) * 2;   // bytes from the footprint's first pixel
    }
    // this lane's MFMA-operand base in the image
    int obase;
    if (POOL) obase = ((px >> 1) & 1) * RPB + (2 * (px >> 2) + (px & 1)) * PITCH + h * 16;
    else obase = px * PITCH + h * 16;

    // this lane's 4 x 4 x FN output channels never change: keep their biases in registers and START the accumulators
    // from them (one v_mov instead of a zero + an LDS read + an add per value)
    f32x16_t bias_acc[FN];
#pragma unroll
    for (int n = 0; n < FN; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(sB + 32 * n + 8 * g + 4 * h);
            bias_acc[n][4 * g + 0] = bv.x; bias_acc[n][4 * g + 1] = bv.y; bias_acc[n][4 * g + 2] = bv.z; bias_acc[n][4 * g + 3] = bv.w;
        }

    u32x4 nxt[IT];
    const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(ec_zero_page3);
    // The tile index is wave-uniform: decode it on the scalar unit (readfirstlane) so that the per-lane work of a
    // fetch is one add + two unsigned range checks per 16-B chunk (these layers are VALU-issue bound).
    // tile coordinates (segment, row, frame) are carried incrementally: t advances by a constant stride, so the three
    // integer divisions per tile become a few scalar adds and compares
    struct Coord { int sg, rr, b; };
    auto decode = [&](int t_) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        return Coord{t % nseg, (t / nseg) % nrow, t / (nseg * nrow)};
    };
    const Coord step = decode(GW);
    auto advance = [&](Coord c) {
        c.sg += step.sg; c.rr += step.rr; c.b += step.b;
        if (c.sg >= nseg) { c.sg -= nseg; ++c.rr; }
        if (c.rr >= nrow) { c.rr -= nrow; ++c.b; }
        return c;
    };
    auto fetch = [&](Coord c) {
        const int sgi = __builtin_amdgcn_readfirstlane(c.sg), rr = __builtin_amdgcn_readfirstlane(c.rr);
        const int b = __builtin_amdgcn_readfirstlane(c.b);
        const int y0 = POOL ? 2 * rr - 1 : rr - 1, x0 = (POOL ? 14 : 28) * sgi - 1;
        const unsigned char* base = in_b + (((long)b * p.H + y0) * p.W + x0) * (CIN * 2);     // scalar (may point before
                                                                                              // the tensor: masked)
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 const bool ok = (unsigned)(y0 + f_r[I]) < (unsigned)p.H && (unsigned)(x0 + f_px[I]) < (unsigned)p.W;
                 nxt[I] = *reinterpret_cast<const u32x4*>(ok ? base + f_goff[I] : zp);   // padding reads a zero page
             }()),
             ...);
        }(std::make_integer_sequence<int, IT>{});
    };

    int t = lb * NW + wave;
    if (t >= p.ntiles) return;
    Coord cn = decode(t);
    fetch(cn);
    for (;;) {
        // registers -> wave-private image (in-order LDS: earlier reads of the previous tile are already done)
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 if (I * 64 + lane < NCH) *reinterpret_cast<u32x4*>(img + f_lds[I]) = nxt[I];
             }()),
             ...);
        }(std::make_integer_sequence<int, IT>{});
        const Coord cc = cn;
        t += GW;
        const bool more = t < p.ntiles;
        if (more) { cn = advance(cn); fetch(cn); }

        f32x16_t acc[FN];
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (([&] {
                 constexpr int tap = S / CB, cb = S % CB, ky = tap / 3, kx = tap % 3;
                 const s16x8_t av = *reinterpret_cast<const s16x8_t*>(img + obase + ky * RPB + kx * PITCH + cb * 32);
#pragma unroll
                 for (int n = 0; n < FN; ++n) {
                     const s16x8_t wf = *reinterpret_cast<const s16x8_t*>(sm + S * (COUT * 32) + wunit(32 * n + px, h));
                     // the first K-step takes the bias registers as its C operand: no accumulator init at all
                     acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                                      __builtin_bit_cast(bf16x8_t, av),
                                                                      S == 0 ? bias_acc[n] : acc[n], 0, 0, 0);
                 }
             }()),
             ...);
        }(std::make_integer_sequence<int, KSTEPS>{});

        // epilogue through the (now free) image region
        unsigned char* stg = img;
#pragma unroll
        for (int n = 0; n < FN; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 32 * n + 8 * g + 4 * h;
                float v0 = relu1(acc[n][4 * g + 0]), v1 = relu1(acc[n][4 * g + 1]);
                float v2 = relu1(acc[n][4 * g + 2]), v3 = relu1(acc[n][4 * g + 3]);
                if (POOL) {
                    v0 += dppq_xor1(v0); v1 += dppq_xor1(v1); v2 += dppq_xor1(v2); v3 += dppq_xor1(v3);
                    v0 += dppq_xor2(v0); v1 += dppq_xor2(v1); v2 += dppq_xor2(v2); v3 += dppq_xor2(v3);
                    if ((lane & 3) == 0) {
                        uint2 o;
                        o.x = ec_pack2(0.25f * v0, 0.25f * v1);
                        o.y = ec_pack2(0.25f * v2, 0.25f * v3);
                        *reinterpret_cast<uint2*>(stg + (px >> 2) * OP + lc * 2) = o;
                    }
                } else {
                    uint2 o;
                    o.x = ec_pack2(v0, v1);
                    o.y = ec_pack2(v2, v3);
                    *reinterpret_cast<uint2*>(stg + px * OP + lc * 2) = o;
                }
            }
        {
            const int sgi = __builtin_amdgcn_readfirstlane(cc.sg), rr = __builtin_amdgcn_readfirstlane(cc.rr);
            const int b = __builtin_amdgcn_readfirstlane(cc.b);
            constexpr int ORows = POOL ? 7 : 28, ZC = COUT / 8;
            const long opix = POOL ? ((long)b * (p.H / 2) + rr) * (p.W / 2) + 7 * sgi : ((long)b * p.H + rr) * p.W + 28 * sgi;
#pragma unroll
            for (int i = 0; i < (ORows * ZC + 63) / 64; ++i) {
                const int idx = i * 64 + lane;
                if (idx < ORows * ZC) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (idx / ZC) * OP + (idx % ZC) * 16);
                    *reinterpret_cast<u32x4*>(p.out + (opix + idx / ZC) * COUT + (idx % ZC) * 8) = v;
                }
            }
        }
        if (!more) break;
    }
}

// -----------------------------------------------------------------------------------------------------------------
// Multi-row tiles (round 3).  The row-tile kernel above reads 1.5 LDS fragments per MFMA (one im2col fragment per K-step
// plus one weight fragment per MFMA: a 28-pixel tile has ONE 32-row block, so a weight fragment is used once): at 6
// waves per CU that is ~260 B/clk of ds_read_b128 -- the LDS port, not the matrix pipe, paces it (MFMA busy 40 %).
// Here a wave owns RT consecutive image rows of a 28-pixel segment: a weight fragment feeds RT MFMAs, the (RT + 2)-row
// footprint is fetched once for RT output rows (input re-read 3x -> (RT + 2) / RT), and the fragment reads are an
// inline-asm stream LEADK K-steps ahead of their MFMAs (one wave per SIMD has no partner to hide LDS latency behind;
// cf. conv_pair.hip lds_stream_mfma).  Same K order as every other kernel of the layer (tap-major, 16 channels per
// K-step), same epilogue arithmetic: results are bit-identical to conv3x3_rows_kernel.
// -----------------------------------------------------------------------------------------------------------------
template <int R>
struct FragRingN { u32x4 f[R]; };
template <int CIN, int COUT, int RT, int NW, int LEADK>
__global__ __launch_bounds__(NW * 64, 1) void conv3x3_rowsN_kernel(N3Args p) {
    constexpr int KSTEPS = 9 * CIN / 16, CB = CIN / 16, FN = COUT / 32, K = 9 * CIN;
    constexpr int CPP = CIN / 8;                              // 16-B chunks per pixel
    constexpr int PITCH = CIN * 2 + 16;                       // bytes per pixel in the LDS image
    constexpr int RW = 30, RPB = RW * PITCH, NR = RT + 2;
    constexpr int REGION = NR * RPB + 4 * PITCH;              // + overrun of the 4 padding lanes
    constexpr int NCH = NR * RW * CPP;                        // 16-B chunks of the footprint
    constexpr int IT = (NCH + 63) / 64;
    constexpr int W_BYTES = KSTEPS * COUT * 32;
    constexpr int OP = COUT * 2 + 16;
    constexpr int G = RT + FN;                                // fragments per K-step: RT im2col blocks, FN weight blocks
    constexpr int RING = (LEADK + 1) * G;
    static_assert(RT * 32 * OP <= REGION, "epilogue image fits the footprint region");
    static_assert(G * LEADK <= 15, "lgkmcnt is a 4-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* sB = reinterpret_cast<float*>(sm + W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* img = sm + W_BYTES + COUT * 4 + wave * REGION;
    const int px = lane & 31, h = lane >> 5;

    ec_stage_all<COUT * (K / 8), NW * 64, u32x4>(
        tid,
        [&](int idx) { return *reinterpret_cast<const u32x4*>(p.w + (long)(idx / (K / 8)) * K + (idx % (K / 8)) * 8); },
        [&](int idx, const u32x4& v) {
            const int n = idx / (K / 8), c = idx % (K / 8);
            *reinterpret_cast<u32x4*>(sm + (c >> 1) * (COUT * 32) + wunit(n, c & 1)) = v;
        });
    for (int i = tid; i < COUT; i += NW * 64) sB[i] = p.bias[i];
    __syncthreads();

    const int GW = gridDim.x * NW;
    const int lb = (int)ec_xcd_remap(blockIdx.x, gridDim.x);
    const int nseg = p.W / 28, nrow = p.H / RT;

    // per-lane footprint geometry (tile-invariant).  Only the footprint's FIRST / LAST row and column can fall outside
    // the frame, and whether they do is a property of the tile (wave-uniform): a chunk carries 4 edge bits, the tile a
    // 4-bit mask, and an invalid chunk is an out-of-range descriptor offset (reads as zeros) -- 4 VALU per chunk where the
    // two range compares + 64-bit pointer select + 64-bit add took ~10 (the per-tile glue of this kernel is as long as
    // its MFMA stream: EC_ROWS_DBG ablations, DESIGN.md section 4.6 d)
    int f_lds[IT];
    unsigned f_goff[IT], f_edge[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int g = i * 64 + lane;
        const int r = g / (RW * CPP), rem = g - r * (RW * CPP), fpx = rem / CPP;
        f_edge[i] = (g >= NCH ? 16u : 0u) | (r == 0 ? 1u : 0u) | (r == NR - 1 ? 2u : 0u) | (fpx == 0 ? 4u : 0u) | (fpx == RW - 1 ? 8u : 0u);
        f_lds[i] = r * RPB + fpx * PITCH + (rem % CPP) * 16;
        f_goff[i] = (unsigned)(((r * p.W + fpx) * CIN + (rem % CPP) * 8) * 2);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
#endif
    // LDS byte addresses of this lane's operand slots: the rest of every fragment address is an instruction immediate
    const unsigned a_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)img + (unsigned)(px * PITCH + h * 16);
    const unsigned w_lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)sm + (unsigned)wunit(px, h);
    constexpr int WSEG = 32768;                               // ds_read immediates are 16 bits: weight image in 32-KB segments
    struct { unsigned v[(W_BYTES + WSEG - 1) / WSEG]; } w_lds;
#pragma unroll
    for (int q = 0; q < (W_BYTES + WSEG - 1) / WSEG; ++q) w_lds.v[q] = w_lds0 + q * WSEG;

    f32x16_t bias_acc[FN];
#pragma unroll
    for (int n = 0; n < FN; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(sB + 32 * n + 8 * g + 4 * h);
            bias_acc[n][4 * g + 0] = bv.x; bias_acc[n][4 * g + 1] = bv.y; bias_acc[n][4 * g + 2] = bv.z; bias_acc[n][4 * g + 3] = bv.w;
        }

    u32x4 nxt[IT];
    struct Coord { int sg, rr, b; };
    auto decode = [&](int t_) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        return Coord{t % nseg, (t / nseg) % nrow, t / (nseg * nrow)};
    };
    const Coord step = decode(GW);
    auto advance = [&](Coord c) {
        c.sg += step.sg; c.rr += step.rr; c.b += step.b;
        if (c.sg >= nseg) { c.sg -= nseg; ++c.rr; }
        if (c.rr >= nrow) { c.rr -= nrow; ++c.b; }
        return c;
    };
    auto fetch = [&](Coord c) {
        const int sgi = __builtin_amdgcn_readfirstlane(c.sg), rr = __builtin_amdgcn_readfirstlane(c.rr);
        const int b = __builtin_amdgcn_readfirstlane(c.b);
        const int y0 = RT * rr - 1, x0 = 28 * sgi - 1;
        const unsigned base = (unsigned)((((long)b * p.H + y0) * p.W + x0) * (CIN * 2));      // (mod 2^32: edge chunks are masked)
        const unsigned tmask = 16u | (y0 < 0 ? 1u : 0u) | (y0 + NR - 1 >= p.H ? 2u : 0u) | (x0 < 0 ? 4u : 0u) |
                               (x0 + RW - 1 >= p.W ? 8u : 0u) | ((p.dbg & 1) ? 15u : 0u);
#if defined(__HIP_DEVICE_COMPILE__)
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 const unsigned off = (f_edge[I] & tmask) ? 0xFFFFFFF0u : base + f_goff[I];
                 nxt[I] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0);
             }()),
             ...);
        }(std::make_integer_sequence<int, IT>{});
#endif
    };

    int t = lb * NW + wave;
    if (t >= p.ntiles) return;
    Coord cn = decode(t);
    fetch(cn);
    for (;;) {
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 if (I * 64 + lane < NCH && !(p.dbg & 16)) *reinterpret_cast<u32x4*>(img + f_lds[I]) = nxt[I];
             }()),
             ...);
        }(std::make_integer_sequence<int, IT>{});
        const Coord cc = cn;
        t += GW;
        const bool more = t < p.ntiles;
        if (more) { cn = advance(cn); fetch(cn); }
        asm volatile("" ::: "memory");                       // the image stores above stay above the asm fragment reads

        f32x16_t acc[RT][FN];
        FragRingN<RING> ring;
        // fragment q of K-step S: q < RT -> im2col block q (image row q + ky of the footprint), else weight block q - RT
        auto issue = [&](auto sc, auto qc) {
            constexpr int S = decltype(sc)::value, q = decltype(qc)::value;
            constexpr int tap = S / CB, cb = S % CB, ky = tap / 3, kx = tap % 3;
            constexpr int slot = (S % (LEADK + 1)) * G + q;
            const auto& wl = w_lds;      // (named outside the discarded branch: implicit capture in a generic lambda)
            const unsigned al = a_lds;
            if constexpr (q < RT) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring.f[slot]) : "v"(al), "n"((q + ky) * RPB + kx * PITCH + cb * 32));
            } else {
                constexpr int off = S * (COUT * 32) + (q - RT) * 1024;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring.f[slot]) : "v"(wl.v[off / WSEG]), "n"(off % WSEG));
            }
        };
        auto issue_step = [&](auto sc) {
            [&]<int... Q>(std::integer_sequence<int, Q...>) { (issue(sc, std::integral_constant<int, Q>{}), ...); }
            (std::make_integer_sequence<int, G>{});
        };
        if (p.dbg & 4) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int n = 0; n < FN; ++n) acc[i][n] = bias_acc[n];
        } else {
        [&]<int... S>(std::integer_sequence<int, S...>) { (issue_step(std::integral_constant<int, S>{}), ...); }
        (std::make_integer_sequence<int, (LEADK < KSTEPS ? LEADK : KSTEPS)>{});
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (([&] {
                 if constexpr (S + LEADK < KSTEPS) issue_step(std::integral_constant<int, S + LEADK>{});
                 constexpr int ahead = (S + LEADK < KSTEPS ? LEADK : KSTEPS - 1 - S) * G;   // reads issued after K-step S's
                 asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(ahead));
                 constexpr int base = (S % (LEADK + 1)) * G;
#pragma unroll
                 for (int q = 0; q < G; ++q) asm volatile("" : "+v"(ring.f[base + q]));   // ties the fragments to the wait: the MFMAs stay below it
#pragma unroll
                 for (int i = 0; i < RT; ++i)
#pragma unroll
                     for (int n = 0; n < FN; ++n)
                         acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ring.f[base + RT + n]),
                                                                            __builtin_bit_cast(bf16x8_t, ring.f[base + i]),
                                                                            S == 0 ? bias_acc[n] : acc[i][n], 0, 0, 0);
             }()),
             ...);
        }(std::make_integer_sequence<int, KSTEPS>{});
        }

        // epilogue through the (now free) image region: RT blocks of 32 pixel rows (28 real each)
        unsigned char* stg = img;
        if (!(p.dbg & 8))
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int n = 0; n < FN; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lc = 32 * n + 8 * g + 4 * h;
                    uint2 o;
                    o.x = ec_pack2(relu1(acc[i][n][4 * g + 0]), relu1(acc[i][n][4 * g + 1]));
                    o.y = ec_pack2(relu1(acc[i][n][4 * g + 2]), relu1(acc[i][n][4 * g + 3]));
                    *reinterpret_cast<uint2*>(stg + (i * 32 + px) * OP + lc * 2) = o;
                }
        {
            const int sgi = __builtin_amdgcn_readfirstlane(cc.sg), rr = __builtin_amdgcn_readfirstlane(cc.rr);
            const int b = __builtin_amdgcn_readfirstlane(cc.b);
            constexpr int ZC = COUT / 8;
            const long opix0 = ((long)b * p.H + RT * rr) * p.W + 28 * sgi;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const long opix = opix0 + (long)i * p.W;
#pragma unroll
                for (int j = 0; j < (28 * ZC + 63) / 64; ++j) {
                    const int idx = j * 64 + lane;
                    if (idx < 28 * ZC && !(p.dbg & 2) && !(p.dbg & 32)) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (i * 32 + idx / ZC) * OP + (idx % ZC) * 16);
                        *reinterpret_cast<u32x4*>(p.out + (opix + idx / ZC) * COUT + (idx % ZC) * 8) = v;
                    }
                }
            }
        }
        if (!more) break;
    }
}

template <int CIN, int COUT, int RT, int NW, int LEADK>
int launch_rowsN(const N3Args& p, hipStream_t s) {
    constexpr int PITCH = CIN * 2 + 16, RPB = 30 * PITCH;
    constexpr size_t lds = (size_t)(9 * CIN / 16) * COUT * 32 + COUT * 4 + (size_t)NW * ((RT + 2) * RPB + 4 * PITCH);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(((RT + 2) * RPB + 4 * PITCH) % 16 == 0, "image alignment");
    auto kern = conv3x3_rowsN_kernel<CIN, COUT, RT, NW, LEADK>;
    const int dbg = 0;   // (profiling field of the kernel arguments: EC_ROWS_DBG existed in rounds 3-4)
    N3Args q = p;
    q.dbg = dbg;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int wgs = (p.ntiles + NW - 1) / NW < 256 ? (p.ntiles + NW - 1) / NW : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(NW * 64), lds, s, q);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

template <int CIN, int COUT, bool POOL, int NW>
int launch_rows(const N3Args& p, hipStream_t s) {
    constexpr int PITCH = CIN * 2 + 16, RW = POOL ? 16 : 30, RPB = POOL ? RW * PITCH + 128 : RW * PITCH, NR = POOL ? 4 : 3;
    constexpr size_t lds = (size_t)(9 * CIN / 16) * COUT * 32 + COUT * 4 + (size_t)NW * (NR * RPB + 4 * PITCH);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((NR * RPB + 4 * PITCH) % 16 == 0, "image alignment");
    auto kern = conv3x3_rows_kernel<CIN, COUT, POOL, NW>;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int wgs = (p.ntiles + NW - 1) / NW < 256 ? (p.ntiles + NW - 1) / NW : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(NW * 64), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

template <int CIN, int COUT, bool POOL, int NW = 16>
int launch_n3(const N3Args& p, hipStream_t s) {
    constexpr size_t lds = (size_t)(9 * CIN / 16) * COUT * 32 + COUT * 4 + NW * (size_t)((POOL ? PX / 4 : PX) * (COUT * 2 + 16));
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_narrow_kernel<CIN, COUT, POOL, NW>;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int wgs = (p.ntiles + NW - 1) / NW < 256 ? (p.ntiles + NW - 1) / NW : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(NW * 64), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

}  // namespace

// Returns EC_OK when it ran, EC_ERR_SHAPE when the shape is not one it handles (the caller then uses conv_igemm).
int ec_conv3x3_narrow(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                      int pool, hipStream_t s) {
    // EC_CONV_NARROW: 0 off; 1 gather kernel only; 2 gather kernel also for Cin = 64; 3 (default) row-tile kernel
    // where the image width allows (W % 28 == 0, POOL: W % 14 == 0), gather kernel for the other 32-channel layers
    const int mode = ec_config().conv_narrow;
    const long M = (long)B * H * W;
    if (!mode || !bias || (pool && ((H | W) & 1))) return EC_ERR_SHAPE;
    if (mode >= 3) {
        const bool fits = pool ? (W % 14 == 0) : (W % 28 == 0);
        const long nt = pool ? (long)B * (H / 2) * (W / 14) : (long)B * H * (W / 28);
        // EC_CONV_ROWSN (default 1): multi-row tiles (conv3x3_rowsN_kernel) for the un-pooled layers whose height allows
        const int rowsn = ec_config().conv_rowsn;
        if (fits && !pool && rowsn && nt <= 0x7fffffffL && M * Cin * 2 < (1L << 32) - 16) {
            if (Cin == 64 && Cout == 64 && H % 2 == 0) {
                N3Args p{(const uint16_t*)in, (const uint16_t*)w, bias, (uint16_t*)out, H, W, (int)(nt / 2)};
                p.in_bytes = (unsigned)(M * Cin * 2);
                return rowsn == 3 ? launch_rowsN<64, 64, 2, 4, 3>(p, s) : launch_rowsN<64, 64, 2, 4, 2>(p, s);
            }
            if (Cin == 32 && Cout == 32 && H % 4 == 0) {
                N3Args p{(const uint16_t*)in, (const uint16_t*)w, bias, (uint16_t*)out, H, W, (int)(nt / 4)};
                p.in_bytes = (unsigned)(M * Cin * 2);
                return rowsn == 2 ? launch_rowsN<32, 32, 4, 4, 2>(p, s) : launch_rowsN<32, 32, 4, 8, 1>(p, s);
            }
        }
        if (fits && nt <= 0x7fffffffL) {
            N3Args p{(const uint16_t*)in, (const uint16_t*)w, bias, (uint16_t*)out, H, W, (int)nt};
            if (Cin == 32 && Cout == 32 && !pool) return launch_rows<32, 32, false, 16>(p, s);
            if (Cin == 32 && Cout == 64 && pool) return launch_rows<32, 64, true, 16>(p, s);
            if (Cin == 32 && Cout == 64 && !pool) return launch_rows<32, 64, false, 16>(p, s);
            if (Cin == 64 && Cout == 64 && !pool) return launch_rows<64, 64, false, 6>(p, s);
            if (Cin == 64 && Cout == 64 && pool) return launch_rows<64, 64, true, 8>(p, s);
        }
    }
    if ((M % PX) != 0 || M / PX > 0x7fffffffL) return EC_ERR_SHAPE;
    N3Args p{(const uint16_t*)in, (const uint16_t*)w, bias, (uint16_t*)out, H, W, (int)(M / PX)};
    if (Cin == 32 && Cout == 32 && !pool) return launch_n3<32, 32, false>(p, s);
    if (Cin == 32 && Cout == 64 && pool) return launch_n3<32, 64, true>(p, s);
    if (Cin == 32 && Cout == 64 && !pool) return launch_n3<32, 64, false>(p, s);
    // Cin = 64 on the gather kernel: one 128-B cache line per pixel -> 32 lines per load instruction, L1-tag bound
    // (measured 157 us vs 106 us for conv_igemm at B = 256); kept for EC_CONV_NARROW=2 experiments only.
    if (mode == 2 && Cin == 64 && Cout == 64 && !pool) return launch_n3<64, 64, false>(p, s);
    if (mode == 2 && Cin == 64 && Cout == 64 && pool) return launch_n3<64, 64, true>(p, s);
    return EC_ERR_SHAPE;
}
