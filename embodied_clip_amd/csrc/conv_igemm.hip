// Implicit-GEMM convolution / GEMM on bf16 MFMA for gfx950 (CDNA4).
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] (+ res[m, n]) )
//
// A is never materialised: row m is an output pixel of an NHWC tensor and
// k = (ky, kx, ci) walks the 3x3 (or 1x1) window, so each 16-byte chunk of the
// K axis is 8 consecutive input channels of one tap and is fetched straight
// from the NHWC activation (zero outside the frame).  W is [Cout][K] with K
// contiguous, i.e. the "B^T" operand the MFMA B-fragment wants.
//
// Replaces: every cuDNN conv + BatchNorm(eval) + ReLU / residual / AvgPool2d
// the reference triggers through `clip_model(clip_input)`
// (primitive_probing/generate_data/thor_image_features.py:109; [U] openai/CLIP
// clip/model.py Bottleneck.forward / ModifiedResNet stem), and the
// nn.Linear / in_proj / out_proj GEMMs of ResidualAttentionBlock and
// AttentionPool2d.
//
// Structure (wave64, v_mfma_f32_32x32x16_bf16):
//   * persistent workgroups of 4 waves; tile BM x BN x 64 (128x128, or 256x64 / 256x32 for narrow Cout);
//     each wave owns (BM/WM) x (BN/WN) as FM x FN 32x32 fp32 accumulators.
//   * operands go global -> LDS directly (LDS-DMA, `global_load_lds_dwordx4`): no staging registers, no
//     ds_write pass.  The LDS image has 128-byte rows; LDS-DMA writes are lane-linear, so the XOR swizzle
//     ((row>>1)&7 on the 16-byte chunk index) is applied to the SOURCE chunk each lane fetches and again
//     on the fragment ds_read_b128 (conflict-free for the 16-lane groups {0-3,12-15,20-27},...).
//     Padding / ragged-K chunks are out-of-range buffer offsets (the hardware returns zeros), selected per row from a 9-bit tap-validity mask that is
//     computed once per tile -- the 3x3 halo costs one v_cndmask per chunk, no branches.
//   * SWAPPED MFMA operands (D[n][m]): a lane owns ONE pixel and, per 4 accumulator registers, 4
//     consecutive channels -> bias/residual/activation/bf16-rounding work on packed 8-byte LDS slots
//     (v_cvt_pk_bf16_f32), and the tile leaves as coalesced 16-byte row chunks.
//   * single 32-KB LDS stage + 35-KB epilogue image, <= 168 VGPRs: 3 workgroups per CU; the load latency
//     of one workgroup hides behind the MFMAs of the other two (two barriers per K-tile).
//
// Fused AvgPool2d(2) (CLIP's anti-aliased stride): in POOL mode row m is ordered m = 4*q + (dy*2+dx) with
// q the pooled raster index, so the four pixels of a pooling window are the four lanes of a quad; the pool
// is two DPP quad-permute adds after bias+ReLU -- the full-resolution output never touches HBM.
//
// What the round-1 measurements say (profiles/): with many tiles per launch the kernel runs
// at 760-850 TFLOP/s; the mid-size layers sit at 570-680 because of (a) the L2->LDS operand feed (3x3 256->256
// @14x14, B=256: MFMA alone 33 us, operand loads alone 42 us, together ~the sum) and (b) tile quantisation (784 tiles
// on 768 workgroups cost 98 us where 766 tiles cost 72 us).  (b) is what the 196-of-224-row tile configuration
// (template parameter MV, see dispatch_tile) removes for single 256-frame launches.  Measured without effect:
// loads two tiles ahead, double vs single LDS stage, fat 128x256 / 256x128 tiles, 8-wave workgroups,
// all-fragments-up-front MFMA scheduling, LDS-DMA pieces interleaved between MFMA groups, start skew between
// co-resident workgroups, wave specialisation, a split-K tail reduced by the last arriver, 64-row tiles.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "common.h"

thread_local int ec_tls_conv8_min_tiles = EC_CONV8_MIN_TILES_DEFAULT;   // common.h: set per call from the encoder handle

namespace {

// profiling switch of the tools-only build (`make tools`, -DEC_TOOLS); the product library never ablates
inline int ec_tools_ablate() {
#ifdef EC_TOOLS
    static const int v = [] { const char* e = getenv("EC_CONV_ABLATE"); return e ? atoi(e) : 0; }();
    return v;
#else
    return 0;
#endif
}

constexpr int BK = 64;              // K-tile (bf16 elements) = 128 B rows in LDS
constexpr int ROW_BYTES = BK * 2;   // 128

struct ConvArgs {
    const uint16_t* in;
    const uint16_t* w;
    const float* bias;
    const uint16_t* res;
    uint16_t* out;
    int H, W, Cin, Cout, K, M;
    int ldo;   // row stride of `out` in elements (== Cout for a dense tensor; > Cout writes a column block of a wider one)
    int cin_log2;
    int act;
    int ntn;   // number of N tiles
    int ntiles; // total output tiles
    int nbuf;  // LDS stages: 2 (double buffer) or 1
    unsigned in_bytes, w_bytes;   // buffer-descriptor extents (out-of-range loads return 0)
    unsigned res_bytes;           // ... of the residual tensor (= output extent)
    int ablate;                   // profiling only (tools build, EC_CONV_ABLATE): 1 no global loads, 2 no MFMA, 8 no epilogue; 0 in the product library
    // ---- LayerNorm folded into the GEMM (conv_igemm8, 1x1 only; ec_gemm_bf16_ln8) ----
    // consumer: out = act(rstd[m] * (acc[m, n] - mean[m] * ln_s[n]) + bias[n]) with acc = x . (W diag(gamma))^T on the RAW rows x,
    //           ln_s[n] = sum_k (W diag(gamma))[n, k], bias[n] = sum_k beta[k] W[n, k] + b[n]; (mean, rstd) of row m from the
    //           ln_np partial records {sum, M2, count, -} the producer of x left in ln_stats[m][ln_np]
    // producer: stats_out != nullptr: the epilogue also writes this tile's record of every output row (over its BN columns,
    //           computed from the ROUNDED bf16 values it stores) to stats_out[m][ntn][4]
    const float* ln_s = nullptr;
    const float* ln_stats = nullptr;
    int ln_np = 0;
    float ln_eps = 1e-5f;
    float* stats_out = nullptr;
    int aux_off = 0;              // byte offset of the 4-KB auxiliary LDS area behind the stages / the epilogue image
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ float dpp_quad_xor1(float v) {   // lane ^ 1 within a quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {   // lane ^ 2 within a quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// ---- epilogue staging, shared by both kernels --------------------------------------------------------------------
// One wave's accumulators -> the workgroup's LDS output image: + bias (+ residual already staged in the image) ->
// activation (or ReLU + 2x2 average pool over the lane quad) -> ONE rounding to bf16 -> 8-byte slots.  ACT / RES are
// COMPILE-TIME: the previous version tested p.act / p.res / p.bias per 4-channel group (a three-way branch with the
// whole QuickGELU body behind it, a waited bias load and a waited residual read per group: ~20k clk per 256x256 tile,
// s_memtime stamps of round 3); here the biases are fetched up front and the body is straight-line code.
__device__ __forceinline__ float ec_relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
template <int ACT, bool RES, bool POOL, int FM, int FN, int PITCH>
__device__ __forceinline__ void epi_stage(const f32x16_t (&acc)[FM][FN], unsigned char* smem, const float* bias_n0,
                                          int row0, int col0, int lane) {
    const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        // the four bias vectors of this 32-channel block up front (one exposed load latency per j) -- except in the residual
        // variants of the 3-workgroups-per-CU kernel, which have no registers for them (they spill: 72 -> 80 us on
        // layer-3 conv3) and fetch one vector per group instead
        constexpr bool BIAS_UP_FRONT = !(RES && FM * FN <= 4);
        float4 bvj[4];
        if constexpr (BIAS_UP_FRONT) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bvj[g] = bias_n0 ? *reinterpret_cast<const float4*>(bias_n0 + col0 + j * 32 + 8 * g + 4 * fhalf)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (!BIAS_UP_FRONT)
                bvj[g] = bias_n0 ? *reinterpret_cast<const float4*>(bias_n0 + col0 + j * 32 + 8 * g + 4 * fhalf)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            const int lcol = col0 + j * 32 + 8 * g + 4 * fhalf;          // 4 consecutive channels
            uint2 rr[FM];
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    rr[i] = *reinterpret_cast<const uint2*>(smem + (row0 + i * 32 + frow) * PITCH + lcol * 2);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int lrow_px = row0 + i * 32 + frow;                // this lane's pixel (tile-local row)
                float v0 = acc[i][j][4 * g + 0] + bvj[g].x, v1 = acc[i][j][4 * g + 1] + bvj[g].y;
                float v2 = acc[i][j][4 * g + 2] + bvj[g].z, v3 = acc[i][j][4 * g + 3] + bvj[g].w;
                if constexpr (POOL) {
                    // the 4 pixels of a pooling window are the 4 lanes of a quad (m = 4*q + dy*2+dx)
                    v0 = ec_relu(v0); v1 = ec_relu(v1); v2 = ec_relu(v2); v3 = ec_relu(v3);
                    v0 += dpp_quad_xor1(v0); v1 += dpp_quad_xor1(v1); v2 += dpp_quad_xor1(v2); v3 += dpp_quad_xor1(v3);
                    v0 += dpp_quad_xor2(v0); v1 += dpp_quad_xor2(v1); v2 += dpp_quad_xor2(v2); v3 += dpp_quad_xor2(v3);
                    if ((lane & 3) == 0) {
                        uint2 o;
                        o.x = ec_pack2(0.25f * v0, 0.25f * v1);
                        o.y = ec_pack2(0.25f * v2, 0.25f * v3);
                        *reinterpret_cast<uint2*>(smem + (lrow_px >> 2) * PITCH + lcol * 2) = o;
                    }
                } else {
                    if constexpr (RES) {
                        v0 += ec_lo(rr[i].x); v1 += ec_hi(rr[i].x); v2 += ec_lo(rr[i].y); v3 += ec_hi(rr[i].y);
                    }
                    if constexpr (ACT == EC_ACT_RELU) {
                        v0 = ec_relu(v0); v1 = ec_relu(v1); v2 = ec_relu(v2); v3 = ec_relu(v3);
                    } else if constexpr (ACT == EC_ACT_QUICKGELU) {
                        // x * sigmoid(1.702 x) with v_exp + v_rcp (1 ulp each; the result is rounded to bf16 next): a true
                        // IEEE division costs ~15 instructions per value here
                        v0 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v0)); v1 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v1));
                        v2 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v2)); v3 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v3));
                    }
                    uint2 o;
                    o.x = ec_pack2(v0, v1);
                    o.y = ec_pack2(v2, v3);
                    *reinterpret_cast<uint2*>(smem + lrow_px * PITCH + lcol * 2) = o;
                }
            }
        }
    }
}
// runtime (act, residual) -> the matching straight-line instance: ONE wave-uniform branch per tile
// (QuickGELU never comes with a residual: ec_conv_bf16 / ec_gemm_bf16 reject the pair.  RES_ONLY: the residual-prefetching
//  instances are only ever launched with a residual, so they carry two variants instead of five -- every variant inlined here
//  counts towards the kernel's register allocation, and those instances have none to spare.)
template <bool POOL, bool RES_ONLY, int FM, int FN, int PITCH>
__device__ __forceinline__ void epi_stage_dispatch(const f32x16_t (&acc)[FM][FN], unsigned char* smem, const float* bias_n0,
                                                   int row0, int col0, int lane, int act, bool has_res) {
    if constexpr (POOL) {
        epi_stage<EC_ACT_RELU, false, true, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);   // (pool => ReLU, no residual)
    } else if constexpr (RES_ONLY) {
        if (act == EC_ACT_RELU) epi_stage<EC_ACT_RELU, true, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
        else epi_stage<EC_ACT_NONE, true, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
    } else {
        if (act == EC_ACT_RELU) {
            if (has_res) epi_stage<EC_ACT_RELU, true, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
            else epi_stage<EC_ACT_RELU, false, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
        } else if (act == EC_ACT_QUICKGELU) {
            epi_stage<EC_ACT_QUICKGELU, false, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
        } else {
            if (has_res) epi_stage<EC_ACT_NONE, true, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
            else epi_stage<EC_ACT_NONE, false, false, FM, FN, PITCH>(acc, smem, bias_n0, row0, col0, lane);
        }
    }
}

// LayerNorm-folded epilogue (ConvArgs::ln_*): v = a_m * acc + (b_m * s[n] + c[n]) with (a_m, b_m) = (rstd, -mean * rstd) of the
// lane's row out of the LDS table `rowab` (built at kernel entry from the producer's partial records), then the activation.
template <int ACT, int FM, int FN, int PITCH>
__device__ __forceinline__ void epi_stage_ln(const f32x16_t (&acc)[FM][FN], unsigned char* smem, const float* s_n0, const float* c_n0,
                                             const float2* rowab, int row0, int col0, int lane) {
    const int frow = lane & 31, fhalf = lane >> 5;
    float2 ab[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) ab[i] = rowab[row0 + i * 32 + frow];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        float4 sv[4], cv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sv[g] = *reinterpret_cast<const float4*>(s_n0 + col0 + j * 32 + 8 * g + 4 * fhalf);
            cv[g] = *reinterpret_cast<const float4*>(c_n0 + col0 + j * 32 + 8 * g + 4 * fhalf);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int lcol = col0 + j * 32 + 8 * g + 4 * fhalf;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int lrow_px = row0 + i * 32 + frow;
                float v0 = fmaf(ab[i].x, acc[i][j][4 * g + 0], fmaf(ab[i].y, sv[g].x, cv[g].x));
                float v1 = fmaf(ab[i].x, acc[i][j][4 * g + 1], fmaf(ab[i].y, sv[g].y, cv[g].y));
                float v2 = fmaf(ab[i].x, acc[i][j][4 * g + 2], fmaf(ab[i].y, sv[g].z, cv[g].z));
                float v3 = fmaf(ab[i].x, acc[i][j][4 * g + 3], fmaf(ab[i].y, sv[g].w, cv[g].w));
                if constexpr (ACT == EC_ACT_QUICKGELU) {
                    v0 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v0)); v1 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v1));
                    v2 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v2)); v3 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v3));
                } else if constexpr (ACT == EC_ACT_RELU) {
                    v0 = ec_relu(v0); v1 = ec_relu(v1); v2 = ec_relu(v2); v3 = ec_relu(v3);
                }
                uint2 o;
                o.x = ec_pack2(v0, v1);
                o.y = ec_pack2(v2, v3);
                *reinterpret_cast<uint2*>(smem + lrow_px * PITCH + lcol * 2) = o;
            }
        }
    }
}

// Software-pipelined fragment stream (cf. conv_pair.hip lds_stream_mfma): the compiler sinks every LDS operand read to
// just before the MFMAs that use it (ds_read x4 -> s_waitcnt -> mfma x4 per k-step, the LDS latency exposed four times per
// K-tile).  Here the reads are inline asm, one k-step (G fragments) ahead of the MFMAs; the waits carry the fragment
// registers so that the MFMAs cannot move above them.  LDS returns data in order: lgkmcnt(n) = all but the newest n arrived.
template <int R>
struct FragRing { u32x4_t f[R]; };
template <int I, int R, class AddrFn>
__device__ __forceinline__ void fr_issue(FragRing<R>& ring, AddrFn& addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(ring.f[I % R]) : "v"(addr(std::integral_constant<int, I>{})));
}
template <int I, int R>
__device__ __forceinline__ void fr_tie(FragRing<R>& ring) { asm volatile("" : "+v"(ring.f[I % R])); }   // orders the consumers after the wait
template <int K, int NK, int G, int R, class AddrFn, class MmaFn>
__device__ __forceinline__ void fr_step(FragRing<R>& ring, AddrFn& addr, MmaFn& mma) {
    if constexpr (K < NK) {
        // issue the next k-step's fragments first, then wait for this k-step's (G reads may stay in flight)
        if constexpr (K + 1 < NK) {
            [&]<int... Q>(std::integer_sequence<int, Q...>) { (fr_issue<(K + 1) * G + Q, R>(ring, addr), ...); }
            (std::make_integer_sequence<int, G>{});
        }
        constexpr int after = (K + 1 < NK) ? G : 0;
        static_assert(G <= 15, "lgkmcnt is a 4-bit counter");
        asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(after));
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (fr_tie<K * G + Q, R>(ring), ...); }(std::make_integer_sequence<int, G>{});
        mma(std::integral_constant<int, K>{}, ring);
        fr_step<K + 1, NK, G, R>(ring, addr, mma);
    }
}

// MV = rows of the tile that are real output rows (tile stride in M); MV < BM pads the tile (see dispatch_tile:
// 196-of-224-row tiles make every RN50 layer's tile count a multiple of the CU count).
// NS >= 3: RING mode for launches with at most 1-2 workgroups per CU (small per-GPU batches, the 7x7 maps): NS LDS stages,
// the LDS-DMA of K-tile kt+NS-1 is issued while K-tile kt computes, each wave waits for its OWN pieces of K-tile kt with a
// COUNTED s_waitcnt vmcnt (the younger NS-2 tiles stay in flight) and one raw s_barrier per K-tile orders them for the other
// waves.  The K walk and the per-element summation order are those of the other modes (bit-identical results); what changes
// is that a K-tile no longer costs a full L2 round trip: measured (round 3, rocprofv3) a 64x64 tile of the single-stage mode
// takes ~750 ns per K-tile with one workgroup per CU -- 128 clk of MFMA issue per wave.
// ILV (ring mode only): the LDS-DMA pieces of K-tile kt + NS - 1 are issued INSIDE the MFMA stream of K-tile kt (a share of
// them after every k-step) instead of in front of it.  A CU's L2 -> LDS path moved ~30 B/clk in these kernels (round 2/3 measurements; not a hardware ceiling:
// tools/ubench/l2_feed.hip streams 64-74 B/clk/CU): the
// 32 KB of a 128 x 128 K-tile keep the issuing waves blocked for ~1,100 clk, and with ONE workgroup per CU (the ring
// launches of small per-GPU batches) all four waves sit in that phase together, then in the MFMA phase together -- the two
// add up (~2,000 clk per K-tile measured at 32 frames).  Issued between MFMAs, the pieces drain while the matrix pipe works.
template <int BM, int BN, int WM, int WN, int KS, bool POOL, bool PF, int MV, int NS, bool ILV = false, bool S2 = false>
__global__ __launch_bounds__(WM * WN * 64, (BM > 192) ? 2 : ((WM * WN == 8) ? 4 : (NS >= 3 ? 2 : 3))) void conv_igemm_kernel(ConvArgs p) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int NT = WM * WN * 64;                // threads per workgroup
    constexpr int LR = NT / 8;                      // tile rows covered per loader pass (8 chunks per row)
    constexpr int A_IT = BM / LR, B_IT = BN / LR;   // 16-B chunks per thread per tile
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    static_assert(BM % LR == 0 && BN % LR == 0, "loader geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // Persistent workgroups: logical block lb walks tiles lb, lb+grid, lb+2*grid, ... (tile_n fastest, so the
    // tiles in flight at any moment are neighbours and each XCD owns a contiguous run of them).
    const unsigned grid = gridDim.x;
    const unsigned lb = ec_xcd_remap(blockIdx.x, grid);
    static_assert(!ILV || (NS >= 3 && !PF && FM + FN <= 4), "interleaved pieces: ring mode with the inline-asm fragment stream");
    const int nvt = p.ntiles;
    int vt = (int)lb;
    if (vt >= nvt) return;
    int tile = vt;
    static_assert(MV <= BM && (MV == BM || !POOL), "padded tiles: non-pooled only");
    int m0 = (tile / p.ntn) * MV;
    int n0 = (tile % p.ntn) * BN;


    // ---- per-thread loader geometry (K-invariant): a byte offset and a 9-bit tap-validity mask per row ----
    // LDS-DMA writes lane l of a wave at (wave-uniform base + 16*l): the LDS image is lane-linear, so the XOR
    // swizzle is applied to the SOURCE chunk each lane fetches (and again on the fragment reads).
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);   // source 16-B chunk of the 128-B K row for LDS position tid&7
    const int lrow = tid >> 3;     // 0..LR-1
    unsigned a_off[A_IT];          // byte offset of the row's centre-tap pixel, channel 0
    unsigned a_msk[A_IT];          // bit (ky*3+kx) set <=> that tap lies inside the frame (bit 0 only for 1x1)
    unsigned b_off[B_IT];
    auto decode = [&]() {          // geometry of tile (m0, n0)
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + lrow + LR * i;
            int pix, y = 0, x = 0;
            if (POOL) {
                const int q = m >> 2, s2 = m & 3;
                const int Hp = p.H >> 1, Wp = p.W >> 1;
                const int b = q / (Hp * Wp);
                const int r2 = q - b * (Hp * Wp);
                const int yp = r2 / Wp;
                y = 2 * yp + (s2 >> 1);
                x = 2 * (r2 - yp * Wp) + (s2 & 1);
                pix = (b * p.H + y) * p.W + x;
            } else if (S2) {
                // stride 2 (torchvision's Bottleneck conv2 / downsample conv): row m is an OUTPUT pixel of the H/2 x W/2 map,
                // its centre tap is input pixel (2 yo, 2 xo); the tap-validity mask below is in input coordinates
                static_assert(!S2 || !POOL, "stride-2 instances have no fused pool");
                const int Ho = p.H >> 1, Wo = p.W >> 1;
                const int b = m / (Ho * Wo);
                const int r2 = m - b * (Ho * Wo);
                const int yo = r2 / Wo;
                y = 2 * yo;
                x = 2 * (r2 - yo * Wo);
                pix = (b * p.H + y) * p.W + x;
            } else if (KS == 1) {
                pix = m;               // raster order: pixel index == m, no halo
            } else {
                const int b = m / (p.H * p.W);
                const int r2 = m - b * (p.H * p.W);
                y = r2 / p.W;
                x = r2 - y * p.W;
                pix = m;
            }
            unsigned msk = 1u;
            if (KS == 3) {
                const unsigned xm = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);
                msk = (y > 0 ? xm : 0u) | (xm << 3) | (y < p.H - 1 ? (xm << 6) : 0u);
            }
            a_msk[i] = (m < p.M && (MV == BM || lrow + LR * i < MV)) ? msk : 0u;
            a_off[i] = (unsigned)pix * (unsigned)p.Cin * 2u;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) b_off[i] = ((unsigned)(n0 + lrow + LR * i) * (unsigned)p.K + chunk * 8) * 2u;
    };
    decode();

    // direct global -> LDS (LDS-DMA): no staging registers, no ds_write pass
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    // residual rows come through a descriptor too: rows past the tensor (ragged last tile) are out-of-range offsets that
    // read as zeros -- no per-load predicate, so the compiler issues all of a tile's loads back to back (with an
    // exec-masked branch around each it waited vmcnt(0) per load: 16 serial round trips in the 8-wave kernel's epilogue)
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res ? p.res_bytes : 0u, 0x00020000);
#endif
    const int wave_lds = wave * 1024;                       // 64 lanes x 16 B
    // per-K-tile addressing state shared by the pieces of one tile
    const int nk = (p.K + BK - 1) / BK;
    const int kt0 = 0, kt1 = nk, k_lim = p.K;   // the K-tile range; k >= k_lim reads as zeros
    int g_toff = 0; unsigned g_tapbit = 0; bool g_kin = false; int g_kt = 0;
    unsigned char* g_sa = smem; unsigned char* g_sb = smem;
    auto glds_begin = [&](int kt, int buf) {
        g_sa = smem + buf * (A_BYTES + B_BYTES) + wave_lds;
        g_sb = g_sa + A_BYTES;
        const int k = kt * BK + chunk * 8;
        int tap = 0;
        g_toff = k * 2;
        if (KS == 3) {
            int ci;
            if (p.cin_log2 >= 0) {            // power-of-two Cin (RN50): shift / mask
                tap = k >> p.cin_log2;
                ci = k & (p.Cin - 1);
            } else {                          // e.g. the 96 k channels of RN50x16
                tap = k / p.Cin;
                ci = k - tap * p.Cin;
            }
            const int ky = (tap * 11) >> 5;   // tap / 3 for tap in 0..8
            g_toff = (((ky - 1) * p.W + (tap - ky * 3 - 1)) * p.Cin + ci) * 2;
        }
        g_kin = k < k_lim;
        g_tapbit = g_kin ? (1u << tap) : 0u;
        g_kt = kt;
    };
    // piece q of the tile begun last: q < A_IT -> A rows, else B rows (each piece = one 1-KiB wave LDS-DMA)
    auto glds_piece = [&](auto qc) {
        [[maybe_unused]] constexpr int q = decltype(qc)::value;
#if defined(__HIP_DEVICE_COMPILE__)   // buffer descriptors: 32-bit offsets, out-of-range = zeros (as in conv_igemm8)
        if constexpr (q < A_IT) {
            const unsigned off = (a_msk[q] & g_tapbit) ? a_off[q] + (unsigned)g_toff : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t*)(g_sa + q * (LR * ROW_BYTES)), 16, off, 0, 0, 0);
        } else {
            constexpr int i = q - A_IT;
            const unsigned off = g_kin ? b_off[i] + (unsigned)(g_kt * (BK * 2)) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(g_sb + i * (LR * ROW_BYTES)), 16, off, 0, 0, 0);
        }
#endif
    };
    auto glds_tile = [&](int kt, int buf) {
        glds_begin(kt, buf);
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (glds_piece(std::integral_constant<int, Q>{}), ...); }
        (std::make_integer_sequence<int, A_IT + B_IT>{});
    };

    f32x16_t acc[FM][FN];

    // epilogue geometry
    constexpr int CH = BN / 8;                 // 16-B chunks per tile row
    constexpr int PITCH = BN * 2 + 16;         // bytes; +16 staggers banks between rows
    constexpr int RPP = NT / CH;               // rows per pass
    constexpr int OUT_ROWS = POOL ? BM / 4 : BM;
    constexpr int NPASS = OUT_ROWS / RPP;
    const int Mout = POOL ? (p.M >> 2) : p.M;
    const int srow = tid / CH, schunk = tid % CH;
    const bool has_res = !POOL && (p.res != nullptr);
    constexpr bool PREFETCH = PF && !POOL && (NPASS <= 8);   // PF: short-K (bandwidth-bound) launches only
    uint4 rres[PREFETCH ? NPASS : 1];

    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    auto compute = [&](int buf) {
        const unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* sb = sa + A_BYTES;
        // SWAPPED operands: D[n][m] = sum_k W[n][k] * A[m][k].  In the 32x32 C/D layout a lane then owns
        // ONE pixel (col = lane&31) and channels (r&3) + 8*(r>>2) + 4*(lane>>5): every 4 accumulator
        // registers are 4 consecutive channels -> 8-byte packed epilogue traffic.
        static_assert(NS < 3 || !(PF || FM + FN > 4), "ring mode needs the inline-asm fragment stream");
        if constexpr (PF || FM + FN > 4) {   // no registers to spare for the fragment ring there (the residual-prefetching variants spill: 68 -> 104 us)
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                s16x8_t af[FM], bfr[FN];
                const int c = ks * 2 + fhalf;
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[i] = *reinterpret_cast<const s16x8_t*>(sa + lds_off(wm * TM + i * 32 + frow, c));
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    bfr[j] = *reinterpret_cast<const s16x8_t*>(sb + lds_off(wn * TN + j * 32 + frow, c));
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, bfr[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
            }
            return;
        }
        constexpr int G = FM + FN, NKS = BK / 16, R = 2 * G;
        const unsigned lds_a = (unsigned)(unsigned long)(lds_void_t*)sa, lds_b = (unsigned)(unsigned long)(lds_void_t*)sb;
        FragRing<R> ring;
        auto addr = [&](auto ic) -> unsigned {                 // fragment q of k-step ks: q < FM -> A rows, else B rows
            constexpr int idx = decltype(ic)::value, ks = idx / G, q = idx % G;
            const int c = ks * 2 + fhalf;
            if constexpr (q < FM) return lds_a + (unsigned)lds_off(wm * TM + q * 32 + frow, c);
            else return lds_b + (unsigned)lds_off(wn * TN + (q - FM) * 32 + frow, c);
        };
        auto mma = [&](auto kc, FragRing<R>& rg) {
            constexpr int ks = decltype(kc)::value;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_t, rg.f[(ks * G + FM + j) % R]), __builtin_bit_cast(bf16x8_t, rg.f[(ks * G + i) % R]),
                        acc[i][j], 0, 0, 0);
            if constexpr (ILV) {   // this k-step's share of the next ring tile's pieces, pinned behind its MFMAs
                constexpr int P = A_IT + B_IT, lo = ks * P / NKS, hi = (ks + 1) * P / NKS;
                __builtin_amdgcn_sched_barrier(0);
                [&]<int... Q>(std::integer_sequence<int, Q...>) { (glds_piece(std::integral_constant<int, lo + Q>{}), ...); }
                (std::make_integer_sequence<int, hi - lo>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (fr_issue<Q, R>(ring, addr), ...); }(std::make_integer_sequence<int, G>{});
        fr_step<0, NKS, G, R>(ring, addr, mma);
    };

    // K pipeline: LDS-DMA of tile t+1 into LDS[(t+1)&1] is in flight while tile t computes from LDS[t&1];
    // the barrier at the end of the iteration carries the vmcnt(0) that lands it.  One barrier per K-tile.
    for (;;) {
        const int e_m0 = m0, e_n0 = n0;                      // coordinates of the tile being finished
        const int orow0 = POOL ? (e_m0 >> 2) : e_m0;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // residual tile of THIS output tile: issued now so its HBM latency overlaps the whole K loop
        if (PREFETCH && has_res) {
#pragma unroll
            for (int i = 0; i < (PREFETCH ? NPASS : 1); ++i) {
                const int row = i * RPP + srow;
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned off = (MV == BM || row < MV) ? ((unsigned)(orow0 + row) * (unsigned)p.Cout + (unsigned)(e_n0 + schunk * 8)) * 2u : 0xFFFFFFF0u;
                rres[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)off, 0, 0));
#endif
            }
        }
        if constexpr (NS >= 3) {
            constexpr int PIECES = A_IT + B_IT;                 // LDS-DMA instructions per wave per K-tile
            static_assert((NS - 2) * PIECES <= 60, "vmcnt is a 6-bit counter");
            // K-tiles past the end are all out-of-range offsets (zeros into a free stage): every K-tile then has exactly
            // PIECES younger-by-one-tile instructions behind it and the counted wait needs no tail cases
#pragma unroll
            for (int t = 0; t < NS - 1; ++t) glds_tile(kt0 + t, t);
            int st_c = 0, st_l = NS - 1;                        // stage of the K-tile computed / loaded next
            for (int kt = kt0; kt < kt1; ++kt) {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"((NS - 2) * PIECES) : "memory");   // own pieces of K-tile kt
                __builtin_amdgcn_s_barrier();                   // everyone's landed; everyone is done with K-tile kt-1
                if constexpr (ILV) {
                    glds_begin(kt + NS - 1, st_l);              // addresses now, the pieces from inside compute()
                    compute(st_c);
                } else {
                if (!(p.ablate & 1)) glds_tile(kt + NS - 1, st_l);   // into the stage K-tile kt-1 just left
                if (!(p.ablate & 2)) compute(st_c);
                }
                st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
                st_l = (st_l + 1 == NS) ? 0 : st_l + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero-filled tail tiles too: LDS becomes the epilogue image
            __syncthreads();
        } else {
        glds_tile(kt0, 0);
        __syncthreads();
        if (p.nbuf == 1) {
            // single LDS stage, two barriers per K-tile: smallest footprint (3 workgroups per CU); the load
            // latency of one workgroup is covered by the MFMAs of the other two
            for (int kt = kt0; kt < kt1; ++kt) {
                if (!(p.ablate & 2)) compute(0);
                if (kt + 1 < kt1) {
                    __syncthreads();
                    if (!(p.ablate & 1)) glds_tile(kt + 1, 0);
                }
                __syncthreads();
            }
        } else {
            for (int kt = kt0; kt < kt1; ++kt) {
                const int cur = (kt - kt0) & 1;
                const bool more = (kt + 1) < kt1;
                if (more && !(p.ablate & 1)) glds_tile(kt + 1, cur ^ 1);
                if (!(p.ablate & 2)) compute(cur);
                __syncthreads();
            }
        }
        }   // NS < 3
        vt += (int)grid;
        const bool has_next = vt < nvt;
        if (has_next) {
            tile = vt;
            m0 = (tile / p.ntn) * MV;
            n0 = (tile % p.ntn) * BN;
            decode();
        }
        if (!(p.ablate & 8)) {
    // ---- epilogue: staged through LDS so every global access is a coalesced 16-B chunk ----
    //   1. prefetched residual tile -> LDS                [only with a residual]
    //   2. each lane folds bias/residual/activation (or the 2x2 pool) into 4 consecutive channels of its
    //      pixel in fp32, rounds ONCE to bf16 (v_cvt_pk_bf16_f32) and writes 8 bytes to its own LDS slot
    //   3. LDS -> global as 16-B row chunks
    if (has_res) {
        if (PREFETCH) {
#pragma unroll
            for (int i = 0; i < (PREFETCH ? NPASS : 1); ++i)
                *reinterpret_cast<uint4*>(smem + (i * RPP + srow) * PITCH + schunk * 16) = rres[i];
        } else {
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int row = i * RPP + srow;
                uint4 v = make_uint4(0, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned off = (MV == BM || row < MV) ? ((unsigned)(orow0 + row) * (unsigned)p.Cout + (unsigned)(e_n0 + schunk * 8)) * 2u : 0xFFFFFFF0u;
                v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)off, 0, 0));
#endif
                *reinterpret_cast<uint4*>(smem + row * PITCH + schunk * 16) = v;
            }
        }
        __syncthreads();
    }
    epi_stage_dispatch<POOL, PF, FM, FN, PITCH>(acc, smem, p.bias ? p.bias + e_n0 : nullptr, wm * TM, wn * TN, lane, p.act, has_res);
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < OUT_ROWS; r0 += RPP) {
        const int row = r0 + srow;
        if (orow0 + row < Mout && (MV == BM || row < MV))
            *reinterpret_cast<uint4*>(p.out + (long)(orow0 + row) * p.ldo + e_n0 + schunk * 8) =
                *reinterpret_cast<const uint4*>(smem + row * PITCH + schunk * 16);
    }
        }   // !(ablate & 8)
        if (!has_next) break;
        __syncthreads();                                     // epilogue's LDS reads are done before the next store_tile
    }
}

template <int BM, int BN, int WM, int WN, int KS, bool POOL, bool PF = false, int MV = BM, int NS = 0, bool ILV = false, bool S2 = false>
int launch(const ConvArgs& a, hipStream_t s) {
    ConvArgs p = a;
    p.ntn = a.Cout / BN;
    const int ntm = (a.M + MV - 1) / MV;
    p.ntiles = ntm * p.ntn;
    const size_t lds_max = 2 * (size_t)(BM + BN) * ROW_BYTES;
    const size_t epi = (size_t)(POOL ? BM / 4 : BM) * (BN * 2 + 16);
    p.nbuf = (BM * BN >= 256 * 256) ? 2 : 1;
    p.ablate = ec_tools_ablate();
    size_t lds = (size_t)(NS >= 3 ? NS : p.nbuf) * (BM + BN) * ROW_BYTES;
    if (lds < epi) lds = epi;
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, KS, POOL, PF, MV, NS, ILV, S2>;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        const size_t want = NS >= 3 ? lds : (lds_max > epi ? lds_max : epi);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
    }
    // persistent: 3 workgroups per CU (<= 168 VGPRs, single 35-KB LDS stage), each walking several tiles
    const int wg_cap = 768;
    // ring mode: as many workgroups per CU as its NS stages fit (160 KiB of LDS, 2 waves per SIMD by registers)
    const int ring_per_cu = NS >= 3 ? (int)std::min<size_t>(2, (160 * 1024) / lds) : 0;
    const int cap = NS >= 3 ? 256 * ring_per_cu : ((BM * BN >= 256 * 256) ? 256 : wg_cap);      // 8-wave 256x256 tiles: one workgroup per CU
    const int nwg = p.ntiles < cap ? p.ntiles : cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// conv_igemm8: 8-wave "ping-pong" kernel for the compute-bound layers (round 2).
//
// Why: with 128x128 tiles the CU's vector-memory path (64 B/clk: one 1-KiB LDS-DMA piece per 16 clk) needs as long to
// feed a K-tile (32 KB) as the four SIMDs need to multiply it (16 MFMAs x 32 clk), and because every wave of the
// workgroup issues its pieces at the same point of the loop the two never overlap well (measured: MFMA 33 us + loads
// 42 us ~ the sum).  This kernel halves the bytes per flop (256 x 256 x 64 tile, 64 KB per K-tile for 4x the MFMAs) and
// makes the overlap structural:
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x (BN/4), two waves per SIMD; one workgroup per CU, two 64-KB LDS stages;
//   * the two waves of a SIMD belong to different GROUPS (waves 0-3 / 4-7) that run the same program shifted by one
//     segment: while one group is in a MEMORY segment (LDS-DMA issue for the next K-tile + ds_read of the fragments of
//     half a K-tile), the other is in a COMPUTE segment (16 back-to-back MFMAs at raised priority).  Segments are
//     separated by raw s_barrier (no vmcnt drain); group 1 enters the loop through one extra barrier.
//   * LDS-DMA pieces of K-tile t+1 are issued in the first two segments of K-tile t and waited for (s_waitcnt vmcnt(0),
//     each wave for its own pieces) two segments later, just before the barrier that precedes the first read of t+1:
//         s = 4t   : g0 MEM0(t)  [issue A pieces t+1 | read frags (t, k 0..31)]        g1 CMP1(t-1)
//         s = 4t+1 : g0 CMP0(t)  [16 MFMA + issue B pieces t+1]                       g1 MEM0(t)
//         s = 4t+2 : g0 MEM1(t)  [read frags (t, k 32..63)]                            g1 CMP0(t)
//         s = 4t+3 : g0 CMP1(t)  [16 MFMA; vmcnt(0)]                                   g1 MEM1(t) [vmcnt(0)]
//     WAR: the stage of K-tile t-1 is re-filled from s = 4t on; its last reader is g1's MEM1(t-1) at s = 4t-1, which
//     ends with lgkmcnt(0) before the barrier.  RAW: every wave has waited for its pieces of t+1 before the barrier
//     that ends s = 4t+3; the first reader is g0's MEM0(t+1) at s = 4t+4.
// Operand fetch (implicit im2col, out-of-range buffer offsets for padding), LDS swizzle, swapped MFMA operands and the epilogue
// (bias / residual / ReLU / fused 2x2 average pool through LDS, 16-byte coalesced stores) are those of conv_igemm_kernel.
// Requires Cin % 64 == 0 (a K-tile never straddles a 3x3 tap) and Cout % BN == 0.
__device__ unsigned long long ec_dbg_stamps[2 * 1024];   // profiling only (tools build, EC_CONV_ABLATE & 32): s_memtime stamps of block 0

// X3 (policy compressor, ec_gemm_bf16a_x3): the weight operand is an fp32 matrix split into three bf16 planes
// ([Cout][3][K], lowest plane first in the K walk); a K-tile of A is walked three times, once against each plane, into
// the same accumulators, and the epilogue writes fp32 (bias / ReLU) -- the bf16x3 "exact fp32" product of gemm_f32.hip on
// this kernel's schedule.
// BM (round 6): 192-row tiles for GEMMs whose 256-row tiling leaves CUs idle -- ViT-B/32's N = 768 GEMMs at 6,400 tokens are
// 25 x 6 = 150 tiles on 256 CUs; 34 x 6 = 204 tiles of 192 x 128 take 0.75 of a 256 x 128 tile's time each (ec_gemm_bf16_ln8).
// XP (X3 only): planes of the weight operand that are walked -- 3 = the exact-fp32 product; 2 = the two leading planes (16
// mantissa bits of W, relative product error 2^-17: ec_gemm_bf16a_xp, the learn pass's compressor conv under EC_POLICY_FAST).
template <int BN, int KS, bool POOL, int ABL, bool X3 = false, bool S2 = false, int BM = 256, int XP = 3>
__global__ __launch_bounds__(512, 2) void conv_igemm8_kernel(ConvArgs p) {
    // wave grid: 2 (M) x 4 (N) for 256-wide tiles (wave tile 128 x 64); 4 x 2 for 128-wide tiles (wave tile 64 x 64:
    // 4 fragment reads per 4 MFMAs instead of the 5 a 128 x 32 wave tile needs); 192-row tiles: 2 x 4, wave tile 96 x 32
    static_assert(BM == 256 || (BM == 192 && BN == 128 && !POOL && !X3), "tile heights: 256; 192 for 128-wide 1x1 / 3x3 tiles");
    constexpr int WN = (BN >= 256 || BM == 192) ? 4 : 2, WMW = 8 / WN;
    constexpr int TM = BM / WMW, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int NT = 512, LR = NT / 8;
    constexpr int A_IT = BM / LR, B_IT = BN / LR;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
    static_assert(FN >= 1 && B_IT >= 1, "BN must be 128 or 256");
    // ABL bit 512 (production variant of the 128-wide tiles): LONG SEGMENTS -- a wave reads the fragments of a WHOLE K-tile
    // in one memory segment and issues its 4 FM FN MFMAs in one compute segment: two barrier-separated segments per K-tile
    // instead of four.  Every segment costs ~350 clk of barrier skew / waits / issue on top of its MFMAs (measured: period =
    // 4 x (512 + 350) clk with 256-wide tiles, 4 x (256 + 350) with 128-wide ones), so the 128-wide tile, whose segments
    // are only 8 MFMAs long, pays the most.  Needs three LDS stages (K-tile kt + 2 is fetched while kt computes: with two
    // segments per K-tile a distance-1 prefetch would have to be waited for in the segment that issues it) and twice the
    // fragment registers -- which only the 128-wide tile (64 accumulator registers) has.
    constexpr bool LS = (ABL & 512) != 0 && BN == 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const unsigned long long t_entry = (ABL & 32) ? __builtin_amdgcn_s_memtime() : 0ull;   // profiling only
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int grp = wave >> 2;                          // waves 0-3 / 4-7: the two waves of every SIMD

    const int tile = (int)ec_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= p.ntiles) return;
    const int m0 = (tile / p.ntn) * BM;
    const int n0 = (tile % p.ntn) * BN;

    // LayerNorm folded in (consumer side): (rstd, -mean * rstd) of the tile's 256 rows from the producer's partial records
    // {sum, M2, count} (Chan's combination: no E[x^2] - mean^2 cancellation), into the auxiliary LDS area behind the stages.
    // Read by the epilogue only, i.e. behind the __syncthreads() that ends the K walk.
    constexpr bool LNOK = (KS == 1) && !POOL && !X3 && !S2;
    if constexpr (LNOK) {
        if (p.ln_stats && tid < BM) {
            float2 ab = make_float2(0.f, 0.f);
            const int row = m0 + tid;
            if (row < p.M) {
                const float4* rec = reinterpret_cast<const float4*>(p.ln_stats) + (size_t)row * p.ln_np;
                float4 r[8];
                float S = 0.f, Nn = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < p.ln_np) { r[q] = rec[q]; S += r[q].x; Nn += r[q].z; }
                const float mean = S / Nn;
                float M2 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < p.ln_np) { const float d = r[q].x / r[q].z - mean; M2 += r[q].y + r[q].z * d * d; }
                const float rstd = rsqrtf(M2 / Nn + p.ln_eps);
                ab = make_float2(rstd, -mean * rstd);
            }
            reinterpret_cast<float2*>(smem + p.aux_off)[tid] = ab;
        }
    }

    // ---- loader geometry (K-invariant) ----
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    const int lrow = tid >> 3;
    unsigned a_off[A_IT], a_msk[A_IT], b_off[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + lrow + LR * i;
        int pix, y = 0, x = 0;
        if (POOL) {
            const int q = m >> 2, s2 = m & 3;
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            const int b = q / (Hp * Wp);
            const int r2 = q - b * (Hp * Wp);
            const int yp = r2 / Wp;
            y = 2 * yp + (s2 >> 1);
            x = 2 * (r2 - yp * Wp) + (s2 & 1);
            pix = (b * p.H + y) * p.W + x;
        } else if (S2) {      // stride 2 (torchvision's strided convs, ec_conv_bf16_s2): row m = output pixel, centre tap at (2 yo, 2 xo)
            static_assert(!S2 || !POOL, "stride-2 instances have no fused pool");
            const int Ho = p.H >> 1, Wo = p.W >> 1;
            const int b = m / (Ho * Wo);
            const int r2 = m - b * (Ho * Wo);
            const int yo = r2 / Wo;
            y = 2 * yo;
            x = 2 * (r2 - yo * Wo);
            pix = (b * p.H + y) * p.W + x;
        } else if (KS == 1) {
            pix = m;
        } else {
            const int b = m / (p.H * p.W);
            const int r2 = m - b * (p.H * p.W);
            y = r2 / p.W;
            x = r2 - y * p.W;
            pix = m;
        }
        unsigned msk = 1u;
        if (KS == 3) {
            const unsigned xm = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);
            msk = (y > 0 ? xm : 0u) | (xm << 3) | (y < p.H - 1 ? (xm << 6) : 0u);
        }
        a_msk[i] = (m < p.M) ? msk : 0u;
        a_off[i] = (unsigned)pix * (unsigned)p.Cin * 2u + (unsigned)chunk * 16u;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_off[i] = ((unsigned)(n0 + lrow + LR * i) * (unsigned)(X3 ? 3 * p.K : p.K) + chunk * 8) * 2u;

#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the kernel's stub; the buffer builtins are device-only
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
#endif
    const int wave_lds = wave * 1024;
    const int nk = (X3 ? XP : 1) * (p.K / BK);          // Cin % 64 == 0: whole K-tiles, one tap per K-tile

    // piece q of K-tile kt into stage buf: q < A_IT -> 8 pixel rows per wave, else 8 weight rows
    int g_toff = 0; unsigned g_tapbit = 1u; int g_kb = 0;
    unsigned char* g_sa = smem; unsigned char* g_sb = smem; bool g_issue_a = true;
    auto glds_begin = [&](int kt, int buf) {            // buf: LDS stage of K-tile kt (kt & 1; kt % 3 with long segments)
        g_sa = smem + buf * STAGE + wave_lds;
        g_sb = g_sa + A_BYTES;
        g_kb = kt * (BK * 2);
        g_toff = kt * (BK * 2);
        g_tapbit = 1u;
        if constexpr ((ABL & 64) != 0) g_issue_a = (kt % 9 == 0) || (kt % 9 == 4);   // timing only: the A bytes a halo window would move
        if (X3) {                                       // K-tile kt = (A chunk kt / 3) x (weight plane 2 - kt % 3)
            // LDS: [A chunk 0][A chunk 1][B 0][B 1] -- the A chunk is fetched ONCE (with the K-tile of its first plane)
            // and read by the three K-tiles of the chunk; only the 16-KB plane tiles alternate per K-tile
            const int ch = kt / XP, pl = (XP - 1) - (kt - XP * ch);          // (lowest walked plane first)
            g_toff = ch * (BK * 2);
            g_kb = (pl * p.K + ch * BK) * 2;
            g_sa = smem + (ch & 1) * A_BYTES + wave_lds;
            g_sb = smem + 2 * A_BYTES + (LS ? buf : (kt & 1)) * B_BYTES + wave_lds;
            g_issue_a = (kt == XP * ch);
        }
        if (KS == 3) {
            const int k = kt * BK;
            const int tap = k >> p.cin_log2, ci = k & (p.Cin - 1);
            const int ky = (tap * 11) >> 5;
            g_toff = (((ky - 1) * p.W + (tap - ky * 3 - 1)) * p.Cin + ci) * 2;
            g_tapbit = 1u << tap;
        }
    };
    auto glds_piece = [&](auto qc) {
        [[maybe_unused]] constexpr int q = decltype(qc)::value;
        // buffer_load ... lds through a descriptor: a 32-bit per-lane offset (no 64-bit pointer arithmetic), and a padding tap
        // is an out-of-range offset, which the hardware answers with zeros (no zero page, no pointer select)
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (q < A_IT) {
            if ((!X3 && !(ABL & 64)) || g_issue_a) {
                const unsigned off = (a_msk[q] & g_tapbit) ? a_off[q] + (unsigned)g_toff : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t*)(g_sa + q * (LR * ROW_BYTES)), 16, off, 0, 0, 0);
            }
        } else {
            constexpr int i = q - A_IT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(g_sb + i * (LR * ROW_BYTES)), 16,
                                                     b_off[i] + (unsigned)g_kb, 0, 0, 0);
        }
#endif
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    // ABL bit 256 (a production variant, not an ablation): DIRECT-B -- the weight fragments of a wave go global -> VGPR
    // (one 16-B buffer load per fragment, issued a K-tile ahead into the registers the MFMAs of the same half K-tile have
    // just read), only the im2col operand is staged through LDS: half the LDS-DMA pieces and a third of the fragment
    // reads per K-tile leave the loop
    constexpr bool DB = (ABL & 256) != 0 && !X3;
    // (fragment-order weights, ec_pack_wfrag: the 64 x 16 B of fragment (32-column block nb, k-step ks) are contiguous, so a
    //  fragment load is ONE coalesced 1-KiB access; straight out of the [Cout][K] layout the same load touches 32 lines and
    //  the texture path becomes the bound: 57.6 -> 94.8 us on the 3x3 256 -> 256 @14 layer, measured round 3)
    unsigned bfo[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j)
        bfo[j] = (unsigned)((n0 >> 5) + wn * (TN / 32) + j) * (unsigned)(p.K >> 4) * 1024u + (unsigned)lane * 16u;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_wf = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
#endif
    s16x8_t fbd[2][2][FN];                              // [half][k-step][fragment]
    auto bload = [&](int kt, int h, int u) {            // fragments (kt, k-step 2h + u) of this wave's FN column blocks
#if defined(__HIP_DEVICE_COMPILE__)
        const int soff = (kt * 4 + h * 2 + u) * 1024;
#pragma unroll
        for (int j = 0; j < FN; ++j)
            fbd[h][u][j] = __builtin_bit_cast(s16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_wf, (int)bfo[j], soff, 0));
#endif
    };
    // fragment byte offsets inside a stage: row-dependent part (the swizzle XOR depends on (row>>1)&7 == (frow>>1)&7
    // for rows that are multiples of 32 apart), k-step part added per read
    int fa_base[FM], fb_base[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) fa_base[i] = (wm * TM + i * 32 + frow) * ROW_BYTES;
#pragma unroll
    for (int j = 0; j < FN; ++j) fb_base[j] = A_BYTES + (wn * TN + j * 32 + frow) * ROW_BYTES;
    const int fsw = (frow >> 1) & 7;

    s16x8_t fa[LS ? 4 : 2][FM], fb[LS ? 4 : 2][FN];     // (long segments: the four k-steps of a K-tile at once)
    auto read_half = [&](int kt, int stg, auto hc) {    // fragments of k-steps 2h, 2h+1 of K-tile kt (LDS stage stg)
        constexpr int h = decltype(hc)::value;
        constexpr int fo = LS ? 2 * h : 0;
        const unsigned char* sta = smem + stg * STAGE;
        const unsigned char* stb = sta;
        if (X3) {                                       // (fb_base carries + A_BYTES: see the LDS map in glds_begin)
            sta = smem + ((kt / XP) & 1) * A_BYTES;
            stb = smem + A_BYTES + stg * B_BYTES;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = ((h * 2 + u) * 2 + fhalf) ^ fsw;
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[fo + u][i] = *reinterpret_cast<const s16x8_t*>(sta + fa_base[i] + (c << 4));
            if constexpr (!DB) {
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[fo + u][j] = *reinterpret_cast<const s16x8_t*>(stb + fb_base[j] + (c << 4));
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: K-tile 0 into stage 0 ----
    constexpr int NPIECE = DB ? A_IT : A_IT + B_IT;     // LDS-DMA pieces per wave and K-tile
    glds_begin(0, 0);
    [&]<int... Q>(std::integer_sequence<int, Q...>) { (glds_piece(std::integral_constant<int, Q>{}), ...); }
    (std::make_integer_sequence<int, NPIECE>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DB) { bload(0, 0, 0); bload(0, 0, 1); bload(0, 1, 0); bload(0, 1, 1); }
    auto issue_all = [&](int kt, int buf) {             // all of this wave's pieces of K-tile kt, back to back
        glds_begin(kt, buf);
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (glds_piece(std::integral_constant<int, Q>{}), ...); }
        (std::make_integer_sequence<int, NPIECE>{});
    };
    // Piece offsets are prepared one segment AHEAD of their issue (in the wave's MEM segment, whose VALU work hides
    // behind the partner wave's MFMAs), so that inside a CMP segment a piece is only {s_mov m0, buffer_load ... lds}:
    // measured (s_memtime stamps, round 3) a CMP segment is 16 x 32 clk of MFMA issue plus whatever the wave issues
    // between them -- address VALU, mask selects and, above all, taken branches (a skipped `if (issue)` per MFMA pair
    // cost ~25-30 clk each: 705-780 clk for a piece-free segment instead of 512).
    unsigned poff[A_IT + B_IT];
    auto prep = [&](int kt, int buf) {                  // geometry + per-lane offsets of this wave's pieces of K-tile kt
        glds_begin(kt, buf);
#pragma unroll
        for (int q = 0; q < A_IT; ++q) poff[q] = (a_msk[q] & g_tapbit) ? a_off[q] + (unsigned)g_toff : 0xFFFFFFF0u;
        if constexpr (!DB) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) poff[A_IT + i] = b_off[i] + (unsigned)g_kb;
        }
    };
    auto piece_pre = [&](auto qc) {                     // piece q of the K-tile prepared last
        [[maybe_unused]] constexpr int q = decltype(qc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (q < A_IT) {
            if ((!X3 && !(ABL & 64)) || g_issue_a)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t*)(g_sa + q * (LR * ROW_BYTES)), 16, poff[q], 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(g_sb + (q - A_IT) * (LR * ROW_BYTES)), 16, poff[q], 0, 0, 0);
        }
#endif
    };
    // 16 MFMAs (the two k-steps held in fa/fb); ISSUE: the wave's prepared LDS-DMA pieces go out in their shadow, one
    // piece per 2 MFMAs.  ISSUE is a template flag: the piece-free instance is 16 MFMAs with nothing between them.
    auto mfma16 = [&](auto issue_c, auto hc, int kt) {
        constexpr bool ISSUE = decltype(issue_c)::value;
        [[maybe_unused]] constexpr int H = decltype(hc)::value;
        constexpr int fo = LS ? 2 * H : 0;              // fragment set; long segments: half of the pieces per call
        constexpr int PN = LS ? (NPIECE + 1 - H) / 2 : NPIECE, PBASE = LS ? (H ? (NPIECE + 1) / 2 : 0) : 0;
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            ([&] {
                constexpr int u = Q / (FM * FN), r = Q % (FM * FN), i = r / FN, j = r % FN;
                if constexpr (!(ABL & 2))
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_t, DB ? fbd[H][u][j] : fb[fo + u][j]), __builtin_bit_cast(bf16x8_t, fa[fo + u][i]), acc[i][j], 0, 0, 0);
                if constexpr (DB && r == FM * FN - 1) {          // k-step u of this half is consumed: refill it for K-tile kt + 1
                    bload(kt + 1, H, u);
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int NP = PN, EVERY = (2 * FM * FN) / (NP > 0 ? NP : 1);
                if constexpr (ISSUE && NP > 0 && Q % EVERY == EVERY - 1 && Q / EVERY < NP) {
                    piece_pre(std::integral_constant<int, PBASE + Q / EVERY>{});
                    // keep one piece per EVERY MFMAs: left alone the scheduler clusters all of them behind the first MFMAs
                    // (measured: 64.6 vs 60.1 us on 3x3 256->256 @14x14, piece segment 1060 vs 852 clk)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 2 * FM * FN>{});
    };
    using BT = std::true_type;
    using BF = std::false_type;
    if constexpr (LS) {
        // long segments: K-tile 1 is fetched by EVERY wave before the loop (K-tile kt + 2 is what a wave fetches during kt)
        if (nk > 1) issue_all(1, 1);
        if (grp) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one segment behind group 0
    } else
    if (grp) {                                          // stagger: group 1 runs one segment behind group 0 ...
        if (nk > 1 && !(ABL & 1)) issue_all(1, 1);      // ... and uses the slot to fetch K-tile 1 (its "CMP1(-1)")
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    }

    constexpr int abl = ABL;                            // profiling only: 1 no loads, 2 no MFMA, 4 no ds_read, 16 no barriers, 32 stamps
    const bool dbg = (abl & 32) && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
    unsigned long long* dbg_lds = reinterpret_cast<unsigned long long*>(smem + 2 * STAGE) + grp * 256;
    int dbg_i = 0;
    auto stamp = [&]() { if constexpr ((abl & 32) != 0) if (dbg && dbg_i < 256) dbg_lds[dbg_i++] = __builtin_amdgcn_s_memtime(); };
    // One K-tile of group G; ISSUE: this K-tile's issuing CMP segment (CMP0 for group 0, CMP1 for group 1) carries pieces.
    // Both are compile-time: each group runs its own copy of the loop (steady state + a piece-free tail), so no segment
    // contains a branch and the accumulators keep their registers across the whole K walk.
    auto ktile = [&](int kt, auto gc, auto ic) {
        constexpr int G = decltype(gc)::value;
        constexpr bool ISSUE = decltype(ic)::value;
        const int cur = kt & 1;
        // ---- MEM0 ---- (group 0 also prepares the offsets of the pieces it issues in CMP0: K-tile kt+1 -> stage cur^1)
        if constexpr (!(abl & 4)) read_half(kt, cur, I0{});
        if constexpr (G == 0 && ISSUE) prep(kt + 1, cur ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        if constexpr (!(abl & 16)) __builtin_amdgcn_s_barrier();
        stamp();
        // ---- CMP0: group 0 issues its pieces of K-tile kt+1 in the MFMA shadow (s = 4kt+1; waited for at s = 4kt+3) ----
        __builtin_amdgcn_s_setprio(1);
        mfma16(std::bool_constant<(G == 0 && ISSUE)>{}, I0{}, kt);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        if constexpr (!(abl & 16)) __builtin_amdgcn_s_barrier();
        stamp();
        // ---- MEM1 ---- (group 1 prepares the pieces of its CMP1: K-tile kt+2 -> stage cur)
        if constexpr (!(abl & 4)) read_half(kt, cur, I1{});
        if constexpr (G == 1 && ISSUE) prep(kt + 2, cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (DIRECT-B: the 2 FN fragment loads of the CMP segment in between are younger than the pieces and stay in flight)
        if constexpr (G == 1) { if constexpr (DB) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * FN) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        if constexpr (!(abl & 16)) __builtin_amdgcn_s_barrier();
        stamp();
        // ---- CMP1: group 1 issues its pieces of K-tile kt+2 (its CMP1(kt) is global segment 4(kt+1): the stage of
        //      K-tile kt is free -- its last reader was this group's own MEM1(kt)); waited for at the end of its MEM1(kt+1)
        __builtin_amdgcn_s_setprio(1);
        mfma16(std::bool_constant<(G == 1 && ISSUE)>{}, I1{}, kt);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (G == 0) { if constexpr (DB) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * FN) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        if constexpr (!(abl & 16)) __builtin_amdgcn_s_barrier();
        stamp();
    };
    auto kloop = [&](auto gc) {
        constexpr int G = decltype(gc)::value;
        // group 0 fetches K-tile kt+1 during K-tile kt, group 1 K-tile kt+2: the last 1 (2) K-tiles issue nothing
        const int n_issue = (abl & 1) ? 0 : nk - 1 - G;
        int kt = 0;
        for (; kt < n_issue; ++kt) ktile(kt, gc, BT{});
        for (; kt < nk; ++kt) ktile(kt, gc, BF{});
    };
    // (a static s_setprio 1 for group 1 instead of the per-segment flips: -2..-5 % on single launches, but -4 % END TO END --
    //  a wave that stays at raised priority through its memory segments also wins arbitration against the OTHER stream's
    //  kernels; measured round 3, not kept)
    // ---- long segments: [MEM(kt): all fragments of K-tile kt | wait for the pieces issued during K-tile kt - 1] barrier
    //      [CMP(kt): 4 FM FN MFMAs, the wave's pieces of K-tile kt + 2 in their shadow] barrier.  Group 0 runs MEM(kt) in global
    //      segment 2 kt, group 1 in 2 kt + 1.  RAW: a wave waits (vmcnt(0) at the end of MEM(kt)) for the pieces of K-tile
    //      kt + 1 it issued in CMP(kt - 1); both groups have passed that wait and a barrier before group 0's MEM(kt + 1) at
    //      2 kt + 2.  WAR: K-tile kt + 2 goes into the stage of kt - 1, last read in segment 2 kt - 1 (group 1's MEM(kt - 1));
    //      the earliest issue is group 0's CMP(kt) at 2 kt + 1.  Same K walk, same accumulation order as the 4-segment loop.
    auto ktile_ls = [&](int kt, int stg, int stg2, auto ic) {
        constexpr bool ISSUE = decltype(ic)::value;
        read_half(kt, stg, I0{});
        read_half(kt, stg, I1{});
        if constexpr (ISSUE) prep(kt + 2, stg2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
        mfma16(std::bool_constant<ISSUE>{}, I0{}, kt);
        mfma16(std::bool_constant<ISSUE>{}, I1{}, kt);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (LS) {
        int stg = 0, stg2 = 2, kt = 0;                  // stages of K-tiles kt and kt + 2
        for (; kt < nk - 2; ++kt) {
            ktile_ls(kt, stg, stg2, BT{});
            stg = (stg == 2) ? 0 : stg + 1;
            stg2 = (stg2 == 2) ? 0 : stg2 + 1;
        }
        for (; kt < nk; ++kt) {
            ktile_ls(kt, stg, stg2, BF{});
            stg = (stg == 2) ? 0 : stg + 1;
        }
        if (!grp) __builtin_amdgcn_s_barrier();         // matches group 1's extra entry barrier
    } else {
    if (grp) kloop(I1{}); else kloop(I0{});
    if (!grp && !(ABL & 16)) __builtin_amdgcn_s_barrier();   // matches group 1's extra entry barrier
    }
    __syncthreads();                                    // every wave is done with the stages: LDS becomes the epilogue image
    const unsigned long long t_loop_end = (ABL & 32) ? __builtin_amdgcn_s_memtime() : 0ull;
    if constexpr ((abl & 32) != 0) if (dbg) {
        for (int i = 0; i < 256; ++i) ec_dbg_stamps[grp * 1024 + i] = i < dbg_i ? dbg_lds[i] : 0ull;
        ec_dbg_stamps[grp * 1024 + 300] = t_entry;
        ec_dbg_stamps[grp * 1024 + 301] = t_loop_end;
    }
    if constexpr ((abl & 32) != 0) __syncthreads();

    if constexpr (X3) {                                 // fp32 rows through LDS, 128 tile rows per pass
        static_assert(!X3 || (BN == 128 && !POOL && KS == 1), "X3: 128-wide 1x1 tiles");
        constexpr int PITCHF = BN * 4 + 16;
        float* outf = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if ((wm >> 1) == pass) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int lcol = wn * TN + j * 32 + 8 * g + 4 * fhalf;
                        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n0 + lcol);
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            const int lr = (wm & 1) * TM + i * 32 + frow;
                            float4 v = make_float4(acc[i][j][4 * g + 0] + bv.x, acc[i][j][4 * g + 1] + bv.y,
                                                   acc[i][j][4 * g + 2] + bv.z, acc[i][j][4 * g + 3] + bv.w);
                            if (p.act == EC_ACT_RELU) {
                                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                            }
                            *reinterpret_cast<float4*>(smem + lr * PITCHF + lcol * 4) = v;
                        }
                    }
            }
            __syncthreads();
#pragma unroll
            for (int r0 = 0; r0 < 128; r0 += NT / 32) {
                const int row = r0 + (tid >> 5), ch = tid & 31;
                const long grow = (long)m0 + pass * 128 + row;
                if (grow < p.M)
                    *reinterpret_cast<float4*>(outf + grow * p.Cout + n0 + ch * 4) =
                        *reinterpret_cast<const float4*>(smem + row * PITCHF + ch * 16);
            }
            __syncthreads();
        }
        return;
    }
    // ---- epilogue (as conv_igemm_kernel) ----
    if constexpr ((ABL & 8) != 0) return;               // profiling only: no epilogue at all
    constexpr int CH = BN / 8;
    constexpr int PITCH = BN * 2 + 16;
    constexpr int RPP = NT / CH;
    constexpr int OUT_ROWS = POOL ? BM / 4 : BM;
    constexpr int NPASS = OUT_ROWS / RPP;
    const int Mout = POOL ? (p.M >> 2) : p.M;
    const int orow0 = POOL ? (m0 >> 2) : m0;
    const int srow = tid / CH, schunk = tid % CH;
    const bool has_res = !POOL && (p.res != nullptr);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_res8 = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res ? p.res_bytes : 0u, 0x00020000);
#endif
    if (has_res) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = i * RPP + srow;
            uint4 v = make_uint4(0, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)   // (descriptor load: rows past the tensor read as zeros, all NPASS loads in flight at once)
            const unsigned off = ((unsigned)(orow0 + row) * (unsigned)p.Cout + (unsigned)(n0 + schunk * 8)) * 2u;
            v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res8, (int)off, 0, 0));
#endif
            *reinterpret_cast<uint4*>(smem + row * PITCH + schunk * 16) = v;
        }
        __syncthreads();
    }
    bool ln_done = false;
    if constexpr (LNOK) {
        if (p.ln_stats) {   // (wave-uniform; LayerNorm-folded launches carry no residual: ec_gemm_bf16_ln8)
            const float2* rowab = reinterpret_cast<const float2*>(smem + p.aux_off);
            if (p.act == EC_ACT_QUICKGELU) epi_stage_ln<EC_ACT_QUICKGELU, FM, FN, PITCH>(acc, smem, p.ln_s + n0, p.bias + n0, rowab, wm * TM, wn * TN, lane);
            else epi_stage_ln<EC_ACT_NONE, FM, FN, PITCH>(acc, smem, p.ln_s + n0, p.bias + n0, rowab, wm * TM, wn * TN, lane);
            ln_done = true;
        }
    }
    if (!ln_done)
    epi_stage_dispatch<POOL, false, FM, FN, PITCH>(acc, smem, p.bias ? p.bias + n0 : nullptr, wm * TM, wn * TN, lane, p.act, has_res);
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < OUT_ROWS; r0 += RPP) {
        const int row = r0 + srow;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + row * PITCH + schunk * 16);
        if (orow0 + row < Mout && !(ABL & 128))         // (ABL & 128, profiling only: LDS staging without the global stores)
            *reinterpret_cast<uint4*>(p.out + (long)(orow0 + row) * p.ldo + n0 + schunk * 8) = v;
        if constexpr (LNOK) {
            if (p.stats_out) {
                // this tile's LayerNorm record of the row: the CH lanes that hold its BN rounded values reduce {sum, then M2 about
                // the tile mean} with a butterfly (CH = 16 / 32 consecutive lanes of a wave), fixed order -> deterministic
                float x[8] = {ec_lo(v.x), ec_hi(v.x), ec_lo(v.y), ec_hi(v.y), ec_lo(v.z), ec_hi(v.z), ec_lo(v.w), ec_hi(v.w)};
                float sm = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
#pragma unroll
                for (int o = CH / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
                const float mu = sm * (1.f / (float)BN);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = x[e] - mu; q = fmaf(d, d, q); }
#pragma unroll
                for (int o = CH / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                if (schunk == 0 && orow0 + row < Mout)
                    reinterpret_cast<float4*>(p.stats_out)[(size_t)(orow0 + row) * p.ntn + (tile % p.ntn)] = make_float4(sm, q, (float)BN, 0.f);
            }
        }
    }
    if constexpr ((ABL & 32) != 0) if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) ec_dbg_stamps[grp * 1024 + 302] = __builtin_amdgcn_s_memtime();
}

template <int BN, int KS, bool POOL, bool X3 = false, bool S2 = false, int BM = 256, int XP = 3>
int launch8(const ConvArgs& a, hipStream_t s) {
    ConvArgs p = a;
    p.ntn = a.Cout / BN;
    p.ntiles = ((a.M + BM - 1) / BM) * p.ntn;
    const int ablate = ec_tools_ablate();
    p.ablate = ablate;
    const bool ls = BN == 128 && ec_config().conv8_longseg;
    // (long segments: three stages; X3: two A chunks + three plane tiles)
    const size_t stages = ls ? (X3 ? (size_t)(2 * 256 + 3 * BN) * ROW_BYTES : 3 * (size_t)(BM + BN) * ROW_BYTES)
                             : 2 * (size_t)(BM + BN) * ROW_BYTES;
    const size_t epi = X3 ? (size_t)128 * (BN * 4 + 16) : (size_t)(POOL ? 64 : BM) * (BN * 2 + 16);
    const size_t lds = (stages > epi ? stages : epi) + 4096;   // + auxiliary area: LayerNorm row table (2 KB) / stamps (profiling)
    p.aux_off = (int)(lds - 4096);
    auto go = [&](auto kern) {
        static std::atomic<uint64_t> attr_done{0};
        if (auto attr_g_ = ec_attr_needed(attr_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)p.ntiles), dim3(512), lds, s, p);
    };
#ifdef EC_CONV8_PROFILE   // ablation / stamp instances (tools/ablate8.sh, tools/stamps8.py): build with -DEC_CONV8_PROFILE
    if constexpr (BN == 256 && KS == 3 && !POOL && !X3 && !S2) {
        switch (ablate) {
            case 1: go(conv_igemm8_kernel<BN, KS, POOL, 1>); break;
            case 2: go(conv_igemm8_kernel<BN, KS, POOL, 2>); break;
            case 3: go(conv_igemm8_kernel<BN, KS, POOL, 3>); break;
            case 4: go(conv_igemm8_kernel<BN, KS, POOL, 4>); break;
            case 5: go(conv_igemm8_kernel<BN, KS, POOL, 5>); break;
            case 6: go(conv_igemm8_kernel<BN, KS, POOL, 6>); break;
            case 7: go(conv_igemm8_kernel<BN, KS, POOL, 7>); break;
            case 16: go(conv_igemm8_kernel<BN, KS, POOL, 16>); break;
            case 32: go(conv_igemm8_kernel<BN, KS, POOL, 32>); break;
            case 48: go(conv_igemm8_kernel<BN, KS, POOL, 48>); break;
            case 64: go(conv_igemm8_kernel<BN, KS, POOL, 64>); break;
            case 8: go(conv_igemm8_kernel<BN, KS, POOL, 8>); break;
            case 128: go(conv_igemm8_kernel<BN, KS, POOL, 128>); break;
            case 40: go(conv_igemm8_kernel<BN, KS, POOL, 40>); break;
            case 160: go(conv_igemm8_kernel<BN, KS, POOL, 160>); break;
            case 96: go(conv_igemm8_kernel<BN, KS, POOL, 96>); break;
            default: go(conv_igemm8_kernel<BN, KS, POOL, 0>);
        }
        EC_CHECK_LAUNCH();
        return EC_OK;
    }
#endif
    if constexpr (BN == 128) {
        if (ls) {
            go(conv_igemm8_kernel<BN, KS, POOL, 512, X3, S2, BM, XP>);
            EC_CHECK_LAUNCH();
            return EC_OK;
        }
    }
    go(conv_igemm8_kernel<BN, KS, POOL, 0, X3, S2, BM, XP>);
    EC_CHECK_LAUNCH();
    return EC_OK;
}


template <int KS, bool POOL>
int dispatch_tile(const ConvArgs& a, hipStream_t s) {
    // Tile choice: 128x128 wherever Cout allows; 256-row tiles for the narrow early layers.
    // (Fatter 128x256 / 256x128 tiles were measured: no gain, and they spill once loads run two tiles ahead.)
    // EC_CONV_BIG: 0 off; 1 (default) the 8-wave ping-pong kernel (conv_igemm8) where it was measured to win; 4 conv_igemm8
    // wherever its preconditions hold (tests / A-B).  Measured and removed again: the plain
    // double-buffered 256x256 configuration of conv_igemm_kernel (8 waves, one barrier per K-tile: 77.9 us where the
    // ping-pong schedule takes 72-75) and a one-wave-per-SIMD 4-wave kernel with in-wave software pipelining (87.8 us:
    // every LDS-DMA piece blocks its issuing wave for ~150 clk and there is no partner wave to keep the matrix pipe busy).
    const int big = ec_config().conv_big;
    if (big == 1 && a.Cin % 64 == 0 && (KS == 1 || a.cin_log2 >= 0) && a.K >= 512 && a.M % (POOL ? 4 : 1) == 0) {
        // measured (B = 256, tools/bench_big.sh): wins on the 3x3 convs with Cout % 256 == 0 once there are enough
        // 256-row tiles to occupy most CUs; loses on N = 128, on the short launches of 7x7 maps and ties on 1x1
        const long nt256 = (long)((a.M + 255) / 256) * (a.Cout / 256), nt128 = (long)((a.M + 255) / 256) * (a.Cout / 128);
        // fewest 256-row tiles for the 8-wave kernel: 150 for a launch that has the chip to itself; a caller that keeps two
        // launches in flight (the engine's two slices) lowers it on its encoder handles (ec_rn50_set_conv8_min_tiles): alone a 50-100-tile launch of the
        // 8-wave kernel is 30-40 % slower than the 4-wave kernel, but it leaves the other launch 150-200 whole CUs instead of
        // sharing all of them (same-box A/B at 2 x 128 frames: +0.4..1.5 % RN50, +3.5 % ViT-B/32 end to end; at 2 x 64: -1.1 %)
        const long mint_env = ec_config().conv8_min_tiles;
        const long mint = mint_env > 0 ? mint_env : (long)ec_tls_conv8_min_tiles;
        // 3x3 launches whose 256-wide tiles would fill only a fraction of the chip take 128-wide ones (twice the tiles; with
        // long segments, EC_CONV8_LONGSEG, a 128-wide K-tile costs half a 256-wide one): EC_CONV8_LOWFILL = tile-count limit
        if (KS == 3 && !POOL && a.Cout % 256 == 0 && a.K >= 2304 && nt256 >= mint && nt256 < EC_CONV8_LOWFILL && nt128 >= mint)
            return launch8<128, KS, POOL>(a, s);
        if (KS == 3 && a.Cout % 256 == 0 && nt256 >= mint) return launch8<256, KS, POOL>(a, s);
        // 3x3 convs with too few 256-wide tiles (layer 4 @7x7 in a single 256-frame launch: 98) but enough 128-wide ones:
        // 196 tiles x 72 K-tiles, 84.7 -> 64.6 us (tools/bench_shapes.py, B = 256)
        if (KS == 3 && !POOL && a.Cout % 256 == 0 && nt256 < mint && nt128 >= mint && a.K >= 2304) return launch8<128, KS, POOL>(a, s);
        // 128-channel 3x3 convs (layer 2): 128-wide tiles of the 8-wave kernel (same-box A/B in the engine, round 3:
        // +0.4..0.8 % at 2 x 128 frames, neutral at 2 x 32; alone 52 -> 42 us at 128 frames).  EC_CONV8_BN128 = -1: off
        if (KS == 3 && a.Cout == 128 && ec_config().conv8_bn128 >= 0 && nt128 >= 2 * mint) return launch8<128, KS, POOL>(a, s);
        // long-K 1x1 convs (tools/bench_l4.sh, B = 256): 1024->2048 @7x7 83.7 -> 70.5 us, 1024->512 @14x14 82.9 -> 70.5,
        // 1024->256 @14x14 40.6 -> 37.0 with 256-wide tiles; 2048->512 @7x7 46.4 -> 40.1 with 128-wide tiles (196 of them);
        // K = 512 and residual launches stay on the 4-wave kernel (slower here)
        // K = 512 and short-K residual launches stay on the 4-wave kernel (slower here).  ViT-B/32 GEMMs (tools/bench_vitgemm.sh,
        // 12800 tokens): in_proj 768->2304 78.6 -> 57.6 us, c_fc 768->3072 105 -> 82, c_proj 3072->768 + residual 91.7 -> 85.1;
        // out_proj 768->768 + residual is slower (31 -> 35) and keeps the 4-wave kernel
        // long-K 1x1 launches that would fill less than ~40 % of the chip with 256-wide tiles take 128-wide ones (twice the
        // tiles, same per-tile efficiency): ViT-B/32 c_proj at 6,400 tokens is 75 tiles x 48 K-tiles (same-box A/B: +2.4 % on the ViT config, +0.4 % RN50)
        if (KS == 1 && !POOL && ec_config().conv8_bn128 >= 0 && a.K >= EC_CONV8_LOWFILL_K && a.Cout % 256 == 0 && nt256 >= mint && nt256 < 100 &&
            nt128 >= mint)
            return launch8<128, KS, POOL>(a, s);
        // residual 1x1 launches with K = 512 .. 2047 that would fill less than ~40 % of the chip with 256-wide tiles (ViT-B/32
        // out_proj at 6,400 tokens: 75 tiles x 12 K-tiles) on the long-segment 128-wide tiles: same-box A/B 74.9 -> 76.3 k
        // env-frames/s on the ViT config, forward 2.10 -> 2.05 ms; with the limit at 150 tiles the rule also caught layer 4's
        // conv3 at 2 x 64 frames (104 tiles) and cost the RN50 config 0.4-1.3 % there.  EC_CONV8_RES128 = 0: off
        if (KS == 1 && !POOL && a.res && a.K >= 512 && a.K < 2048 && a.Cout % 128 == 0 && nt256 < 100 && nt128 >= mint)
            return launch8<128, KS, POOL>(a, s);
        if (KS == 1 && !POOL && ((!a.res && a.K >= 768) || (a.res && (a.K >= 2048 || (a.K >= 512 && nt256 >= 150))))) {   // (residual, K = 512: 49.5 -> 45.7 us on 512 -> 2048 @7x7; ViT out_proj, 75 tiles x 12 K-tiles, stays on the 4-wave kernel: 19.8 vs 23.3 us)
            if (a.Cout % 256 == 0 && nt256 >= mint) return launch8<256, KS, POOL>(a, s);
            if (!a.res && a.K >= 1024 && a.Cout % 128 == 0 && nt256 < mint && nt128 >= mint) return launch8<128, KS, POOL>(a, s);
        }
    }
    if (big == 4 && a.Cin % 64 == 0 && (KS == 1 || a.cin_log2 >= 0) && a.K >= 512 && a.M >= 256 * 32 && a.M % (POOL ? 4 : 1) == 0) {
        const int bn128 = ec_config().conv8_bn128;
        if (a.Cout % 256 == 0 && !bn128) return launch8<256, KS, POOL>(a, s);     // (A/B: everywhere its preconditions hold)
        if (a.Cout % 128 == 0) return launch8<128, KS, POOL>(a, s);
    }
    if (a.Cout % 128 == 0) {
        // 196-of-224-row tiles: 196 = 14^2 divides every RN50 feature map (56^2, 28^2, 14^2, 4 x 7^2), so the tile count
        // becomes a multiple of the frame count -- e.g. layer 3 at 256 frames: 512 tiles = 2 per CU instead of 784
        // tiles on 768 workgroups (16 tail tiles running alone).  Costs 12.5 % padding and runs 2 workgroups per CU, so
        // it only pays where quantisation hurts most: measured 3x3 256->256 @14x14 100.9 -> 79.3 us, 1x1 1024->256
        // 49.6 -> 39.2 us (B = 256); slower on short-K / residual convs and on 256-tile launches.
        // In the two-stream engine (128-frame launches) the other stream's kernels already fill the tail, and there the
        // padded tiles cost ~0.7 % (measured on the 28x28 3x3 convs, which also have 512 padded tiles at 128 frames):
        // the rule is therefore limited to 14x14 maps, i.e. to single launches of 256 frames.
        // EC_CONV_T224: 0 off, 1 everywhere (tests/experiments), 2 (default) the rule below, 3 also 256-tile launches.
        const int t224 = ec_config().conv_t224;
        if constexpr (!POOL) {
            const long t196 = (long)(a.M / 196) * (a.Cout / 128), t128 = (long)((a.M + 127) / 128) * (a.Cout / 128);
            if (a.M % 196 == 0 &&
                (t224 == 1 || (t224 >= 2 && !a.res && a.K >= 1024 && (t196 == 256 || t196 == 512) && (t128 % 768) != 0 &&
                               (t224 == 3 || (t196 == 512 && (long)a.H * a.W == 196)))))
                return launch<224, 128, 1, 4, KS, POOL, false, 196>(a, s);
        }
        // Small launches (strong scaling: 32-64 frames per GPU): with fewer 128x128 tiles than CUs every workgroup is one
        // long serial K chain and most of the chip idles -> 64x64 tiles give 4x the workgroups for the long-K layers
        // (batch 32: layer-4 3x3 52 tiles x 72 K-tiles = 76 us).  EC_CONV_T64: tile-count threshold (0 = off).
        const int t64 = ec_config().conv_t64;
        // EC_CONV_RING (default 1): those launches, and 128x128 launches with at most one workgroup per CU, run the
        // multi-stage ring pipeline (conv_igemm_kernel NS >= 3): with so few waves per CU nothing hides the L2 round trip
        // of the single-stage loop.  2: every non-prefetching 128x128 launch (A/B).
        const int ring = ec_config().conv_ring;
        if constexpr (!POOL) {
            const long t128 = (long)((a.M + 127) / 128) * (a.Cout / 128);
            if (t128 < t64 && a.K >= 512 && a.Cout % 64 == 0)
                // (deeper rings -- 6 / 8 stages for launches with at most one workgroup per CU, 4 stages for the 128x128 ring --
                //  measured round 3: 0.944 -> 0.95-0.96 ms at 32 frames, 1.50 -> 1.50-1.51 at 64: stages in flight are not
                //  what bounds these launches any more; what is left per K-tile is barrier + piece issue)
                return ring ? launch<64, 64, 2, 2, KS, POOL, false, 64, 4, true>(a, s) : launch<64, 64, 2, 2, KS, POOL>(a, s);
        }
        {
            const long t128 = (long)((a.M + 127) / 128) * (a.Cout / 128);
            const bool pf = (KS == 1 && !POOL && a.res && a.K <= 256);
            if (!pf && a.K >= 512 && ((ring == 1 && t128 <= 256) || ring == 2)) {
                // the one-workgroup-per-CU ring launches run on 8 waves (2 x 4, wave tile 64 x 32): two waves per SIMD, so one
                // wave's fragment reads / piece issue run beside the other's MFMAs; pieces interleaved with the MFMAs (ILV)
                return launch<128, 128, 2, 4, KS, POOL, false, 128, 3, true>(a, s);
            }
        }
        // residual register prefetch only for the short-K, bandwidth-bound expanding 1x1 convs
        if (KS == 1 && !POOL && a.res && a.K <= 256) return launch<128, 128, 2, 2, KS, POOL, (KS == 1 && !POOL)>(a, s);
        return launch<128, 128, 2, 2, KS, POOL>(a, s);
    }
    if (a.Cout % 64 == 0) return launch<256, 64, 4, 1, KS, POOL>(a, s);
    if (a.Cout % 32 == 0) return launch<256, 32, 4, 1, KS, POOL>(a, s);   // (one stem layer; 32 B rows < 8-wave loader pass)
    return EC_ERR_SHAPE;
}

}  // namespace

// conv3x3_narrow.hip: resident-weight kernel for the narrow early 3x3 layers (EC_ERR_SHAPE = not handled)
int ec_conv3x3_narrow(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                      int pool, hipStream_t s);

// conv_pair.hip: register-weight kernel for a few bandwidth-bound 1x1 shapes (EC_ERR_SHAPE = not handled)
int ec_conv1x1_regw(const void* a, const void* w, const float* bias, const void* res, void* y, long M, int K, int N, int act,
                    hipStream_t s);

#ifdef EC_TOOLS   // tools-only build (`make tools`): the stamp read-back of the 8-wave kernel's profiling instances
extern "C" int ec_debug_stamps(unsigned long long* host_dst, int n) {
    if (!host_dst || n <= 0 || n > 2048) return EC_ERR_ARG;
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ec_dbg_stamps), (size_t)n * 8) == hipSuccess ? EC_OK : EC_ERR_LAUNCH;
}
#endif

extern "C" int ec_conv_bf16(const void* in, const void* w, const float* bias, const void* res, void* out, int B,
                            int H, int W, int Cin, int Cout, int ksize, int pool, int act, ec_stream_t stream) {
    return ec_conv_bf16_ld(in, w, bias, res, out, B, H, W, Cin, Cout, ksize, pool, act, Cout, stream);
}

// ec_conv_bf16 writing a COLUMN BLOCK of a wider tensor: output row m starts at out + m * out_row_stride (elements).  What the
// trunk uses to lay a Bottleneck's pooled conv2 output and its pooled block input side by side, so that conv3 and the
// downsample conv become ONE GEMM over the concatenated K axis (rn50.hip).
extern "C" int ec_conv_bf16_ld(const void* in, const void* w, const float* bias, const void* res, void* out, int B,
                               int H, int W, int Cin, int Cout, int ksize, int pool, int act, int out_row_stride,
                               ec_stream_t stream) {
    if (!in || !w || !out) return EC_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0) return EC_ERR_SHAPE;
    if (out_row_stride < Cout || (out_row_stride & 7)) return EC_ERR_SHAPE;
    const bool dense = out_row_stride == Cout;
    if (ksize != 1 && ksize != 3) return EC_ERR_SHAPE;
    if (Cin < 8 || Cin % 8 != 0 || Cout % 32 != 0) return EC_ERR_SHAPE;
    if (pool && ((H & 1) || (W & 1) || res != nullptr || act != EC_ACT_RELU)) return EC_ERR_SHAPE;
    if (res && act == EC_ACT_QUICKGELU) return EC_ERR_UNSUPPORTED;   // (no caller: CLIP applies QuickGELU to c_fc only)
    if (H >= 4096 || W >= 65536) return EC_ERR_SHAPE;
    if ((long)B * H * W >= (1L << 31) / 4) return EC_ERR_SHAPE;
    ConvArgs a;
    a.in = (const uint16_t*)in;
    a.w = (const uint16_t*)w;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.out = (uint16_t*)out;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldo = Cout;
    a.K = ksize * ksize * Cin;
    a.M = B * H * W;
    a.cin_log2 = (Cin & (Cin - 1)) == 0 ? ec_ilog2(Cin) : -1;
    a.act = act;
    a.ntn = 0;
    if ((long)B * H * W * Cin * 2 >= (1L << 31) || (long)Cout * a.K * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    a.in_bytes = (unsigned)((long)B * H * W * Cin * 2);
    a.w_bytes = (unsigned)((long)Cout * a.K * 2);
    if (res && (long)B * H * W * Cout * 2 >= (1L << 32) - 16) return EC_ERR_SHAPE;
    a.res_bytes = res ? (unsigned)((long)B * H * W * Cout * 2) : 0u;
    a.ldo = out_row_stride;
    hipStream_t s = (hipStream_t)stream;
    // (the register-weight and resident-weight kernels write dense tensors only)
    if (dense && ksize == 1 && !pool && ec_conv1x1_regw(in, w, bias, res, out, (long)B * H * W, Cin, Cout, act, s) == EC_OK) return EC_OK;
    if (dense && ksize == 3 && !res && act == EC_ACT_RELU && Cin <= 64 && Cout <= 64 &&
        ec_conv3x3_narrow(in, w, bias, out, B, H, W, Cin, Cout, pool, s) == EC_OK)
        return EC_OK;
    if (ksize == 3) return pool ? dispatch_tile<3, true>(a, s) : dispatch_tile<3, false>(a, s);
    return pool ? dispatch_tile<1, true>(a, s) : dispatch_tile<1, false>(a, s);
}

// Stride-2 convolution (1x1, or 3x3 pad 1) + folded BatchNorm + residual + activation: torchvision's ResNet v1.5 strides inside
// the Bottleneck's 3x3 conv and in the 1x1 downsample conv (the `resnet_model` of
// primitive_probing/generate_data/thor_image_features.py:46-49), where CLIP's ModifiedResNet pools instead.  Same implicit
// GEMM: only the row decode differs (row m = output pixel, centre tap at input pixel (2 yo, 2 xo)).
extern "C" int ec_conv_bf16_s2(const void* in, const void* w, const float* bias, const void* res, void* out, int B, int H, int W,
                               int Cin, int Cout, int ksize, int act, ec_stream_t stream) {
    if (!in || !w || !out) return EC_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return EC_ERR_SHAPE;
    if (ksize != 1 && ksize != 3) return EC_ERR_SHAPE;
    if (Cin < 8 || Cin % 8 != 0 || Cout % 64 != 0) return EC_ERR_SHAPE;
    if (act != EC_ACT_RELU && act != EC_ACT_NONE) return EC_ERR_UNSUPPORTED;
    if (H >= 4096 || W >= 65536) return EC_ERR_SHAPE;
    const int Ho = H / 2, Wo = W / 2;
    ConvArgs a;
    a.in = (const uint16_t*)in;
    a.w = (const uint16_t*)w;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.out = (uint16_t*)out;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldo = Cout;
    a.K = ksize * ksize * Cin;
    a.M = B * Ho * Wo;
    a.cin_log2 = (Cin & (Cin - 1)) == 0 ? ec_ilog2(Cin) : -1;
    a.act = act;
    a.ntn = 0;
    if ((long)B * H * W * Cin * 2 >= (1L << 31) || (long)Cout * a.K * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    a.in_bytes = (unsigned)((long)B * H * W * Cin * 2);
    a.w_bytes = (unsigned)((long)Cout * a.K * 2);
    if (res && (long)B * Ho * Wo * Cout * 2 >= (1L << 32) - 16) return EC_ERR_SHAPE;
    a.res_bytes = res ? (unsigned)((long)B * Ho * Wo * Cout * 2) : 0u;
    hipStream_t s = (hipStream_t)stream;
    // the 8-wave ping-pong kernel where the CLIP trunk's rule would take it (Cin % 64 == 0, K >= 512, enough tiles to fill the chip):
    // 256-wide tiles from 150 tiles on, 128-wide ones when only those reach 150
    if (Cin % 64 == 0 && a.K >= 512 && (ksize == 1 || a.cin_log2 >= 0) && ec_config().conv_big != 0) {
        const long rt = (a.M + 255) / 256, nt256 = rt * (Cout / 256), nt128 = rt * (Cout / 128);
        if (Cout % 256 == 0 && nt256 >= EC_CONV8_MIN_TILES_DEFAULT)
            return ksize == 3 ? launch8<256, 3, false, false, true>(a, s) : launch8<256, 1, false, false, true>(a, s);
        if (Cout % 128 == 0 && nt128 >= EC_CONV8_MIN_TILES_DEFAULT && (ksize == 3 || !res))
            return ksize == 3 ? launch8<128, 3, false, false, true>(a, s) : launch8<128, 1, false, false, true>(a, s);
    }
    // 128 x 128 tiles (three workgroups per CU) where Cout allows, 64 x 64 ring tiles for launches that would leave CUs idle
    if (Cout % 128 == 0) {
        const long t128 = (long)((a.M + 127) / 128) * (Cout / 128);
        if (t128 < ec_config().conv_t64 && a.K >= 512)       // (the CLIP trunk's rule: dispatch_tile)
            return ksize == 3 ? launch<64, 64, 2, 2, 3, false, false, 64, 4, false, true>(a, s)
                              : launch<64, 64, 2, 2, 1, false, false, 64, 4, false, true>(a, s);
        if (t128 <= 256 && a.K >= 512 && !(ksize == 1 && res && a.K <= 256))   // one workgroup per CU: 8-wave ring tiles, pieces between the MFMAs
            return ksize == 3 ? launch<128, 128, 2, 4, 3, false, false, 128, 3, true, true>(a, s)
                              : launch<128, 128, 2, 4, 1, false, false, 128, 3, true, true>(a, s);
        return ksize == 3 ? launch<128, 128, 2, 2, 3, false, false, 128, 0, false, true>(a, s)
                          : launch<128, 128, 2, 2, 1, false, false, 128, 0, false, true>(a, s);
    }
    return ksize == 3 ? launch<256, 64, 4, 1, 3, false, false, 256, 0, false, true>(a, s)
                      : launch<256, 64, 4, 1, 1, false, false, 256, 0, false, true>(a, s);
}

// out[M, N] (fp32) = act(A[M, K] (bf16) @ W^T + bias) with W given as three bf16 planes [N][3][K] (ec_split3_bf16):
// the exact-fp32 product of a bf16 activation matrix with an fp32 weight matrix, on the 8-wave ping-pong kernel.
// Replaces the resnet_compressor's first 1x1 conv over the stored features
// (allenact_plugins/.../resnet_tensor... ResnetTensorGoalEncoder.resnet_compressor[0], SURVEY.md section 8 row a).
int ec_gemm_bf16a_xp(const void* A, const void* Wplanes, const float* bias, float* out, long M, int N, int K, int act, int planes,
                     ec_stream_t stream);
extern "C" int ec_gemm_bf16a_x3(const void* A, const void* Wplanes, const float* bias, float* out, long M, int N, int K,
                                int act, ec_stream_t stream) {
    return ec_gemm_bf16a_xp(A, Wplanes, bias, out, M, N, K, act, 3, stream);
}
// ... walking only the `planes` (2 or 3) leading planes of W (not part of the C-ABI; policy.hip's learn pass)
int ec_gemm_bf16a_xp(const void* A, const void* Wplanes, const float* bias, float* out, long M, int N, int K, int act, int planes,
                     ec_stream_t stream) {
    if (!A || !Wplanes || !out || (planes != 2 && planes != 3)) return EC_ERR_ARG;
    if (M <= 0 || N % 128 != 0 || K % 64 != 0 || K < 64) return EC_ERR_SHAPE;
    if ((long)N * 3 * K * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    // 32-bit byte offsets inside a launch: rows are processed in slabs of < 4 GiB of A
    const long slab = (((1L << 32) - 1) / ((long)K * 2)) / 256 * 256;
    for (long r0 = 0; r0 < M; r0 += slab) {
        const long rows = (M - r0 < slab) ? M - r0 : slab;
        ConvArgs a;
        a.in = (const uint16_t*)A + r0 * K;
        a.w = (const uint16_t*)Wplanes;
        a.bias = bias;
        a.res = nullptr;
        a.out = reinterpret_cast<uint16_t*>(out + r0 * N);
        a.H = 1; a.W = (int)rows; a.Cin = K; a.Cout = N; a.ldo = N;
        a.K = K; a.M = (int)rows;
        a.cin_log2 = 0;
        a.act = act;
        a.ntn = 0;
        a.in_bytes = (unsigned)(rows * K * 2);
        a.w_bytes = (unsigned)((long)N * 3 * K * 2);
        a.res_bytes = 0u;
        int rc = planes == 3 ? launch8<128, 1, false, true>(a, (hipStream_t)stream)
                             : launch8<128, 1, false, true, false, 256, 2>(a, (hipStream_t)stream);
        if (rc != EC_OK) return rc;
    }
    return EC_OK;
}

// GEMM on the 8-wave kernel with LayerNorm folded in (vit.hip run_blocks; not part of the C-ABI):
//   consumer (ln_s != nullptr): out = act(LN(A) Wt^T + b) computed as rstd (A Wg^T - mean s) + c on the RAW rows A, with
//     Wt = W diag(gamma) (bf16), ln_s[n] = sum_k Wt[n, k], bias = c[n] = sum_k beta[k] W[n, k] + b[n], and the rows' statistics from
//     the ln_np partial records the producer of A left in ln_stats ([M][ln_np][4] floats: sum, M2, count, -);
//   producer (stats_out != nullptr): out = act(A Wt^T + bias (+ res)) and the tile's record of every output row into
//     stats_out[M][N / tile width][4]; *np_out = records per row.
// Replaces `x + attn(ln_1(x))` / `x + mlp(ln_2(x))`'s nn.LayerNorm launches ([U] clip/model.py ResidualAttentionBlock.forward).
int ec_gemm_bf16_ln8(const void* A, const void* Wt, const float* bias, const void* res, void* out, int M, int N, int K, int act,
                     const float* ln_s, const float* ln_stats, int ln_np, float* stats_out, int* np_out, ec_stream_t stream) {
    if (!A || !Wt || !out) return EC_ERR_ARG;
    if (M <= 0 || N % 128 != 0 || K % 64 != 0 || K < 64) return EC_ERR_SHAPE;
    if (res && act == EC_ACT_QUICKGELU) return EC_ERR_UNSUPPORTED;
    if (ln_s && (res || !bias || !ln_stats || ln_np < 1 || ln_np > 8)) return EC_ERR_ARG;   // (plain calls -- the patch embedding -- may come without a bias)
    ConvArgs a;
    a.in = (const uint16_t*)A;
    a.w = (const uint16_t*)Wt;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.out = (uint16_t*)out;
    a.H = 1; a.W = M; a.Cin = K; a.Cout = N; a.ldo = N;
    a.K = K; a.M = M;
    a.cin_log2 = 0;
    a.act = act;
    a.ntn = 0;
    if ((long)M * K * 2 >= (1L << 31) || (long)N * K * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    a.in_bytes = (unsigned)((long)M * K * 2);
    a.w_bytes = (unsigned)((long)N * K * 2);
    if (res && (long)M * N * 2 >= (1L << 32) - 16) return EC_ERR_SHAPE;
    a.res_bytes = res ? (unsigned)((long)M * N * 2) : 0u;
    a.ln_s = ln_s; a.ln_stats = ln_s ? ln_stats : nullptr; a.ln_np = ln_np; a.stats_out = stats_out;
    // Tile width: 256 where it divides N and leaves at least 100 tiles (QKV / c_fc at 6,400 rows: 225 / 300), 128-wide
    // long-segment tiles otherwise (out_proj / c_proj: N = D).  Measured and rejected (same box, round 6): choosing the width by
    // rounds of workgroups -- c_fc as 600 narrow tiles (three half rounds) instead of 300 wide ones (two rounds, the second
    // 17 % full) -- loses alone (1.905 -> 1.969 ms per 128 frames) and with two launches in flight (81.1 -> 79.2 k
    // env-frames/s): the 256-wide tile is the cheaper one per flop.  EC_VIT_WIDE: 0 = always 128-wide, 2 = 256-wide wherever N allows.
    const long rt = (M + 255) / 256;
    static const int wide_mode = [] { const char* e = getenv("EC_VIT_WIDE"); return e ? atoi(e) : 1; }();
    bool wide = (N % 256 == 0) && ((wide_mode == 1 && rt * (N / 256) >= 100) || wide_mode == 2);
    if (stats_out && N / 128 > 8 && N % 256 == 0) wide = true;        // (a producer's records must fit the consumer's 8 slots)
    if (stats_out && N / (wide ? 256 : 128) > 8) return EC_ERR_SHAPE;
    if (np_out) *np_out = N / (wide ? 256 : 128);
    if (wide) return launch8<256, 1, false>(a, (hipStream_t)stream);
    // 128-wide tiles: 192 rows instead of 256 when that costs fewer rounds-worth of work (a 192-row tile = 0.75 of a 256-row one):
    // N = 768 at 6,400 rows is 150 tiles (one round on 59 % of the CUs) against 204 (one round at 0.75).  Same box, round 6: the
    // forward alone 1.905 -> 1.80 ms per 128 frames; with two launches in flight (whose idle CUs the other launch fills) neutral
    // (81.1 / 80.6 -> 81.1 / 80.7 k env-frames/s).  EC_VIT_BM192 = 0: off
    static const int bm192 = [] { const char* e = getenv("EC_VIT_BM192"); return e ? atoi(e) : 1; }();
    const long t256 = rt * (N / 128), t192 = ((M + 191) / 192) * (long)(N / 128);
    if (bm192 && ((t192 + 255) / 256) * 3 < ((t256 + 255) / 256) * 4)
        return launch8<128, 1, false, false, false, 192>(a, (hipStream_t)stream);
    return launch8<128, 1, false>(a, (hipStream_t)stream);
}

extern "C" int ec_gemm_bf16(const void* A, const void* Wt, const float* bias, const void* res, void* out, int M, int N,
                            int K, int act, ec_stream_t stream) {
    if (!A || !Wt || !out) return EC_ERR_ARG;
    if (M <= 0 || N % 32 != 0 || K % 8 != 0 || K < 8) return EC_ERR_SHAPE;
    if (res && act == EC_ACT_QUICKGELU) return EC_ERR_UNSUPPORTED;
    // GEMM = 1x1 conv over a [1, 1, M] "image" with Cin = K (no power-of-two need: KS==1 never splits k).
    ConvArgs a;
    a.in = (const uint16_t*)A;
    a.w = (const uint16_t*)Wt;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.out = (uint16_t*)out;
    a.H = 1; a.W = M; a.Cin = K; a.Cout = N; a.ldo = N;
    a.K = K; a.M = M;
    a.cin_log2 = 0;
    a.act = act;
    a.ntn = 0;
    if ((long)M * K * 2 >= (1L << 31) || (long)N * K * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    a.in_bytes = (unsigned)((long)M * K * 2);
    a.w_bytes = (unsigned)((long)N * K * 2);
    if (res && (long)M * N * 2 >= (1L << 32) - 16) return EC_ERR_SHAPE;
    a.res_bytes = res ? (unsigned)((long)M * N * 2) : 0u;
    return dispatch_tile<1, false>(a, (hipStream_t)stream);
}
