// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the
// embodied-clip hot path.  HIP only -- no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/ec_amd.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define EC_WAVE 64

__device__ __forceinline__ uint16_t ec_f2bf(float f) {
    // round-to-nearest-even fp32 -> bf16 (finite inputs)
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float ec_bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __bf16 ec_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float ec_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ec_pack2(float lo, float hi) {
    // native conversion: one v_cvt_pk_bf16_f32 (round-to-nearest-even) for the pair
    ec_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ec_bf16x2_t));
}
__device__ __forceinline__ float ec_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float ec_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// fp32 -> three bf16 planes (leading bits, then two residuals): x == p0 + p1 + p2 to ~2^-24; four values at a time
__device__ __forceinline__ void ec_split3x4(const float (&x)[4], uint2& p0, uint2& p1, uint2& p2) {
    p0.x = ec_pack2(x[0], x[1]); p0.y = ec_pack2(x[2], x[3]);
    const float r0 = x[0] - ec_lo(p0.x), r1 = x[1] - ec_hi(p0.x), r2 = x[2] - ec_lo(p0.y), r3 = x[3] - ec_hi(p0.y);
    p1.x = ec_pack2(r0, r1); p1.y = ec_pack2(r2, r3);
    p2.x = ec_pack2(r0 - ec_lo(p1.x), r1 - ec_hi(p1.x)); p2.y = ec_pack2(r2 - ec_lo(p1.y), r3 - ec_hi(p1.y));
}

// CategoricalDistr.sample() + log_prob() of one row of logits ([U] allenact base_abstractions/distributions.py): inverse CDF on a
// counter-based uniform keyed by (seed, step, GLOBAL actor id).  Shared by sample_kernel (ppo.hip) and the act step's heads
// launch (policy.hip: ec_policy_act) -- the same arithmetic, so the two routes sample identical actions.
__device__ __forceinline__ uint64_t ec_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class RowFn>
__device__ __forceinline__ void ec_sample_row(RowFn row, int A, uint64_t seed, uint64_t step, uint64_t global_actor, int& a_out,
                                              float& logp_out) {
    float mx = -INFINITY;
    for (int k = 0; k < A; ++k) mx = fmaxf(mx, row(k));
    float se = 0.f;
    for (int k = 0; k < A; ++k) se += expf(row(k) - mx);
    const float lse = mx + logf(se);
    const uint64_t h = ec_mix64(ec_mix64(seed) ^ (step * 0x100000001B3ull + global_actor));
    const float u = (float)((h >> 40) * (1.0 / 16777216.0));   // [0,1) with 24 bits
    float cdf = 0.f;
    int a = A - 1;
    for (int k = 0; k < A; ++k) {
        cdf += expf(row(k) - lse);
        if (u < cdf) { a = k; break; }
    }
    a_out = a;
    logp_out = row(a) - lse;
}

// Bijective XCD-aware remap of a linear block id (cdna guide T1): the
// dispatcher places block b on XCD b % 8; give each XCD a contiguous chunk of
// the logical tile space so neighbouring tiles share an L2.
__device__ __forceinline__ unsigned ec_xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    if (nwg < nx * 2) return bid;
    unsigned xcd = bid % nx, q = nwg / nx, r = nwg % nx;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / nx;
}

static inline int ec_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: one flag bit per device ordinal.  Usage:
//     if (auto g = ec_attr_needed(done)) { hipFuncSetAttribute(...); }
// The guard publishes the device's bit when it goes out of scope, i.e. AFTER the attribute call has returned: a second
// thread on the same device either sees the bit (the attribute is in force) or sets the attribute itself (idempotent) --
// it can never skip the call and launch with the old LDS limit.
struct ec_attr_guard {
    std::atomic<uint64_t>* a;
    uint64_t bit;
    explicit operator bool() const { return a != nullptr; }
    ec_attr_guard(std::atomic<uint64_t>* a_, uint64_t b_) : a(a_), bit(b_) {}
    ec_attr_guard(const ec_attr_guard&) = delete;
    ec_attr_guard& operator=(const ec_attr_guard&) = delete;
    ~ec_attr_guard() { if (a) a->fetch_or(bit, std::memory_order_release); }
};
static inline ec_attr_guard ec_attr_needed(std::atomic<uint64_t>& done) {
    int d = 0;
    (void)hipGetDevice(&d);
    const uint64_t bit = 1ull << (d & 63);
    if (done.load(std::memory_order_acquire) & bit) return ec_attr_guard(nullptr, 0);
    return ec_attr_guard(&done, bit);
}

// Fewest 256-row output tiles for which the conv / GEMM dispatch takes the 8-wave kernel (conv_igemm.hip dispatch_tile).
// A property of the encoder HANDLE (ec_rn50_set_conv8_min_tiles / ec_vit_set_conv8_min_tiles), in force for the calling
// thread while that handle's forward issues its launches; everything else sees the default.
constexpr int EC_CONV8_MIN_TILES_DEFAULT = 150;
extern thread_local int ec_tls_conv8_min_tiles;
struct ec_min_tiles_scope {
    int prev;
    explicit ec_min_tiles_scope(int n) : prev(ec_tls_conv8_min_tiles) { ec_tls_conv8_min_tiles = n > 0 ? n : EC_CONV8_MIN_TILES_DEFAULT; }
    ec_min_tiles_scope(const ec_min_tiles_scope&) = delete;
    ~ec_min_tiles_scope() { ec_tls_conv8_min_tiles = prev; }
};

// Once-per-workgroup staging loops (weights -> LDS).  Written as `for (idx = tid; ...) lds[f(idx)] = global[g(idx)]` hipcc
// emits load -> s_waitcnt vmcnt(0) -> ds_write per iteration: TOTAL / NT serial L2 round trips at the head of EVERY
// launch (18 of them, ~15 us, in the narrow 3x3 kernels -- round-3 EC_ROWS_DBG ablation).  Here all of a thread's loads
// are issued before its first store (compile-time trip count, fully unrolled; the ragged last pass re-reads the last
// valid element and only its STORE is predicated, so no load sits behind an exec-masked branch).
template <int TOTAL, int NT, class T, class LoadFn, class StoreFn>
__device__ __forceinline__ void ec_stage_all(int tid, LoadFn load, StoreFn store) {
    constexpr int IT = (TOTAL + NT - 1) / NT;
    T v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int idx = tid + i * NT;
        if (TOTAL % NT != 0) idx = idx < TOTAL ? idx : TOTAL - 1;
        v[i] = load(idx);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NT;
        if (TOTAL % NT == 0 || idx < TOTAL) store(idx, v[i]);
    }
}

// ---- library configuration -------------------------------------------------------------------------------------------
// Every tuning / experiment switch of the library, read from the environment ONCE (first use) into one immutable
// struct; the dispatch functions read fields of ec_config() instead of keeping getenv-initialised statics each.
// ec_config_hash() goes into ec_rn50_plan_hash / ec_vit_plan_hash, so a profile under profiles/ names the switch
// settings it was measured with.  DESIGN.md section 5.1 documents each variable.
struct EcConfig {
    // --- encoder: selection between ADOPTED kernel paths (each one is exercised by a parity test) ---
    int conv_narrow;      // EC_CONV_NARROW   (3)   narrow 3x3 kernels: 0 off, 1 gather (32 ch), 2 gather (also 64 ch), 3 row tiles
    int conv_rowsn;       // EC_CONV_ROWSN    (1)   multi-row tiles for the un-pooled narrow 3x3 layers (0: single-row tiles)
    int conv_big;         // EC_CONV_BIG      (1)   0 no conv_igemm8, 1 where measured faster, 4 wherever it applies (tests)
    long conv8_min_tiles; // EC_CONV8_MIN_TILES (0) overrides the handles' dispatch threshold when > 0
    int conv8_bn128;      // EC_CONV8_BN128   (0)   1 with EC_CONV_BIG=4: force 128-wide tiles; -1: layer-2 3x3 convs stay on the 4-wave kernel
    int conv8_longseg;    // EC_CONV8_LONGSEG (1)   conv_igemm8, 128-wide tiles: two segments per K-tile, three LDS stages (0: four segments)
    int conv_t224;        // EC_CONV_T224     (2)   196-of-224-row tiles: 0 off, 1 everywhere, 2 rule, 3 also 256-tile launches
    int conv_t64;         // EC_CONV_T64      (150) launches with fewer 128x128 tiles use 64x64 tiles
    int conv_ring;        // EC_CONV_RING     (1)   ring pipeline for launches with <= 1-2 workgroups per CU (0: single-stage loop)
    int conv_regw;        // EC_CONV_REGW     (1)   register-weight 1x1 kernel
    int rn50_fuse;        // EC_RN50_FUSE     (1)   fused layer-1 / layer-2 block boundaries in the trunk plan
    int rn50_bneck;       // EC_RN50_BNECK    (128) launches of at least this many frames run layer3.1-5 as fused bottleneck launches (conv_bneck.hip); 0: never
    int rn50_bneck3;      // EC_RN50_BNECK3   (1)   the fused bottleneck launches include conv1 (the whole block in one launch)
    int rn50_img3;        // EC_RN50_IMG3     (1)   small launches run the 14x14x256 (<= 32 frames) / 7x7x512 (<= 64 frames) 3x3 convs on the image-resident K-split kernel
    int rn50_dscat;       // EC_RN50_DSCAT    (1)   stride-2 Bottlenecks of layers 3-4: conv3 and the downsample conv as ONE GEMM over the concatenated K axis (pooled conv2 output | pooled block input)
    int rn50_poolout;     // EC_RN50_POOLOUT  (1)   layer 2's last conv3 also emits AvgPool2d(2) of its output (the pooled block input of layer3.0's K-concatenated GEMM): no pooling pass
    // --- policy / update ---
    int gemm_no_x3;       // EC_GEMM_NO_X3    (0)   policy GEMMs on the fp32 MFMA instead of bf16x3
    int gemm_bwd3;        // EC_GEMM_BWD3     (0)   ec_policy_backward's large gradient GEMMs on three of the six bf16x3 products (also on under EC_POLICY_FAST)
    int policy_fast;      // EC_POLICY_FAST   (1)   learn pass: the backward's large gradient GEMMs on the three leading bf16x3 products, the compressor conv over the
                          //                        stored features and its weight gradient on the two leading planes of the fp32 operand (16 mantissa bits; products
                          //                        accurate to 2^-16 .. 2^-17 -- finer than the TF32 the reference's torch 1.7 uses for fp32 matmuls on Ampere);
                          //                        0: every policy GEMM fp32-exact (six products / three planes)
    int act_split;        // EC_ACT_SPLIT     (1)   act step: K-split kernels for its two long-K GEMMs
    int tail_fused;       // EC_TAIL_FUSED    (1)   compressor tail / combiner fused kernels
    int gru_fused;        // EC_GRU_FUSED     (2)   0 GEMM + gate kernels, 1 fused 32x32-tile step kernels, 2 + 16x16-tile kernels in the update
    int c1_pingpong;      // EC_C1_PINGPONG   (1)   compressor conv 1 over stored features on the 8-wave kernel
    int dw1_tr;           // EC_DW1_TR        (1)   dW1 on the transpose-read kernel
    int wih_perm;         // EC_WIH_PERM      (1)   learn pass: re-ordered weight_ih instead of activation transposes
    int dw_transposed;    // EC_DW_TRANSPOSED (1)   GRU weight-gradient GEMMs on transposed (K-contiguous) operands
};
// Fixed since round 5 (measured winners; the A/B numbers live in docs/experiments.md): 768 persistent workgroups, residual 1x1
// launches of K 512..2047 with < 100 256-wide tiles on 128-wide 8-wave tiles, low-fill limits 100 tiles / K >= 1024, ring-mode
// LDS-DMA pieces interleaved with the MFMAs, 128 x 128 ring tiles on 8 waves.
constexpr int EC_CONV8_LOWFILL = 100, EC_CONV8_LOWFILL_K = 1024;
const EcConfig& ec_config();          // api.hip
uint64_t ec_config_hash();            // FNV-1a over the fields above

#define EC_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return EC_ERR_LAUNCH;        \
    } while (0)
