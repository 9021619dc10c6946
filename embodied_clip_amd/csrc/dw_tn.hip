// Weight gradient of the policy's first compressor conv over the stored features (round 2):
//     dW1[128][C] += sum_tokens dc1[token][128]^T  feat[token][C]
// i.e. the "TN" GEMM behind `loss.backward()` for ResnetTensorGoalEncoder.resnet_compressor[0]
// (allenact_plugins/robothor_plugin/... resnet tensor encoders; SURVEY.md section 8 row a, HOT LOOP B), with
// K = every feature-map pixel of the slice (T*N*49 = 802 816 tokens per call in the headline configuration).
//
// Both operands are token-major (the contraction index is the SLOW index of both), which the matrix cores do not take
// directly: a 32x32x16 operand wants 8 consecutive k per lane.  gfx950's LDS transpose read does the turn for free:
//   * token rows go global -> LDS as they lie (LDS-DMA through buffer descriptors, 16-byte pieces, rows past the end of
//     the slice are out-of-range reads = zeros), 32 tokens per K-tile: feat [32][256 ch] (16 KB) and dc1 as its three bf16
//     planes [32][3][128 ch] (24 KB, written in that form by tail_bwd_kernel: the bf16x3 split of gemm_f32.hip);
//   * fragments come out of LDS with ds_read_b64_tr_b16 (a 16-lane group reads a [4 tokens][16 channels] block and each
//     lane receives one channel's 4 tokens): two reads per operand.  The 16-byte chunks of token row r sit XOR-ed with
//     (r & 3) << 2 (applied on the global side of the copy), so the four rows a half-wave reads fall into four
//     different 64-byte bank blocks: conflict-free.
// Workgroup: 8 waves, output tile 128 (dc1 channels) x 256 (feature channels), wave tile 64 x 64 (4 accumulator tiles);
// per K-tile and wave 32 transpose reads feed 24 MFMAs (2 k-steps x 2 x 2 tiles x 3 planes, smallest plane first).
// Three LDS stages, K-tile t+2 in flight while t is consumed, one barrier per K-tile.  The token range is split
// over token splits (8 column tiles x 32 splits = 256 workgroups, the column tiles of a split on one XCD); each writes its partial tile, and
// dw_reduce_kernel folds the splits into the gradient (deterministic: no atomics).
#include "common.h"

namespace {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

constexpr int KT = 32;                         // tokens per K-tile
constexpr int NYT = 128, NXT = 256;            // output tile: dY channels x X channels
constexpr int X_ROW = NXT * 2, Y_ROW = 3 * NYT * 2;       // bytes per token row in LDS: 512, 768
constexpr int X_BYTES = KT * X_ROW, Y_BYTES = KT * Y_ROW; // 16 KB, 24 KB
constexpr int STG = X_BYTES + Y_BYTES;         // 40 KB
constexpr int NSTG = 3;
constexpr int X_PIECES = X_BYTES / 1024 / 8, Y_PIECES = Y_BYTES / 1024 / 8;   // per wave: 2, 3

struct DwArgs {
    const uint16_t* yp;   // [M][3][128] bf16 planes of dY
    const uint16_t* x;    // [M][NX] bf16
    float* part;          // [nsplit][128][NX]
    long M;
    int NX;
    long tok_per_split;   // multiple of KT
    int ntile, nsplit;    // column tiles (NX / 256), token splits
};

// NPL: planes of dY that are multiplied -- 3 = the exact-fp32 product, 2 = the two leading planes (EC_POLICY_FAST; all three
// are still copied to LDS: the copy geometry is the plane layout's)
template <int NPL>
__global__ __launch_bounds__(512) void dw_tn_x3_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware work assignment: consecutive workgroup ids go to different XCDs (id % 8), each with its own L2.  The
    // column tiles of one token split read the SAME dY rows, so they are placed on one XCD (ids id, id + 8, ...):
    // dY then crosses HBM once per split instead of once per column tile (measured: 8.2 -> 2.3 GB per call).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int col = slot % p.ntile, split = xcd + 8 * (slot / p.ntile);
    if (split >= p.nsplit) return;
    const int n0 = col * NXT;
    const long tok0 = (long)split * p.tok_per_split;
    long ntok = p.M - tok0;
    if (ntok > p.tok_per_split) ntok = p.tok_per_split;
    if (ntok <= 0) ntok = 0;
    const int nk = (int)((ntok + KT - 1) / KT);

#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + tok0 * p.NX), 0,
                                                                           (int)(unsigned)(ntok * p.NX * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.yp + tok0 * (3 * NYT)), 0,
                                                                           (int)(unsigned)(ntok * Y_ROW), 0x00020000);
#endif
    // ---- loader geometry: LDS position P = piece * 1024 + lane * 16 holds chunk (c' ^ swizzle(r)) of token row r ----
    unsigned x_off[X_PIECES], y_off[Y_PIECES];
#pragma unroll
    for (int i = 0; i < X_PIECES; ++i) {
        const int P = (wave * X_PIECES + i) * 1024 + lane * 16;
        const int r = P / X_ROW, cs = (P % X_ROW) >> 4, c = cs ^ ((r & 3) << 2);
        x_off[i] = (unsigned)r * (unsigned)p.NX * 2u + (unsigned)n0 * 2u + (unsigned)c * 16u;
    }
#pragma unroll
    for (int i = 0; i < Y_PIECES; ++i) {
        const int P = (wave * Y_PIECES + i) * 1024 + lane * 16;
        const int r = P / Y_ROW, rem = P % Y_ROW, pl = rem >> 8, cs = (rem & 255) >> 4, c = cs ^ ((r & 3) << 2);
        y_off[i] = (unsigned)(r * Y_ROW + pl * 256 + c * 16);
    }
    const unsigned x_kstep = (unsigned)KT * (unsigned)p.NX * 2u;
    auto issue = [&](int kt) {
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned char* st = smem + (kt % NSTG) * STG;
#pragma unroll
        for (int i = 0; i < X_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(st + (wave * X_PIECES + i) * 1024), 16,
                                                     x_off[i] + (unsigned)kt * x_kstep, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < Y_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_t*)(st + X_BYTES + (wave * Y_PIECES + i) * 1024), 16,
                                                     y_off[i] + (unsigned)kt * (unsigned)Y_BYTES, 0, 0, 0);
#endif
    };

    // ---- fragment geometry (per lane, K-invariant) ----
    const int t = lane & 15, cb = (lane >> 4) & 1, khalf = lane >> 5;
    const int rr = khalf * 8 + (t >> 2);                  // token row inside a k-step's 16 (plus 4 for the second read)
    const int sw = ((t >> 2) & 3) << 2;                   // == (row & 3) << 2: the other row terms are multiples of 4
    int xb[2], yb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int chx = wn * 8 + j * 4 + cb * 2 + ((t & 3) >> 1);
        xb[j] = rr * X_ROW + ((chx ^ sw) << 4) + (t & 1) * 8;
        const int chy = wm * 8 + j * 4 + cb * 2 + ((t & 3) >> 1);
        yb[j] = X_BYTES + rr * Y_ROW + ((chy ^ sw) << 4) + (t & 1) * 8;
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment reads are inline asm: the compiler would otherwise order every LDS read after ALL outstanding LDS-DMA
    // (s_waitcnt vmcnt(0) right after the prefetch is issued), which serialises the pipeline.  LDS returns data in order,
    // so "lgkmcnt(4)" = everything but the newest group of four reads has arrived; the waits carry the fragment registers
    // as operands so that the MFMAs that consume them cannot be scheduled above the wait.
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void_t*)smem;
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define READ_X(ks, L0, H0, L1, H1)                                                                     \
    TR_READ(L0, ax0, (ks * 16) * X_ROW); TR_READ(H0, ax0, (ks * 16 + 4) * X_ROW);                      \
    TR_READ(L1, ax1, (ks * 16) * X_ROW); TR_READ(H1, ax1, (ks * 16 + 4) * X_ROW)
#define READ_Y(ks, pl, L0, H0, L1, H1)                                                                 \
    TR_READ(L0, ay0, (ks * 16) * Y_ROW + pl * 256); TR_READ(H0, ay0, (ks * 16 + 4) * Y_ROW + pl * 256); \
    TR_READ(L1, ay1, (ks * 16) * Y_ROW + pl * 256); TR_READ(H1, ay1, (ks * 16 + 4) * Y_ROW + pl * 256)
#define WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define WAIT0(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define MFMA4(yl0, yh0, yl1, yh1, xl0, xh0, xl1, xh1)                                                  \
    do {                                                                                               \
        const bf16x8_t y0 = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(yl0, yh0, 0, 1, 2, 3, 4, 5, 6, 7)); \
        const bf16x8_t y1 = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(yl1, yh1, 0, 1, 2, 3, 4, 5, 6, 7)); \
        const bf16x8_t x0 = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(xl0, xh0, 0, 1, 2, 3, 4, 5, 6, 7)); \
        const bf16x8_t x1 = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(xl1, xh1, 0, 1, 2, 3, 4, 5, 6, 7)); \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y0, x0, acc[0][0], 0, 0, 0);                \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y0, x1, acc[0][1], 0, 0, 0);                \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y1, x0, acc[1][0], 0, 0, 0);                \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y1, x1, acc[1][1], 0, 0, 0);                \
    } while (0)

    // (tiles past the end are issued too: out-of-range reads, zeros into a stage nobody reads -- keeps vmcnt uniform)
    issue(0);
    issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // this wave's pieces of K-tile kt have landed (kt+1's 5 may be in flight)
        __builtin_amdgcn_s_barrier();                      // ... everyone's have, and everyone is done reading stage (kt-1) % 3
        issue(kt + 2);
        const unsigned sb = lds0 + (unsigned)((kt % NSTG) * STG);
        const unsigned ax0 = sb + xb[0], ax1 = sb + xb[1], ay0 = sb + yb[0], ay1 = sb + yb[1];
        s16x4_t xa0, xa1, xa2, xa3, xc0, xc1, xc2, xc3;    // X fragments of k-step 0 / 1
        s16x4_t p0, p1, p2, p3, q0, q1, q2, q3;            // Y plane fragments, two sets in rotation
        if constexpr (NPL == 3) {
        READ_X(0, xa0, xa1, xa2, xa3);
        READ_Y(0, 2, p0, p1, p2, p3);
        WAIT4(xa0, xa1, xa2, xa3);
        READ_Y(0, 1, q0, q1, q2, q3);
        WAIT4(p0, p1, p2, p3);
        MFMA4(p0, p1, p2, p3, xa0, xa1, xa2, xa3);
        READ_Y(0, 0, p0, p1, p2, p3);
        WAIT4(q0, q1, q2, q3);
        MFMA4(q0, q1, q2, q3, xa0, xa1, xa2, xa3);
        READ_X(1, xc0, xc1, xc2, xc3);
        WAIT4(p0, p1, p2, p3);
        MFMA4(p0, p1, p2, p3, xa0, xa1, xa2, xa3);
        READ_Y(1, 2, q0, q1, q2, q3);
        WAIT4(xc0, xc1, xc2, xc3);
        READ_Y(1, 1, p0, p1, p2, p3);
        WAIT4(q0, q1, q2, q3);
        MFMA4(q0, q1, q2, q3, xc0, xc1, xc2, xc3);
        READ_Y(1, 0, q0, q1, q2, q3);
        WAIT4(p0, p1, p2, p3);
        MFMA4(p0, p1, p2, p3, xc0, xc1, xc2, xc3);
        WAIT0(q0, q1, q2, q3);
        MFMA4(q0, q1, q2, q3, xc0, xc1, xc2, xc3);
        } else {   // planes 1 and 0 only, lower one first (same read-ahead pattern: one group of four reads in flight)
        READ_X(0, xa0, xa1, xa2, xa3);
        READ_Y(0, 1, p0, p1, p2, p3);
        WAIT4(xa0, xa1, xa2, xa3);
        READ_Y(0, 0, q0, q1, q2, q3);
        WAIT4(p0, p1, p2, p3);
        MFMA4(p0, p1, p2, p3, xa0, xa1, xa2, xa3);
        READ_X(1, xc0, xc1, xc2, xc3);
        WAIT4(q0, q1, q2, q3);
        MFMA4(q0, q1, q2, q3, xa0, xa1, xa2, xa3);
        READ_Y(1, 1, p0, p1, p2, p3);
        WAIT4(xc0, xc1, xc2, xc3);
        READ_Y(1, 0, q0, q1, q2, q3);
        WAIT4(p0, p1, p2, p3);
        MFMA4(p0, p1, p2, p3, xc0, xc1, xc2, xc3);
        WAIT0(q0, q1, q2, q3);
        MFMA4(q0, q1, q2, q3, xc0, xc1, xc2, xc3);
        }
    }
#undef TR_READ
#undef READ_X
#undef READ_Y
#undef WAIT4
#undef WAIT0
#undef MFMA4
    // ---- partial tile out: C/D layout, column = lane & 31 (feature channel), rows = dY channels ----
    float* out = p.part + (long)split * NYT * p.NX;
    const int ocol = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                out[(long)m * p.NX + n0 + wn * 64 + j * 32 + ocol] = acc[i][j][r];
            }
}

__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ part, int nsplit, long n4, float* __restrict__ dW) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nsplit; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((long)k * n4 + q) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4 d = *reinterpret_cast<float4*>(dW + q * 4);
    d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
    *reinterpret_cast<float4*>(dW + q * 4) = d;
}

}  // namespace

// number of token splits ec_dw_tn_x3 will use for M tokens (the caller sizes `part` as nsplit * 128 * NX floats)
extern "C" int ec_dw_tn_x3_splits(long M, int NX) {
    if (M <= 0 || NX <= 0 || NX % NXT != 0) return 0;
    const int tiles = NX / NXT;
    int ns = (256 + tiles - 1) / tiles;                       // one workgroup per CU
    const long ktiles = (M + KT - 1) / KT;
    if (ns > ktiles) ns = (int)ktiles;
    return ns < 1 ? 1 : ns;
}

// dW[128][NX] += dY^T X with dY given as bf16 planes [M][3][128] (ec_split3_bf16 layout) and X bf16 [M][NX]
int ec_dw_tn_xp(const void* dYplanes, const void* X, float* part, float* dW, long M, int NX, int planes, ec_stream_t stream);
extern "C" int ec_dw_tn_x3(const void* dYplanes, const void* X, float* part, float* dW, long M, int NX, ec_stream_t stream) {
    return ec_dw_tn_xp(dYplanes, X, part, dW, M, NX, 3, stream);
}
// ... multiplying only the `planes` (2 or 3) leading planes of dY (not part of the C-ABI; policy.hip under EC_POLICY_FAST)
int ec_dw_tn_xp(const void* dYplanes, const void* X, float* part, float* dW, long M, int NX, int planes, ec_stream_t stream) {
    if (!dYplanes || !X || !part || !dW || (planes != 2 && planes != 3)) return EC_ERR_ARG;
    const int ns = ec_dw_tn_x3_splits(M, NX);
    if (ns <= 0) return EC_ERR_SHAPE;
    DwArgs a;
    a.yp = (const uint16_t*)dYplanes;
    a.x = (const uint16_t*)X;
    a.part = part;
    a.M = M;
    a.NX = NX;
    a.tok_per_split = (((M + ns - 1) / ns) + KT - 1) / KT * KT;
    if (a.tok_per_split * (long)NX * 2 >= (1L << 32) - (1L << 20)) return EC_ERR_SHAPE;   // 32-bit offsets inside a split
    const size_t lds = (size_t)NSTG * STG;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_tn_x3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_tn_x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    a.ntile = NX / NXT;
    a.nsplit = ns;
    const dim3 grid((unsigned)(8 * a.ntile * ((ns + 7) / 8)));
    if (planes == 3) hipLaunchKernelGGL(dw_tn_x3_kernel<3>, grid, dim3(512), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dw_tn_x3_kernel<2>, grid, dim3(512), lds, (hipStream_t)stream, a);
    const long n4 = (long)NYT * NX / 4;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, ns, n4, dW);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
