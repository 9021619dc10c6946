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
// Tiling (wave64, v_mfma_f32_32x32x16_bf16):
//   workgroup = 4 waves (WM x WN), tile BM x BN x 64; each wave owns
//   (BM/WM) x (BN/WN) as FM x FN 32x32 accumulators (fp32, 16 regs each).
//   A/B K-tiles are register-staged (global_load_dwordx4 -> ds_write_b128)
//   into a double-buffered LDS image with 128-byte rows and the 16-byte chunk
//   index XOR-swizzled by (row>>1)&7, which makes both the ds_write_b128
//   (8-lane groups = one row) and the fragment ds_read_b128 (16-lane groups
//   {0-3,12-15,20-27},...) conflict-free.  One barrier per K-tile.
//
// Fused AvgPool2d(2) (CLIP's anti-aliased stride): in POOL mode row m is
// ordered  m = 4*q + (dy*2+dx)  with q the pooled raster index, so the four
// pixels of a pooling window are accumulator registers r&3 = 0..3 of ONE lane
// (32x32 C/D layout: row = (r&3) + 8*(r>>2) + 4*(lane>>5)); the pool is an
// in-register sum after bias+ReLU -- no extra pass over HBM.
#include "common.h"

namespace {

constexpr int BK = 64;              // K-tile (bf16 elements) = 128 B rows in LDS
constexpr int ROW_BYTES = BK * 2;   // 128

struct ConvArgs {
    const uint16_t* in;
    const uint16_t* w;
    const float* bias;
    const uint16_t* res;
    uint16_t* out;
    int H, W, Cin, Cout, K, M;
    int cin_log2;
    int act;
    int ntn;  // number of N tiles
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BM, int BN, int WM, int WN, int KS, bool POOL>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;   // 16-B chunks per thread per tile
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const unsigned nwg = gridDim.x;
    const unsigned bid = ec_xcd_remap(blockIdx.x, nwg);
    const int tile_n = bid % p.ntn;
    const int tile_m = bid / p.ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread loader geometry (K-invariant) ----
    const int chunk = tid & 7;     // which 16-B chunk of the 128-B K row
    const int lrow = tid >> 3;     // 0..31
    int a_pix[A_IT];               // pixel index (b*H + y)*W + x of the row's centre tap
    int a_yx[A_IT];                // (y << 16) | x, or y = -4096 when the row is out of range
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + lrow + 32 * i;
        int b, y, x;
        if (POOL) {
            const int q = m >> 2, s = m & 3;
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            b = q / (Hp * Wp);
            const int r2 = q - b * (Hp * Wp);
            const int yp = r2 / Wp;
            y = 2 * yp + (s >> 1);
            x = 2 * (r2 - yp * Wp) + (s & 1);
        } else if (KS == 1) {
            b = 0; y = 0; x = 0;   // raster order: pixel index == m, no halo to bound-check
        } else {
            b = m / (p.H * p.W);
            const int r2 = m - b * (p.H * p.W);
            y = r2 / p.W;
            x = r2 - y * p.W;
        }
        a_pix[i] = (KS == 1 && !POOL) ? m : (b * p.H + y) * p.W + x;
        const int yx = (KS == 1) ? 0 : ((y << 16) | x);
        a_yx[i] = (m < p.M) ? yx : (-4096 * 65536);
    }

    uint4 ra[A_IT], rb[B_IT];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + chunk * 8;
        const bool kin = k < p.K;
        int dy = 0, dx = 0, ci = k;
        if (KS == 3) {
            const int tap = k >> p.cin_log2;
            ci = k & (p.Cin - 1);
            const int ky = (tap * 11) >> 5;   // tap / 3 for tap in 0..8
            dy = ky - 1;
            dx = tap - ky * 3 - 1;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int y = (a_yx[i] >> 16) + dy;
            const int x = (a_yx[i] & 0xffff) + dx;
            bool ok = kin && (y >= 0) && (y < p.H);
            if (KS == 3) ok = ok && (x >= 0) && (x < p.W);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const long off = (long)(a_pix[i] + dy * p.W + dx) * p.Cin + ci;
                v = *reinterpret_cast<const uint4*>(p.in + off);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + lrow + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kin) v = *reinterpret_cast<const uint4*>(p.w + (long)n * p.K + k);
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *reinterpret_cast<uint4*>(sa + lds_off(lrow + 32 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *reinterpret_cast<uint4*>(sb + lds_off(lrow + 32 * i, chunk)) = rb[i];
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) load_tile(kt + 1);
        const unsigned char* sa = smem + cur * (A_BYTES + B_BYTES);
        const unsigned char* sb = sa + A_BYTES;
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
                        __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: staged through LDS so every global access is a coalesced 16-B chunk ----
    // (the A/B staging buffers are free: the K loop ended on a barrier)
    //   1. residual tile -> LDS (16-B loads)            [only with a residual]
    //   2. each lane folds bias/residual/activation (or the 2x2 pool) into its accumulator
    //      elements in fp32, rounds ONCE to bf16 and writes them to its own LDS slots
    //   3. LDS -> global as 16-B row chunks
    constexpr int CH = BN / 8;                 // 16-B chunks per tile row
    constexpr int PITCH = BN * 2 + 16;         // bytes; +16 staggers banks between rows
    constexpr int RPP = 256 / CH;              // rows per pass
    constexpr int OUT_ROWS = POOL ? BM / 4 : BM;
    const int orow0 = POOL ? (m0 >> 2) : m0;
    const int Mout = POOL ? (p.M >> 2) : p.M;
    const int srow = tid / CH, schunk = tid % CH;
    if (!POOL && p.res) {
#pragma unroll
        for (int r0 = 0; r0 < OUT_ROWS; r0 += RPP) {
            const int row = r0 + srow;
            if (orow0 + row < Mout)
                *reinterpret_cast<uint4*>(smem + row * PITCH + schunk * 16) =
                    *reinterpret_cast<const uint4*>(p.res + (long)(orow0 + row) * p.Cout + n0 + schunk * 8);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int lcol = wn * TN + j * 32 + frow;
        const float bv = p.bias ? p.bias[n0 + lcol] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int lrow0 = wm * TM + i * 32 + 4 * fhalf;
            if (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum += fmaxf(acc[i][j][g * 4 + r] + bv, 0.f);
                    const int row = (lrow0 + 8 * g) >> 2;
                    *reinterpret_cast<uint16_t*>(smem + row * PITCH + lcol * 2) = ec_f2bf(0.25f * sum);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = lrow0 + (r & 3) + 8 * (r >> 2);
                    uint16_t* slot = reinterpret_cast<uint16_t*>(smem + row * PITCH + lcol * 2);
                    float v = acc[i][j][r] + bv;
                    if (p.res) v += ec_bf2f(*slot);
                    if (p.act == EC_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == EC_ACT_QUICKGELU) v = v / (1.f + __expf(-1.702f * v));
                    *slot = ec_f2bf(v);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < OUT_ROWS; r0 += RPP) {
        const int row = r0 + srow;
        if (orow0 + row < Mout)
            *reinterpret_cast<uint4*>(p.out + (long)(orow0 + row) * p.Cout + n0 + schunk * 8) =
                *reinterpret_cast<const uint4*>(smem + row * PITCH + schunk * 16);
    }
}

template <int BM, int BN, int WM, int WN, int KS, bool POOL>
int launch(const ConvArgs& a, hipStream_t s) {
    ConvArgs p = a;
    p.ntn = a.Cout / BN;
    const int ntm = (a.M + BM - 1) / BM;
    const size_t lds = 2 * (size_t)(BM + BN) * ROW_BYTES;
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, KS, POOL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntm * p.ntn)), dim3(256), lds, s, p);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

template <int KS, bool POOL>
int dispatch_tile(const ConvArgs& a, hipStream_t s) {
    if (a.Cout % 128 == 0) return launch<128, 128, 2, 2, KS, POOL>(a, s);
    if (a.Cout % 64 == 0) return launch<256, 64, 4, 1, KS, POOL>(a, s);
    if (a.Cout % 32 == 0) return launch<256, 32, 4, 1, KS, POOL>(a, s);
    return EC_ERR_SHAPE;
}

}  // namespace

extern "C" int ec_conv_bf16(const void* in, const void* w, const float* bias, const void* res, void* out, int B,
                            int H, int W, int Cin, int Cout, int ksize, int pool, int act, ec_stream_t stream) {
    if (!in || !w || !out) return EC_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0) return EC_ERR_SHAPE;
    if (ksize != 1 && ksize != 3) return EC_ERR_SHAPE;
    if (Cin < 8 || (Cin & (Cin - 1)) != 0 || Cout % 32 != 0) return EC_ERR_SHAPE;
    if (pool && ((H & 1) || (W & 1) || res != nullptr || act != EC_ACT_RELU)) return EC_ERR_SHAPE;
    if (H >= 4096 || W >= 65536) return EC_ERR_SHAPE;
    if ((long)B * H * W >= (1L << 31) / 4) return EC_ERR_SHAPE;
    ConvArgs a;
    a.in = (const uint16_t*)in;
    a.w = (const uint16_t*)w;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.out = (uint16_t*)out;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.K = ksize * ksize * Cin;
    a.M = B * H * W;
    a.cin_log2 = ec_ilog2(Cin);
    a.act = act;
    a.ntn = 0;
    hipStream_t s = (hipStream_t)stream;
    if (ksize == 3) return pool ? dispatch_tile<3, true>(a, s) : dispatch_tile<3, false>(a, s);
    return pool ? dispatch_tile<1, true>(a, s) : dispatch_tile<1, false>(a, s);
}

extern "C" int ec_gemm_bf16(const void* A, const void* Wt, const float* bias, const void* res, void* out, int M, int N,
                            int K, int act, ec_stream_t stream) {
    if (!A || !Wt || !out) return EC_ERR_ARG;
    if (M <= 0 || N % 32 != 0 || K % 8 != 0 || K < 8) return EC_ERR_SHAPE;
    // GEMM = 1x1 conv over a [1, 1, M] "image" with Cin = K (no power-of-two need: KS==1 never splits k).
    ConvArgs a;
    a.in = (const uint16_t*)A;
    a.w = (const uint16_t*)Wt;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.out = (uint16_t*)out;
    a.H = 1; a.W = M; a.Cin = K; a.Cout = N;
    a.K = K; a.M = M;
    a.cin_log2 = 0;
    a.act = act;
    a.ntn = 0;
    return dispatch_tile<1, false>(a, (hipStream_t)stream);
}
