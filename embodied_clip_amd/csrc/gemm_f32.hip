// General fp32-accumulate GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
// for the actor-critic policy's forward AND backward contractions.
//
//   C[m,n] (+)= epi( sum_k A(m,k) * B(k,n) )
//   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn]   (fp32 or bf16 storage)
//
// One kernel serves NT (x W^T: nn.Linear / 1x1 conv forward), NN (dY W: input
// gradients) and TN (dY^T X: weight gradients, split-K + fp32 atomics) through
// the strides.  Replaces the cuBLAS calls behind nn.Conv2d(1x1) / nn.Linear /
// nn.GRU and their autograd backward in [U] allenact
// ResnetTensorGoalEncoder / RNNStateEncoder / LinearActorHead / LinearCriticHead
// (SURVEY.md §8a a11-a14).  gfx950 has no TF32-like mode, so the policy stays in
// true fp32: the MFMA result is bitwise an fp32 fmaf chain.
//
// Tiling: 4 waves (WM x WN), tile BM x BN x 32; operands staged (register
// prefetch + double-buffered LDS) as K-major [32][R+4] fp32 images so the
// one-float-per-lane MFMA operands (A[i=l&31][k=l>>5]) are conflict-free
// ds_read_b32 rows; bf16 operands (the frozen CLIP features) are widened on
// the way into LDS.
#include "common.h"

namespace {

constexpr int GBK = 32;

struct GemmArgs {
    const void* A;
    const void* B;
    float* C;
    int M, N, K;
    long sam, sak, sbk, sbn;
    int ldc;
    int a_bf16, b_bf16;
    int a_vec, b_vec;           // 16-B (8-B for bf16) vector loads allowed
    const float* bias;          // [N]
    const float* gbias;         // [*, N] row-group bias table
    const int* gidx;            // [M/group] row of gbias per group, or null -> group index itself
    int group;
    const float* dmask;         // same indexing as C: multiply by (dmask > 0)
    const float* rowscale;      // [M]
    int relu, accumulate, splitk;
    int ntn;
    long part_stride;           // > 0: split-K slice z writes its partial sums to C + z * part_stride (no atomics)
    int maxsum;                 // bf16x3 path: plane pairs (pa, pb) with pa + pb <= maxsum are multiplied (2: all six; 1: EC_GEMM_3PRODUCTS)
};

// element (r, k) of an operand lives at src[r*sr + k*sk]; tile rows r0.., k0..
template <int R, bool BF16>
__device__ __forceinline__ void load_frag(float (&v)[R / 8], const void* src, long sr, long sk, int r0, int k0,
                                          int Rmax, int Kmax, bool vec, int tid) {
    constexpr int PASS = R / 32;
    if (sk == 1) {   // k-contiguous: 4 consecutive k per thread, rows strided by 32
        const int kq = tid & 7, rr = tid >> 3;
        const int k = k0 + 4 * kq;
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int r = r0 + rr + 32 * i;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            if (r < Rmax) {
                const long off = (long)r * sr + k;
                if (vec && k + 3 < Kmax) {
                    if (BF16) {
                        const uint2 u = *reinterpret_cast<const uint2*>((const uint16_t*)src + off);
                        t0 = ec_lo(u.x); t1 = ec_hi(u.x); t2 = ec_lo(u.y); t3 = ec_hi(u.y);
                    } else {
                        const float4 u = *reinterpret_cast<const float4*>((const float*)src + off);
                        t0 = u.x; t1 = u.y; t2 = u.z; t3 = u.w;
                    }
                } else {
                    if (BF16) {
                        const uint16_t* p = (const uint16_t*)src + off;
                        if (k + 0 < Kmax) t0 = ec_bf2f(p[0]);
                        if (k + 1 < Kmax) t1 = ec_bf2f(p[1]);
                        if (k + 2 < Kmax) t2 = ec_bf2f(p[2]);
                        if (k + 3 < Kmax) t3 = ec_bf2f(p[3]);
                    } else {
                        const float* p = (const float*)src + off;
                        if (k + 0 < Kmax) t0 = p[0];
                        if (k + 1 < Kmax) t1 = p[1];
                        if (k + 2 < Kmax) t2 = p[2];
                        if (k + 3 < Kmax) t3 = p[3];
                    }
                }
            }
            v[4 * i + 0] = t0; v[4 * i + 1] = t1; v[4 * i + 2] = t2; v[4 * i + 3] = t3;
        }
    } else {         // row-contiguous (sr == 1) or generic: 4 consecutive rows per thread
        constexpr int RQ = R / 4;            // float4 groups along r
        constexpr int KSTEP = 256 / RQ;      // k rows covered per pass
        const int rq = tid % RQ, kk = tid / RQ;
        const int r = r0 + 4 * rq;
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int k = k0 + kk + KSTEP * i;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            if (k < Kmax) {
                const long off = (long)k * sk + (long)r * sr;
                if (vec && sr == 1 && r + 3 < Rmax) {
                    if (BF16) {
                        const uint2 u = *reinterpret_cast<const uint2*>((const uint16_t*)src + off);
                        t0 = ec_lo(u.x); t1 = ec_hi(u.x); t2 = ec_lo(u.y); t3 = ec_hi(u.y);
                    } else {
                        const float4 u = *reinterpret_cast<const float4*>((const float*)src + off);
                        t0 = u.x; t1 = u.y; t2 = u.z; t3 = u.w;
                    }
                } else {
                    if (BF16) {
                        const uint16_t* p = (const uint16_t*)src + off;
                        if (r + 0 < Rmax) t0 = ec_bf2f(p[0]);
                        if (r + 1 < Rmax) t1 = ec_bf2f(p[sr]);
                        if (r + 2 < Rmax) t2 = ec_bf2f(p[2 * sr]);
                        if (r + 3 < Rmax) t3 = ec_bf2f(p[3 * sr]);
                    } else {
                        const float* p = (const float*)src + off;
                        if (r + 0 < Rmax) t0 = p[0];
                        if (r + 1 < Rmax) t1 = p[sr];
                        if (r + 2 < Rmax) t2 = p[2 * sr];
                        if (r + 3 < Rmax) t3 = p[3 * sr];
                    }
                }
            }
            v[4 * i + 0] = t0; v[4 * i + 1] = t1; v[4 * i + 2] = t2; v[4 * i + 3] = t3;
        }
    }
}

template <int R>
__device__ __forceinline__ void store_frag(const float (&v)[R / 8], float* S, bool kcontig, int tid) {
    constexpr int LD = R + 4;
    constexpr int PASS = R / 32;
    if (kcontig) {
        const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int r = rr + 32 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(4 * kq + j) * LD + r] = v[4 * i + j];
        }
    } else {
        constexpr int RQ = R / 4;
        constexpr int KSTEP = 256 / RQ;
        const int rq = tid % RQ, kk = tid / RQ;
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int k = kk + KSTEP * i;
            *reinterpret_cast<float4*>(S + k * LD + 4 * rq) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        }
    }
}

template <int BM, int BN, int WM, int WN, bool ABF, bool BBF>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int STAGE = GBK * (LDA + LDB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_n = blockIdx.x % p.ntn, tile_m = blockIdx.x / p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // split-K range of this z-slice (multiples of GBK)
    const int nk_total = (p.K + GBK - 1) / GBK;
    const int nk_per = (nk_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * nk_per;
    const int kt1 = min(nk_total, kt0 + nk_per);
    if (kt0 >= kt1 && p.part_stride == 0) return;   // (parts mode: an empty slice still writes its zeros)

    const bool a_kc = (p.sak == 1), b_kc = (p.sbk == 1);
    float ra[BM / 8], rb[BN / 8];
    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_frag<BM, ABF>(ra, p.A, p.sam, p.sak, m0, kt0 * GBK, p.M, p.K, p.a_vec, tid);
    load_frag<BN, BBF>(rb, p.B, p.sbn, p.sbk, n0, kt0 * GBK, p.N, p.K, p.b_vec, tid);
    store_frag<BM>(ra, smem, a_kc, tid);
    store_frag<BN>(rb, smem + GBK * LDA, b_kc, tid);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = (kt + 1) < kt1;
        if (more) {
            load_frag<BM, ABF>(ra, p.A, p.sam, p.sak, m0, (kt + 1) * GBK, p.M, p.K, p.a_vec, tid);
            load_frag<BN, BBF>(rb, p.B, p.sbn, p.sbk, n0, (kt + 1) * GBK, p.N, p.K, p.b_vec, tid);
        }
        const float* sa = smem + cur * STAGE;
        const float* sb = sa + GBK * LDA;
#pragma unroll
        for (int ks = 0; ks < GBK / 2; ++ks) {
            float af[FM], bf[FN];
            const int k = 2 * ks + fh;
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = sa[k * LDA + wm * TM + i * 32 + fr];
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = sb[k * LDB + wn * TN + j * 32 + fr];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* da = smem + (cur ^ 1) * STAGE;
            store_frag<BM>(ra, da, a_kc, tid);
            store_frag<BN>(rb, da + GBK * LDA, b_kc, tid);
        }
        __syncthreads();
    }

    // epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const bool first = (blockIdx.z == 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * TN + j * 32 + fr;
        if (col >= p.N) continue;
        const float bv = (p.bias && first) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.gbias && first) {
                    const int g = row / p.group;
                    v += p.gbias[(long)(p.gidx ? p.gidx[g] : g) * p.N + col];
                }
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.rowscale) v *= p.rowscale[row];
                const long o = (long)row * p.ldc + col;
                if (p.dmask) v = (p.dmask[o] > 0.f) ? v : 0.f;
                if (p.part_stride > 0) p.C[(long)blockIdx.z * p.part_stride + o] = v;
                else if (p.splitk > 1) atomicAdd(p.C + o, v);
                else if (p.accumulate) p.C[o] += v;
                else p.C[o] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16x3 path: exact-fp32-equivalent products on the bf16 MFMA.
// An fp32 value is split while it is staged into LDS:  x = x0 + x1 + x2  (three bf16 planes, 8+8+8 mantissa
// bits = all 24).  Each bf16*bf16 product is exact in fp32 and the MFMA accumulates in fp32, so
//   a*b = sum_{i+j<=2} a_i*b_j  + O(2^-24 |a||b|)      (6 MFMAs; 3 when one operand is stored as bf16)
// i.e. fp32-GEMM accuracy at 6/16 (3/16) of the fp32-MFMA pipe time.  Used for the aligned, regular
// shapes (all the large policy GEMMs); irregular shapes stay on the fp32-MFMA kernel above.
// LDS image per operand plane: [row][32 bf16] (64-B rows), 16-B chunk index XOR (row>>2)&3: conflict-free for
// the 16-lane ds_read_b128 groups.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int xoff(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

using ::ec_split3x4;
__device__ __forceinline__ void split3x4(const float (&x)[4], uint2& p0, uint2& p1, uint2& p2) { ec_split3x4(x, p0, p1, p2); }

// fp32 [rows][K] -> three bf16 planes [rows][3][K] (plane 0 = leading bits), the same split the x3 kernels do on the fly
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ W, uint16_t* __restrict__ P, long rows, int K) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;      // one float4 of W
    const long n4 = rows * (K / 4);
    if (q >= n4) return;
    const long r = q / (K / 4);
    const int k = (int)(q - r * (K / 4)) * 4;
    const float4 w = *reinterpret_cast<const float4*>(W + r * K + k);
    const float x[4] = {w.x, w.y, w.z, w.w};
    uint2 p0, p1, p2;
    split3x4(x, p0, p1, p2);
    uint16_t* d = P + r * 3 * K + k;
    *reinterpret_cast<uint2*>(d) = p0;
    *reinterpret_cast<uint2*>(d + K) = p1;
    *reinterpret_cast<uint2*>(d + 2 * (long)K) = p2;
}

// One operand tile (R rows x 32 k): global -> registers.  KC: k-contiguous source (4 consecutive k per lane),
// else row-contiguous (a 4x4 (k x row) block per lane, transposed in registers on the way to LDS).
template <int R, bool BF16, bool KC>
struct XStage {
    static constexpr int PASS = KC ? R / 32 : (R + 127) / 128;
    float v[PASS][KC ? 4 : 16];
    __device__ __forceinline__ void load(const void* src, long sr, long sk, int r0, int k0, int Rmax, int Kmax, int tid) {
        const int kq = tid & 7, rr = tid >> 3;
        if (KC) {
            const int k = k0 + 4 * kq;
#pragma unroll
            for (int i = 0; i < PASS; ++i) {
                const int r = r0 + rr + 32 * i;
                float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
                if (r < Rmax && k < Kmax) {
                    const long off = (long)r * sr + k;
                    if (BF16) {
                        const uint2 u = *reinterpret_cast<const uint2*>((const uint16_t*)src + off);
                        t0 = ec_lo(u.x); t1 = ec_hi(u.x); t2 = ec_lo(u.y); t3 = ec_hi(u.y);
                    } else {
                        const float4 u = *reinterpret_cast<const float4*>((const float*)src + off);
                        t0 = u.x; t1 = u.y; t2 = u.z; t3 = u.w;
                    }
                }
                v[i][0] = t0; v[i][1] = t1; v[i][2] = t2; v[i][3] = t3;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PASS; ++i) {
                const int r = r0 + 4 * (rr + 32 * i);
                const bool rin = (4 * (rr + 32 * i) < R) && (r < Rmax);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = k0 + 4 * kq + kk;
                    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
                    if (rin && k < Kmax) {
                        const long off = (long)k * sk + r;
                        if (BF16) {
                            const uint2 u = *reinterpret_cast<const uint2*>((const uint16_t*)src + off);
                            t0 = ec_lo(u.x); t1 = ec_hi(u.x); t2 = ec_lo(u.y); t3 = ec_hi(u.y);
                        } else {
                            const float4 u = *reinterpret_cast<const float4*>((const float*)src + off);
                            t0 = u.x; t1 = u.y; t2 = u.z; t3 = u.w;
                        }
                    }
                    v[i][kk * 4 + 0] = t0; v[i][kk * 4 + 1] = t1; v[i][kk * 4 + 2] = t2; v[i][kk * 4 + 3] = t3;
                }
            }
        }
    }
    // registers -> LDS planes (plane stride R*64 bytes); bf16 sources fill plane 0 only
    __device__ __forceinline__ void store(unsigned char* S, int tid) const {
        const int kq = tid & 7, rr = tid >> 3;
        const int c = kq >> 1, half = (kq & 1) * 8;
        constexpr int PL = R * 64;
        if (KC) {
#pragma unroll
            for (int i = 0; i < PASS; ++i) {
                const int row = rr + 32 * i;
                unsigned char* d = S + xoff(row, c) + half;
                const float x[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
                if (BF16) {
                    *reinterpret_cast<uint2*>(d) = make_uint2(ec_pack2(x[0], x[1]), ec_pack2(x[2], x[3]));
                } else {
                    uint2 p0, p1, p2;
                    split3x4(x, p0, p1, p2);
                    *reinterpret_cast<uint2*>(d) = p0;
                    *reinterpret_cast<uint2*>(d + PL) = p1;
                    *reinterpret_cast<uint2*>(d + 2 * PL) = p2;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PASS; ++i) {
                if (4 * (rr + 32 * i) >= R) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 4 * (rr + 32 * i) + j;
                    unsigned char* d = S + xoff(row, c) + half;
                    const float x[4] = {v[i][j], v[i][4 + j], v[i][8 + j], v[i][12 + j]};
                    if (BF16) {
                        *reinterpret_cast<uint2*>(d) = make_uint2(ec_pack2(x[0], x[1]), ec_pack2(x[2], x[3]));
                    } else {
                        uint2 p0, p1, p2;
                        split3x4(x, p0, p1, p2);
                        *reinterpret_cast<uint2*>(d) = p0;
                        *reinterpret_cast<uint2*>(d + PL) = p1;
                        *reinterpret_cast<uint2*>(d + 2 * PL) = p2;
                    }
                }
            }
        }
    }
};

template <int BM, int BN, int WM, int WN, bool ABF, bool BBF, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmArgs p) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int PA = ABF ? 1 : 3, PB = BBF ? 1 : 3;
    constexpr int A_BYTES = PA * BM * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_n = blockIdx.x % p.ntn, tile_m = blockIdx.x / p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk_total = (p.K + GBK - 1) / GBK;
    const int nk_per = (nk_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * nk_per;
    const int kt1 = min(nk_total, kt0 + nk_per);
    if (kt0 >= kt1 && p.part_stride == 0) return;   // (parts mode: an empty slice still writes its zeros)

    XStage<BM, ABF, AKC> sa;
    XStage<BN, BBF, BKC> sb;
    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    sa.load(p.A, p.sam, p.sak, m0, kt0 * GBK, p.M, p.K, tid);
    sb.load(p.B, p.sbn, p.sbk, n0, kt0 * GBK, p.N, p.K, tid);
    sa.store(xsm, tid);
    sb.store(xsm + A_BYTES, tid);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (more) {
            sa.load(p.A, p.sam, p.sak, m0, (kt + 1) * GBK, p.M, p.K, tid);
            sb.load(p.B, p.sbn, p.sbk, n0, (kt + 1) * GBK, p.N, p.K, tid);
        }
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            s16x8_t af[PA][FM], bf[PB][FN];
            const int c = ks * 2 + fh;
#pragma unroll
            for (int pl = 0; pl < PA; ++pl)
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[pl][i] = *reinterpret_cast<const s16x8_t*>(xsm + pl * BM * 64 + xoff(wm * TM + i * 32 + fr, c));
#pragma unroll
            for (int pl = 0; pl < PB; ++pl)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    bf[pl][j] = *reinterpret_cast<const s16x8_t*>(xsm + A_BYTES + pl * BN * 64 + xoff(wn * TN + j * 32 + fr, c));
            // smallest terms first
#pragma unroll
            for (int sum = 2; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa < PA; ++pa) {
                    const int pb = sum - pa;
                    if (pb < 0 || pb >= PB) continue;
                    if (sum > p.maxsum) continue;                    // (wave-uniform)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[pa][i]),
                                                                               __builtin_bit_cast(bf16x8_t, bf[pb][j]),
                                                                               acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            sa.store(xsm, tid);
            sb.store(xsm + A_BYTES, tid);
        }
        __syncthreads();
    }

    const bool first = (blockIdx.z == 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * TN + j * 32 + fr;
        if (col >= p.N) continue;
        const float bv = (p.bias && first) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.gbias && first) {
                    const int g = row / p.group;
                    v += p.gbias[(long)(p.gidx ? p.gidx[g] : g) * p.N + col];
                }
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.rowscale) v *= p.rowscale[row];
                const long o = (long)row * p.ldc + col;
                if (p.dmask) v = (p.dmask[o] > 0.f) ? v : 0.f;
                if (p.part_stride > 0) p.C[(long)blockIdx.z * p.part_stride + o] = v;
                else if (p.splitk > 1) atomicAdd(p.C + o, v);
                else if (p.accumulate) p.C[o] += v;
                else p.C[o] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_x3(GemmArgs& a, hipStream_t s) {
    a.ntn = (a.N + BN - 1) / BN;
    const int ntm = (a.M + BM - 1) / BM;
    const int pa = a.a_bf16 ? 1 : 3, pb = a.b_bf16 ? 1 : 3;
    const size_t lds = (size_t)(pa * BM + pb * BN) * 64;
    dim3 grid((unsigned)(ntm * a.ntn), 1, (unsigned)a.splitk);
    const bool akc = (a.sak == 1), bkc = (a.sbk == 1);
#define EC_X3(ABF, BBF, AKC, BKC)                                                                            \
    do {                                                                                                     \
        auto kern = gemm_x3_kernel<BM, BN, WM, WN, ABF, BBF, AKC, BKC>;                                      \
        static std::atomic<uint64_t> attr_done{0};                                                                        \
        if (auto attr_g_ = ec_attr_needed(attr_done)) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)((3 * BM + 3 * BN) * 64)); \
        }                                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);                                                \
    } while (0)
    if (a.a_bf16 && a.b_bf16) return EC_ERR_UNSUPPORTED;
    if (a.a_bf16) {            // bf16 A is always k-contiguous here (feature rows)
        if (!akc) return EC_ERR_UNSUPPORTED;
        if (bkc) EC_X3(true, false, true, true); else EC_X3(true, false, true, false);
    } else if (a.b_bf16) {     // bf16 B is row(n)-contiguous (features as the TN right operand)
        if (bkc) return EC_ERR_UNSUPPORTED;
        if (akc) EC_X3(false, true, true, false); else EC_X3(false, true, false, false);
    } else {
        if (akc && bkc) EC_X3(false, false, true, true);
        else if (akc) EC_X3(false, false, true, false);
        else if (bkc) EC_X3(false, false, false, true);
        else EC_X3(false, false, false, false);
    }
#undef EC_X3
    EC_CHECK_LAUNCH();
    return EC_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(GemmArgs& a, hipStream_t s) {
    a.ntn = (a.N + BN - 1) / BN;
    const int ntm = (a.M + BM - 1) / BM;
    const size_t lds = 2 * (size_t)GBK * (BM + 4 + BN + 4) * sizeof(float);
    dim3 grid((unsigned)(ntm * a.ntn), 1, (unsigned)a.splitk);
#define EC_GEMM_LAUNCH(ABF, BBF)                                                                            \
    do {                                                                                                    \
        auto kern = gemm_f32_kernel<BM, BN, WM, WN, ABF, BBF>;                                              \
        static std::atomic<uint64_t> attr_done{0};                                                                       \
        if (auto attr_g_ = ec_attr_needed(attr_done)) {                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        }                                                                                                   \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);                                               \
    } while (0)
    if (a.a_bf16 && a.b_bf16) return EC_ERR_UNSUPPORTED;
    if (a.a_bf16) EC_GEMM_LAUNCH(true, false);
    else if (a.b_bf16) EC_GEMM_LAUNCH(false, true);
    else EC_GEMM_LAUNCH(false, false);
#undef EC_GEMM_LAUNCH
    EC_CHECK_LAUNCH();
    return EC_OK;
}

}  // namespace

extern "C" int ec_gemm_f32(const void* A, const void* B, float* Cp, int M, int N, int K, long sam, long sak, long sbk,
                           long sbn, int ldc, int flags, const float* bias, const float* gbias, const int* gidx,
                           int group, const float* dmask, const float* rowscale, int splitk, ec_stream_t stream) {
    if (!A || !B || !Cp) return EC_ERR_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return EC_ERR_SHAPE;
    if (gbias && group <= 0) return EC_ERR_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.C = Cp;
    a.M = M; a.N = N; a.K = K;
    a.sam = sam; a.sak = sak; a.sbk = sbk; a.sbn = sbn; a.ldc = ldc;
    a.a_bf16 = (flags & EC_GEMM_A_BF16) ? 1 : 0;
    a.b_bf16 = (flags & EC_GEMM_B_BF16) ? 1 : 0;
    a.relu = (flags & EC_GEMM_RELU) ? 1 : 0;
    a.accumulate = (flags & EC_GEMM_ACCUMULATE) ? 1 : 0;
    a.maxsum = (flags & EC_GEMM_3PRODUCTS) ? 1 : 2;
    const bool parts = (flags & EC_GEMM_SPLIT_PARTS) != 0;
    a.bias = bias; a.gbias = gbias; a.gidx = gidx; a.group = group;
    a.dmask = dmask; a.rowscale = rowscale;
    a.splitk = splitk < 1 ? 1 : splitk;
    const int nk_total = (K + GBK - 1) / GBK;
    if (a.splitk > nk_total) a.splitk = nk_total;
    // split-K partials are atomically ADDED into C: that is only a GEMM when C already holds the value to accumulate
    // onto (ACCUMULATE) and no nonlinearity sits between the partial sums
    // ... or every slice writes its own partial matrix (EC_GEMM_SPLIT_PARTS: C holds splitk matrices of M x ldc, summed by
    // the consumer in a fixed order -- deterministic, used by the act step); no epilogue that needs the full sum
    if (parts) {
        if (a.relu || a.accumulate || dmask || rowscale || gbias) return EC_ERR_ARG;
        if (a.splitk != splitk) return EC_ERR_ARG;          // the caller sized C for exactly `splitk` parts
        a.part_stride = (long)M * ldc;
    } else {
        a.part_stride = 0;
        if (a.splitk > 1 && (a.relu || !a.accumulate)) return EC_ERR_ARG;
    }
    // vector loads need the contiguous axis to be unit stride, the other stride a multiple of 4
    // elements and a 16-byte (8-byte for bf16) aligned base
    auto vec_ok = [](const void* p, long s_contig, long s_other, int bf16) {
        if (s_contig != 1 || (s_other & 3)) return 0;
        return (((uintptr_t)p) & (bf16 ? 7 : 15)) == 0 ? 1 : 0;
    };
    a.a_vec = (sak == 1) ? vec_ok(A, sak, sam, a.a_bf16) : vec_ok(A, sam, sak, a.a_bf16);
    a.b_vec = (sbk == 1) ? vec_ok(B, sbk, sbn, a.b_bf16) : vec_ok(B, sbn, sbk, a.b_bf16);
    hipStream_t s = (hipStream_t)stream;
    // bf16x3 path for regular shapes: vector-loadable operands, contiguous extents multiple of 4, not tiny
    const int x3_off = ec_config().gemm_no_x3;
    const bool a_ok = a.a_vec && ((sak == 1) ? (K % 4 == 0) : (M % 4 == 0 && sam == 1));
    const bool b_ok = a.b_vec && ((sbk == 1) ? (K % 4 == 0) : (N % 4 == 0 && sbn == 1));
    if (!x3_off && a_ok && b_ok && M >= 32 && N >= 32 && K >= 32 && !(a.a_bf16 && sak != 1) && !(a.b_bf16 && sbk == 1)) {
        const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128) * a.splitk;
        // N <= 32 (the compressor / combiner 128 -> 32 convs over T*N*49 rows and their input gradients): a 128-wide
        // tile would spend 3/4 of its MFMAs on padding columns
        if (N <= 32 && (long)((M + 255) / 256) * a.splitk >= 512) return launch_x3<256, 32, 4, 1>(a, s);
        // split-K parts (the act step, which runs beside the other slice's encoder): keep the workgroup count near the
        // un-split launch -- the slices only shorten the K walk; flooding the CUs with small workgroups slows the encoder
        // (measured at 256 actors: 784 workgroups -1.3 % end to end).  The tile shape does not change the summation order.
        if (a.part_stride > 0 && (long)((M + 63) / 64) * ((N + 63) / 64) * a.splitk > 400) return launch_x3<128, 128, 2, 2>(a, s);
        if (blocks128 < 512) return launch_x3<64, 64, 2, 2>(a, s);
        return launch_x3<128, 128, 2, 2>(a, s);
    }
    if (N <= 32) return launch_cfg<256, 32, 4, 1>(a, s);
    if (M <= 32) return launch_cfg<32, 256, 1, 4>(a, s);
    // fp32 MFMA is 64 cycles per 32x32x2: a 128x128 tile is a long serial chain, so shapes that give
    // fewer 128^2 tiles than ~2 per CU run on 64x64 tiles (4x the workgroups, 1/4 the chain).
    const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128) * a.splitk;
    if (blocks128 < 512) return launch_cfg<64, 64, 2, 2>(a, s);
    return launch_cfg<128, 128, 2, 2>(a, s);
}

// W fp32 [rows][K] -> bf16 planes [rows][3][K] for ec_gemm_bf16a_x3 (K % 4 == 0)
extern "C" int ec_split3_bf16(const float* W, void* planes, long rows, int K, ec_stream_t stream) {
    if (!W || !planes) return EC_ERR_ARG;
    if (rows <= 0 || K <= 0 || K % 4 != 0) return EC_ERR_SHAPE;
    const long n4 = rows * (K / 4);
    hipLaunchKernelGGL(split3_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (uint16_t*)planes, rows, K);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
