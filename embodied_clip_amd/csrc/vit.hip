// CLIP VisionTransformer embedder (ClipViTPreprocessor) and AttentionPool2d.
//
// Replaces ([U] openai/CLIP clip/model.py @40f5484c, pinned by
// primitive_probing/environment.yml:22):
//   * VisionTransformer.forward as driven by [U] allenact ClipViTEmbedder:
//     conv1 patch-embed -> +class_embedding -> +positional_embedding -> ln_pre ->
//     resblocks[:-1]  (no ln_post / proj)                     
//   * ResidualAttentionBlock: x += out_proj(MHA(ln_1 x)); x += c_proj(QuickGELU(c_fc(ln_2 x)))
//   * AttentionPool2d.forward, called detached at
//     primitive_probing/generate_data/thor_image_features.py:62,112          (a6)
//
// All contractions (patch-embed, QKV, out-proj, MLP, k/v/q/c projections) run on
// the bf16 MFMA GEMM of conv_igemm.hip with fused bias / QuickGELU / residual
// epilogues; this file holds the bandwidth-bound glue (patchify, LayerNorm,
// token assembly) and the attention cores, which keep the 50 KV rows of one
// (frame, head) in LDS.  The residual stream is bf16, statistics are fp32.
#include <new>

#include "common.h"

extern "C" int ec_gemm_bf16(const void* A, const void* Wt, const float* bias, const void* res, void* out, int M, int N,
                            int K, int act, ec_stream_t stream);

extern "C" int ec_bf16_to_f32(const void* in, float* out, long rows, long row_len, long in_stride, ec_stream_t stream);

// conv_igemm.hip: the 8-wave GEMM with LayerNorm folded in (consumer) / emitting the rows' LayerNorm records (producer)
int ec_gemm_bf16_ln8(const void* A, const void* Wt, const float* bias, const void* res, void* out, int M, int N, int K, int act,
                     const float* ln_s, const float* ln_stats, int ln_np, float* stats_out, int* np_out, ec_stream_t stream);

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// fp32 NHWC frame -> bf16 [B*G*G, P*P*3] rows, K ordered (ky, kx, c): each (patch, ky) is one
// contiguous run of P*3 floats of an image row.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ rgb, uint16_t* __restrict__ out, int R,
                                                      int P, int G, long total4) {
    const int run = P * 3;                 // floats per (patch, ky)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        long e = i * 4;                    // 4 consecutive K elements (run % 4 == 0)
        const int k = (int)(e % ((long)P * run));
        const long patch = e / ((long)P * run);
        const int ky = k / run, rem = k - ky * run;
        const int px = (int)(patch % G);
        const long t = patch / G;
        const int py = (int)(t % G);
        const long b = t / G;
        const float4 v = *reinterpret_cast<const float4*>(rgb + ((b * R + (long)py * P + ky) * R + (long)px * P) * 3 + rem);
        uint2 o;
        o.x = ec_pack2(v.x, v.y);
        o.y = ec_pack2(v.z, v.w);
        *reinterpret_cast<uint2*>(out + e) = o;
    }
}

// LayerNorm over D (fp32 statistics), one wave per row.  MODE 0: in = bf16 rows.
// MODE 1 (token assembly + ln_pre): row t of frame b is (t==0 ? cls : patch_emb[b, t-1]) + pos[t].
template <int MODE>
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ in, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, uint16_t* __restrict__ out,
                                                       long rows, int D, int L, float eps, float4* __restrict__ stats_out = nullptr) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    constexpr int MAXV = 16;               // D <= 1024
    if (MODE == 0 && (D & 255) == 0) {
        // a lane owns D / 64 CONSECUTIVE channels (a multiple of 4): 8-byte bf16 loads / stores and 16-byte fp32 loads of
        // gamma / beta instead of D / 64 two-byte accesses per lane (the kernel is latency-bound: 18.8 -> ~10 us per call)
        const int per4 = D / 256;          // groups of 4 channels per lane
        const uint16_t* p = in + row * D + lane * (4 * per4);
        float x[MAXV];
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < MAXV / 4; ++g)
            if (g < per4) {
                const uint2 u = *reinterpret_cast<const uint2*>(p + 4 * g);
                x[4 * g + 0] = ec_lo(u.x); x[4 * g + 1] = ec_hi(u.x); x[4 * g + 2] = ec_lo(u.y); x[4 * g + 3] = ec_hi(u.y);
                s += (x[4 * g + 0] + x[4 * g + 1]) + (x[4 * g + 2] + x[4 * g + 3]);
            }
        const float mean = wave_sum_f(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i < 4 * per4) { const float d0 = x[i] - mean; q += d0 * d0; }
        const float rstd = rsqrtf(wave_sum_f(q) / (float)D + eps);
        uint16_t* o = out + row * D + lane * (4 * per4);
        const float* gp = gamma + lane * (4 * per4);
        const float* bp = beta + lane * (4 * per4);
#pragma unroll
        for (int g = 0; g < MAXV / 4; ++g)
            if (g < per4) {
                const float4 ga = *reinterpret_cast<const float4*>(gp + 4 * g);
                const float4 be = *reinterpret_cast<const float4*>(bp + 4 * g);
                uint2 u;
                u.x = ec_pack2((x[4 * g + 0] - mean) * rstd * ga.x + be.x, (x[4 * g + 1] - mean) * rstd * ga.y + be.y);
                u.y = ec_pack2((x[4 * g + 2] - mean) * rstd * ga.z + be.z, (x[4 * g + 3] - mean) * rstd * ga.w + be.w);
                *reinterpret_cast<uint2*>(o + 4 * g) = u;
            }
        return;
    }
    float v[MAXV];
    const int per = D / 64;
    float s = 0.f;
    if (MODE == 0) {
        const uint16_t* p = in + row * D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i < per) { v[i] = ec_bf2f(p[i * 64 + lane]); s += v[i]; }
    } else {
        const long b = row / L;
        const int t = (int)(row - b * L);
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i < per) {
                const int d = i * 64 + lane;
                const float base = (t == 0) ? cls[d] : ec_bf2f(in[(b * (L - 1) + (t - 1)) * D + d]);
                v[i] = base + pos[(long)t * D + d];
                s += v[i];
            }
    }
    const float mean = wave_sum_f(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < per) { const float d0 = v[i] - mean; q += d0 * d0; }
    const float rstd = rsqrtf(wave_sum_f(q) / (float)D + eps);
    uint16_t* o = out + row * D;
    float so = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < per) {
            const int d = i * 64 + lane;
            const uint16_t ob = (uint16_t)(ec_pack2((v[i] - mean) * rstd * gamma[d] + beta[d], 0.f) & 0xffffu);
            o[d] = ob;
            v[i] = ec_bf2f(ob);                                  // the ROUNDED value: what the next block's folded LayerNorm sees
            so += v[i];
        }
    if (stats_out) {   // the output row's LayerNorm record {sum, M2, count} for the first block's folded ln_1 (one record per row)
        so = wave_sum_f(so);
        const float mo = so / (float)D;
        float qo = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i < per) { const float d0 = v[i] - mo; qo += d0 * d0; }
        qo = wave_sum_f(qo);
        if (lane == 0) stats_out[row] = make_float4(so, qo, (float)D, 0.f);
    }
}

// LayerNorm record {sum, M2, count} of every bf16 row of x [rows, D] (one wave per row): the input of a tower whose first block
// is not preceded by a LayerNorm launch (the text tower's token + positional embedding)
__global__ __launch_bounds__(256) void row_stats_kernel(const uint16_t* __restrict__ x, float4* __restrict__ stats, long rows, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[16];
    const int per = D / 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < per) { v[i] = ec_bf2f(x[row * D + i * 64 + lane]); s += v[i]; }
    s = wave_sum_f(s);
    const float m = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < per) { const float d0 = v[i] - m; q += d0 * d0; }
    q = wave_sum_f(q);
    if (lane == 0) stats[row] = make_float4(s, q, (float)D, 0.f);
}

// Folds a LayerNorm into the Linear that consumes it ([U] clip/model.py ResidualAttentionBlock: attn(ln_1(x)), mlp(ln_2(x))):
//   LN(x) W^T + b = rstd (x (W diag(gamma))^T - mean s) + c,   s[n] = sum_k Wg[n, k],   c[n] = sum_k beta[k] W[n, k] + b[n]
// Wg is rounded to bf16 and s sums the ROUNDED values (what the MFMA multiplies), so the mean term cancels exactly.  One
// workgroup per output row n; once per set of weights.
__global__ __launch_bounds__(256) void ln_fold_kernel(const uint16_t* __restrict__ W, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ b,
                                                     uint16_t* __restrict__ Wg, float* __restrict__ s_out, float* __restrict__ c_out, int K) {
    const int n = blockIdx.x;
    float s = 0.f, c = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = ec_bf2f(W[(long)n * K + k]);
        const uint16_t g = (uint16_t)(ec_pack2(w * gamma[k], 0.f) & 0xffffu);
        Wg[(long)n * K + k] = g;
        s += ec_bf2f(g);
        c = fmaf(beta[k], w, c);
    }
    __shared__ float rs[4], rc[4];
    s = wave_sum_f(s); c = wave_sum_f(c);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_out[n] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
        c_out[n] = ((rc[0] + rc[1]) + (rc[2] + rc[3])) + b[n];
    }
}

// Multi-head self-attention core for L <= 64 tokens and head dim 64: one wave per (frame, head).
// qkv bf16 [B*L, 3*D] (q | k | v, head h at columns h*64..), out bf16 [B*L, D].
//   S^T[key][query] = K Q^T   (swapped operands: a lane owns ONE query column and 32 of the 64 keys,
//                              so the softmax is in-lane + one exchange with the other half-wave)
//   O[query][d]     = P V     (P stays in registers; V is read from LDS through ds_read_b64_tr_b16 with the matching k-permutation)
// (component-wise: `ok ? v : zero` on the uint4 STRUCT is a pointer select and pins both operands in scratch memory)
__device__ __forceinline__ uint4 ec_sel4(bool ok, const uint4& v) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); }

__global__ __launch_bounds__(64) void mha_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int L,
                                                 int D, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) uint16_t sv[64 * 72];    // V: [token][d], row pitch 72 (144 B)
    const int lane = threadIdx.x;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const uint16_t* base = qkv + (long)b * L * 3 * D + h * 64;
    const int fr = lane & 31, fh = lane >> 5;
    // Q and K fragments straight from global memory in MFMA operand layout (lane = token fr / fr + 32, 8 consecutive
    // channels per k-step: one 16-byte load each, all 16 issued before the first use); only V goes through LDS, because
    // the PV step needs it transposed.  9 KB of LDS per workgroup instead of 27: every (frame, head) wave of a CU is
    // resident at once.  Tokens >= L are zero.
    // Every global load of the wave is issued BEFORE the first use (24 x 16 B per lane in flight): rows past L are read from a
    // clamped address and zeroed with a select -- behind an exec-masked `if (t < L)` the compiler waits for each load before it
    // issues the next (a serial chain of L2 round trips: this kernel was 14 us for 38 MB; round 6).
    uint4 qg[4][2], kg[4][2], vv[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int t = a * 32 + fr, tc = t < L ? t : L - 1;
            const uint16_t* r = base + (long)tc * 3 * D + ks * 16 + fh * 8;
            qg[ks][a] = *reinterpret_cast<const uint4*>(r);
            kg[ks][a] = *reinterpret_cast<const uint4*>(r + D);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = i * 64 + lane, t = e >> 3, c = e & 7, tc = t < L ? t : L - 1;
        vv[i] = *reinterpret_cast<const uint4*>(base + (long)tc * 3 * D + 2 * D + c * 8);
    }
    __builtin_amdgcn_sched_barrier(0);     // (keeps the 24 loads together: the scheduler otherwise interleaves load / wait / use)
    const bool okt[2] = {fr < L, 32 + fr < L};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = i * 64 + lane, t = e >> 3, c = e & 7;
        *reinterpret_cast<uint4*>(sv + t * 72 + c * 8) = ec_sel4(t < L, vv[i]);     // row-major: the PV step turns it with ds_read_b64_tr_b16
    }
    __syncthreads();
    // ---- S^T = K Q^T : acc[kf][qf], rows = keys, cols = queries ----
    f32x16_t st[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[a][c][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                st[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ec_sel4(okt[a], kg[ks][a])),
                                                                  __builtin_bit_cast(bf16x8_t, ec_sel4(okt[c], qg[ks][c])), st[a][c], 0, 0, 0);
    }
    // ---- softmax over keys, per query column (query = qf*32 + fr) ----
    // this lane's keys: kf*32 + (r&3) + 8*(r>>2) + 4*fh
    uint32_t pk[2][2][8];   // [qf][kf][pairs]: bf16 probabilities, packed in register order
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float s = (key < L) ? st[a][c][r] * scale : -INFINITY;
                st[a][c][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[a][c][r] - mx);   // exp(-inf) = 0 for padded keys
                st[a][c][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; r += 2) pk[c][a][r >> 1] = ec_pack2(st[a][c][r] * inv, st[a][c][r + 1] * inv);
    }
    // ---- O = P V : per 16-key step s (keys 16s..16s+15) this lane holds, as MFMA k-slots e=0..7,
    //      keys 16s + 8*(e>>2) + 4*fh + (e&3)  ==  registers r = 8*(s&1) + {0..3} and {4..7} of key-frag s>>1;
    //      the B operand reads V^T[d][same keys] (two 8-byte pieces), so the permutation cancels. ----
    f32x16_t oacc[2][2];   // [qf][df]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[c][d][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int a = s >> 1, rb = 4 * (s & 1);      // pairs rb..rb+3 of pk[c][a]
        s16x8_t vf[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            // transpose read: a 16-lane group reads V[4 keys][16 channels] and lane t receives channel t's 4 keys
            // (lane t supplies the address of row t >> 2, channels 4 (t & 3)..): keys 16 s + 4 fh + 0..3 and + 8
            typedef short s16x4_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
            const int t16 = lane & 15;
            const uint16_t* vblk = sv + (16 * s + 4 * fh + (t16 >> 2)) * 72 + d * 32 + (fr & 16) + (t16 & 3) * 4;
#if defined(__HIP_DEVICE_COMPILE__)
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)vblk);
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vblk + 8 * 72));
            vf[d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint4 pu = make_uint4(pk[c][a][rb], pk[c][a][rb + 1], pk[c][a][rb + 2], pk[c][a][rb + 3]);
            const s16x8_t pf = __builtin_bit_cast(s16x8_t, pu);
#pragma unroll
            for (int d = 0; d < 2; ++d)
                oacc[c][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pf),
                                                                    __builtin_bit_cast(bf16x8_t, vf[d]), oacc[c][d], 0, 0, 0);
        }
    }
    // D[query][d]: col = d (lane&31), rows = queries (r&3) + 8*(r>>2) + 4*fh
    uint16_t* ob = out + (long)b * L * D + h * 64;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (q < L) ob[(long)q * D + d * 32 + fr] = (uint16_t)(ec_pack2(oacc[c][d][r], 0.f) & 0xffffu);
            }
}

__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, long n, long in_stride,
                                   long row_len) {
    // out[r, c] = in[r * in_stride + c], n = rows * row_len
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long r = i / row_len, c = i - r * row_len;
    out[i] = ec_bf2f(in[r * in_stride + c]);
}

// AttentionPool2d token assembly: tok[b,0] = mean_p feat[b,p] + pos[0]; tok[b,1+p] = feat[b,p] + pos[1+p];
// cls[b] = tok[b,0] (dense copy for the q projection)
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const uint16_t* __restrict__ feat,
                                                             const float* __restrict__ pos, uint16_t* __restrict__ tok,
                                                             uint16_t* __restrict__ cls, int HW, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) {
            const float v = ec_bf2f(feat[((long)b * HW + p) * C + c]);
            s += v;
            tok[((long)b * (HW + 1) + 1 + p) * C + c] = (uint16_t)(ec_pack2(v + pos[(long)(1 + p) * C + c], 0.f) & 0xffffu);
        }
        const uint16_t m = (uint16_t)(ec_pack2(s / (float)HW + pos[c], 0.f) & 0xffffu);
        tok[(long)b * (HW + 1) * C + c] = m;
        cls[(long)b * C + c] = m;
    }
}

// CLS-only query attention (AttentionPool2d returns token 0): one wave per (frame, head), head dim 64, up to
// ATTNPOOL_MAX_L tokens (RN50: 7x7 + 1 = 50; RN50x16 at 384 px: 12x12 + 1 = 145, 48 heads; RN50x64 at 448 px: 197).
// q bf16 [B, C]; kv bf16 [B*L, 2C] (k | v); out bf16 [B, C]
constexpr int ATTNPOOL_MAX_L = 1024;
__global__ __launch_bounds__(64) void attnpool_core_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kv,
                                                          uint16_t* __restrict__ out, int L, int C, int heads,
                                                          float scale) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const float qd = ec_bf2f(q[(long)b * C + h * 64 + lane]) * scale;   // q is scaled before the dot product
    const uint16_t* kb = kv + (long)b * L * 2 * C + h * 64;
    __shared__ float sc[ATTNPOOL_MAX_L];
    for (int t = 0; t < L; ++t) {
        const float s = wave_sum_f(qd * ec_bf2f(kb[(long)t * 2 * C + lane]));
        if (lane == 0) sc[t] = s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = lane; t < L; t += 64) mx = fmaxf(mx, sc[t]);
    mx = wave_max_f(mx);
    float sum = 0.f;
    for (int t = lane; t < L; t += 64) {
        const float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
    const float inv = 1.f / wave_sum_f(sum);
    for (int t = lane; t < L; t += 64) sc[t] *= inv;
    __syncthreads();
    float o = 0.f;
    for (int t = 0; t < L; ++t) o += sc[t] * ec_bf2f(kb[(long)t * 2 * C + C + lane]);
    out[(long)b * C + h * 64 + lane] = (uint16_t)(ec_pack2(o, 0.f) & 0xffffu);
}

// General multi-head self-attention core: any number of tokens (K/V of one head fit the LDS up to ~500 tokens),
// head dim 64, optional causal mask (the CLIP text tower's `build_attention_mask`).  One workgroup of 4 waves per
// (sequence, head); K and V rows live in LDS as bf16 (pitch 72), each wave walks query rows: lane j scores keys
// j, j+64, ..., the softmax is a wave reduction, then lane d accumulates output dim d.  Same rounding points as
// mha_kernel (P and O rounded to bf16).  Used for L > 64 (ViT-B/16, ViT-L/14) and for the text tower -- functional
// coverage, not a tuned kernel (the 50-token ViT-B/32 path keeps the MFMA core above).
template <bool CAUSAL>
__global__ __launch_bounds__(256) void mha_general_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                         int L, int D, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    uint16_t* sk = reinterpret_cast<uint16_t*>(gsm);
    uint16_t* sv = sk + (size_t)L * 72;
    const int Lp = (L + 63) / 64 * 64;
    float* sq = reinterpret_cast<float*>(sv + (size_t)L * 72);      // [4][64] scaled query
    float* sp = sq + 4 * 64;                                         // [4][Lp] probabilities
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const uint16_t* base = qkv + (long)b * L * 3 * D + h * 64;
    for (int e = tid; e < L * 8; e += 256) {
        const int t = e >> 3, c = e & 7;
        const uint16_t* r = base + (long)t * 3 * D + c * 8;
        *reinterpret_cast<uint4*>(sk + t * 72 + c * 8) = *reinterpret_cast<const uint4*>(r + D);
        *reinterpret_cast<uint4*>(sv + t * 72 + c * 8) = *reinterpret_cast<const uint4*>(r + 2 * D);
    }
    __syncthreads();
    float* q = sq + wave * 64;
    float* pr = sp + wave * Lp;
    constexpr int MAXJ = 8;                                          // L <= 512
    for (int qi = wave; qi < L; qi += 4) {
        q[lane] = ec_bf2f(base[(long)qi * 3 * D + lane]) * scale;
        __builtin_amdgcn_wave_barrier();
        const int nk = CAUSAL ? qi + 1 : L;
        float sc[MAXJ];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXJ; ++i) {
            const int j = i * 64 + lane;
            sc[i] = -INFINITY;
            if (i * 64 < nk && j < nk) {
                float a = 0.f;
                const uint16_t* kr = sk + j * 72;
#pragma unroll 8
                for (int d = 0; d < 64; ++d) a += q[d] * ec_bf2f(kr[d]);
                sc[i] = a;
                mx = fmaxf(mx, a);
            }
        }
        mx = wave_max_f(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXJ; ++i) {
            const int j = i * 64 + lane;
            if (i * 64 < nk) {
                const float e = (j < nk) ? __expf(sc[i] - mx) : 0.f;
                sc[i] = e;
                sum += e;
            }
        }
        const float inv = 1.f / wave_sum_f(sum);
#pragma unroll
        for (int i = 0; i < MAXJ; ++i) {
            const int j = i * 64 + lane;
            if (i * 64 < nk && j < nk) pr[j] = ec_bf2f((uint16_t)(ec_pack2(sc[i] * inv, 0.f) & 0xffffu));   // P in bf16
        }
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int j = 0; j < nk; ++j) o += pr[j] * ec_bf2f(sv[j * 72 + lane]);
        out[((long)b * L + qi) * D + h * 64 + lane] = (uint16_t)(ec_pack2(o, 0.f) & 0xffffu);
        __builtin_amdgcn_wave_barrier();
    }
}

// text tower input: x[b, t, :] = token_embedding[tokens[b, t]] + positional_embedding[t]   (bf16 residual stream)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ emb,
                                                          const float* __restrict__ pos, uint16_t* __restrict__ x,
                                                          long rows, int ctx, int D, int vocab) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (D / 2)) return;
    const long row = i / (D / 2);
    const int d = (int)(i - row * (D / 2)) * 2;
    const int t = (int)(row % ctx);
    int tok = tokens[row];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const float a = emb[(long)tok * D + d] + pos[(long)t * D + d], bq = emb[(long)tok * D + d + 1] + pos[(long)t * D + d + 1];
    reinterpret_cast<uint32_t*>(x)[i] = ec_pack2(a, bq);
}

// ln_final on the EOT row (first arg-max of the token ids, as torch.argmax) of every sequence: one wave per sequence
__global__ __launch_bounds__(64) void eot_layernorm_kernel(const int32_t* __restrict__ tokens, const uint16_t* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          uint16_t* __restrict__ out, int ctx, int D, float eps) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int best = -1, at = 0;
    for (int t = 0; t < ctx; ++t) {            // uniform over the wave
        const int v = tokens[(long)b * ctx + t];
        if (v > best) { best = v; at = t; }
    }
    const uint16_t* p = x + ((long)b * ctx + at) * D;
    float v[16];
    const int per = D / 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < per) { v[i] = ec_bf2f(p[i * 64 + lane]); s += v[i]; }
    const float mean = wave_sum_f(s) / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < per) { const float d0 = v[i] - mean; qq += d0 * d0; }
    const float rstd = rsqrtf(wave_sum_f(qq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < per) {
            const int d = i * 64 + lane;
            out[(long)b * D + d] = (uint16_t)(ec_pack2((v[i] - mean) * rstd * gamma[d] + beta[d], 0.f) & 0xffffu);
        }
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// LayerNorm-folded copies of a tower's blocks (handle-owned, built once at create time by ln_fold_kernel): per block
// Wg_qkv [3D][D] and Wg_fc [4D][D] (bf16, = W diag(gamma)) and the fp32 vectors s_qkv, c_qkv (3D each), s_fc, c_fc (4D each).
struct ec_lnfold {
    uint16_t* w = nullptr;
    float* f = nullptr;
    bool ok = false;
    ~ec_lnfold() {
        if (w) (void)hipFree(w);
        if (f) (void)hipFree(f);
    }
};

struct ec_vit {
    int width, layers, heads, patch, res, grid, L;
    const uint16_t* w;
    const float* f;
    size_t n_w, n_f;
    int conv8_min_tiles = 0;      // 0 = library default (ec_vit_set_conv8_min_tiles)
    ec_lnfold fold;
};

#define RC(x) do { int rc__ = (x); if (rc__ != EC_OK) return rc__; } while (0)

namespace {
constexpr int MHA_GENERAL_MAX_TOKENS = 512;
size_t mha_general_lds(int L) { return (size_t)L * 72 * 2 * 2 + 4 * 64 * 4 + 4 * (size_t)((L + 63) / 64 * 64) * 4; }

// `layers` ResidualAttentionBlocks ([U] clip/model.py) on the bf16 residual stream x [B*L, D]; w / f point at the
// first block's weights (wqkv, wo, wfc, wpr) / params (ln1 w,b, bqkv, bo, ln2 w,b, bfc, bpr) and are advanced.
// Builds the folded copies for `layers` blocks whose weights / params start at (w, f): EC_OK, or leaves fold.ok = false when the
// geometry does not suit the folded GEMMs (D % 128 != 0, more than 8 records per row) -- run_blocks then keeps the LayerNorm launches.
int build_lnfold(ec_lnfold& fold, const uint16_t* w, const float* f, int layers, int D) {
    if (layers <= 0 || D % 128 != 0 || D / 128 > 8) return EC_OK;
    // (the weights may have been uploaded on a non-blocking stream of the caller's: the fold below reads them on the null stream)
    if (hipDeviceSynchronize() != hipSuccess) return EC_ERR_LAUNCH;
    const size_t Dz = (size_t)D, wl = 7 * Dz * Dz, fl = 14 * Dz;
    if (hipMalloc(&fold.w, layers * wl * sizeof(uint16_t)) != hipSuccess) { fold.w = nullptr; return EC_ERR_ALLOC; }
    if (hipMalloc(&fold.f, layers * fl * sizeof(float)) != hipSuccess) { fold.f = nullptr; return EC_ERR_ALLOC; }
    for (int l = 0; l < layers; ++l) {
        const float *ln1w = f, *ln1b = f + D, *bqkv = f + 2 * D, *bo = bqkv + 3 * D, *ln2w = bo + D, *ln2b = ln2w + D, *bfc = ln2b + D;
        const uint16_t *wqkv = w, *wfc = w + (size_t)4 * D * D;
        uint16_t* wg = fold.w + l * wl;
        float* fg = fold.f + l * fl;
        hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)(3 * D)), dim3(256), 0, nullptr, wqkv, ln1w, ln1b, bqkv, wg, fg, fg + 3 * Dz, D);
        hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)(4 * D)), dim3(256), 0, nullptr, wfc, ln2w, ln2b, bfc, wg + 3 * Dz * Dz,
                           fg + 6 * Dz, fg + 10 * Dz, D);
        f += 13 * Dz;
        w += 12 * Dz * Dz;
    }
    if (hipStreamSynchronize(nullptr) != hipSuccess || hipGetLastError() != hipSuccess) return EC_ERR_LAUNCH;
    fold.ok = true;
    return EC_OK;
}

// `layers` ResidualAttentionBlocks ([U] clip/model.py) on the bf16 residual stream x [B*L, D]; w / f point at the
// first block's weights (wqkv, wo, wfc, wpr) / params (ln1 w,b, bqkv, bo, ln2 w,b, bfc, bpr) and are advanced.
// fold != nullptr (built by build_lnfold): ln_1 / ln_2 are folded into the QKV / c_fc GEMMs -- no LayerNorm launch, no
// normalised copy of x: the GEMMs multiply the RAW rows against W diag(gamma) and apply (mean, rstd) in their epilogue; the rows'
// statistics come as partial records out of the epilogue of the GEMM that produced x (out_proj / c_proj + residual), `stats`
// holding `np` records per row on entry (the caller's: ln_pre's output rows, or row_stats_kernel).
int run_blocks(const uint16_t*& w, const float*& f, int layers, int heads, uint16_t* x, uint16_t* hbuf, uint16_t* qkv,
               uint16_t* att, uint16_t* mlp, int B, int L, int D, bool causal, const ec_lnfold* fold, float* stats, int np,
               ec_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)B * L;
    const unsigned lnb = (unsigned)((rows + 3) / 4);
    const bool general = causal || L > 64;
    const bool folded = fold && fold->ok && stats;
    if (general) {
        if (L > MHA_GENERAL_MAX_TOKENS) return EC_ERR_SHAPE;
        static std::atomic<uint64_t> attr_done{0};
        if (auto attr_g_ = ec_attr_needed(attr_done)) {
            const int mx = (int)mha_general_lds(MHA_GENERAL_MAX_TOKENS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_general_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_general_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        }
    }
    for (int l = 0; l < layers; ++l) {
        const float *ln1w = f, *ln1b = f + D, *bqkv = f + 2 * D, *bo = bqkv + 3 * D, *ln2w = bo + D, *ln2b = ln2w + D;
        const float *bfc = ln2b + D, *bpr = bfc + 4 * D;
        f = bpr + D;
        const uint16_t *wqkv = w, *wo = w + (size_t)3 * D * D, *wfc = wo + (size_t)D * D, *wpr = wfc + (size_t)4 * D * D;
        w = wpr + (size_t)4 * D * D;
        const size_t Dz = (size_t)D;
        const uint16_t* wg = folded ? fold->w + (size_t)l * 7 * Dz * Dz : nullptr;
        const float* fg = folded ? fold->f + (size_t)l * 14 * Dz : nullptr;
        if (folded)
            RC(ec_gemm_bf16_ln8(x, wg, fg + 3 * Dz, nullptr, qkv, (int)rows, 3 * D, D, EC_ACT_NONE, fg, stats, np, nullptr, nullptr, stream));
        else {
        hipLaunchKernelGGL(layernorm_kernel<0>, dim3(lnb), dim3(256), 0, s, x, nullptr, nullptr, ln1w, ln1b, hbuf, rows, D,
                           L, 1e-5f, nullptr);
        RC(ec_gemm_bf16(hbuf, wqkv, bqkv, nullptr, qkv, (int)rows, 3 * D, D, EC_ACT_NONE, stream));
        }
        if (!general)
            hipLaunchKernelGGL(mha_kernel, dim3((unsigned)(B * heads)), dim3(64), 0, s, qkv, att, L, D, heads, 0.125f);
        else if (causal)
            hipLaunchKernelGGL(mha_general_kernel<true>, dim3((unsigned)(B * heads)), dim3(256), mha_general_lds(L), s, qkv, att,
                               L, D, heads, 0.125f);
        else
            hipLaunchKernelGGL(mha_general_kernel<false>, dim3((unsigned)(B * heads)), dim3(256), mha_general_lds(L), s, qkv, att,
                               L, D, heads, 0.125f);
        if (folded) {
            RC(ec_gemm_bf16_ln8(att, wo, bo, x, x, (int)rows, D, D, EC_ACT_NONE, nullptr, nullptr, 0, stats, &np, stream));   // x += out_proj(...); + ln_2's records
            RC(ec_gemm_bf16_ln8(x, wg + 3 * Dz * Dz, fg + 10 * Dz, nullptr, mlp, (int)rows, 4 * D, D, EC_ACT_QUICKGELU, fg + 6 * Dz, stats, np,
                                nullptr, nullptr, stream));
            RC(ec_gemm_bf16_ln8(mlp, wpr, bpr, x, x, (int)rows, D, 4 * D, EC_ACT_NONE, nullptr, nullptr, 0, stats, &np, stream));   // x += c_proj(...); + the next ln_1's records
            continue;
        }
        RC(ec_gemm_bf16(att, wo, bo, x, x, (int)rows, D, D, EC_ACT_NONE, stream));          // x += out_proj(...)
        hipLaunchKernelGGL(layernorm_kernel<0>, dim3(lnb), dim3(256), 0, s, x, nullptr, nullptr, ln2w, ln2b, hbuf, rows, D,
                           L, 1e-5f, nullptr);
        RC(ec_gemm_bf16(hbuf, wfc, bfc, nullptr, mlp, (int)rows, 4 * D, D, EC_ACT_QUICKGELU, stream));
        RC(ec_gemm_bf16(mlp, wpr, bpr, x, x, (int)rows, D, 4 * D, EC_ACT_NONE, stream));     // x += c_proj(...)
    }
    return EC_OK;
}
}  // namespace

extern "C" int ec_vit_create(ec_vit_t** out, int width, int layers_run, int heads, int patch, int input_resolution,
                             const void* w_bf16, size_t n_w, const float* params_f32, size_t n_f) {
    if (!out || !w_bf16 || !params_f32) return EC_ERR_ARG;
    if (width % 64 != 0 || width > 1024 || width / heads != 64 || input_resolution % patch != 0 || (patch * 3) % 4 != 0)
        return EC_ERR_SHAPE;
    const int G = input_resolution / patch, L = G * G + 1;
    if (L > MHA_GENERAL_MAX_TOKENS) return EC_ERR_SHAPE;   // <= 64 tokens: MFMA core; above: general LDS core
    const size_t D = width, Kp = (size_t)patch * patch * 3;
    const size_t need_w = D * Kp + (size_t)layers_run * (3 * D * D + D * D + 4 * D * D + 4 * D * D);
    const size_t need_f = D + L * D + 2 * D + (size_t)layers_run * (2 * D + 3 * D + D + 2 * D + 4 * D + D);
    if (n_w != need_w || n_f != need_f) return EC_ERR_SHAPE;
    ec_vit* h = new (std::nothrow) ec_vit();
    if (!h) return EC_ERR_ALLOC;
    h->width = width; h->layers = layers_run; h->heads = heads; h->patch = patch; h->res = input_resolution;
    h->grid = G; h->L = L; h->w = (const uint16_t*)w_bf16; h->f = params_f32; h->n_w = n_w; h->n_f = n_f;
    {   // LayerNorm-folded copies of the blocks' QKV / c_fc weights (the weights must be on the device by now: they are read here)
        const int rc = build_lnfold(h->fold, h->w + D * Kp, h->f + D + L * D + 2 * D, layers_run, width);
        if (rc != EC_OK) { delete h; return rc; }
    }
    *out = h;
    return EC_OK;
}
extern "C" void ec_vit_destroy(ec_vit_t* h) { delete h; }
extern "C" int ec_vit_tokens(const ec_vit_t* h) { return h ? h->L : 0; }
extern "C" int ec_vit_set_conv8_min_tiles(ec_vit_t* h, int n) {
    if (!h) return EC_ERR_ARG;
    h->conv8_min_tiles = n > 0 ? n : 0;
    return EC_OK;
}
// FNV-1a over what fixes the launch plan of ec_vit_forward (geometry, dispatch threshold, library version): the key of
// the PMC summaries under profiles/ (as ec_rn50_plan_hash).
extern "C" uint64_t ec_vit_plan_hash(const ec_vit_t* h) {
    if (!h) return 0;
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](long v) {
        for (int i = 0; i < 8; ++i) { x ^= (uint64_t)((v >> (8 * i)) & 0xff); x *= 1099511628211ull; }
    };
    mix(ec_version()); mix((long)ec_config_hash());
    mix(h->width); mix(h->layers); mix(h->heads); mix(h->patch); mix(h->res); mix(h->L); mix(h->conv8_min_tiles);
    return x;
}

extern "C" size_t ec_vit_workspace_bytes(const ec_vit_t* h, int batch) {
    if (!h || batch <= 0) return 0;
    const size_t D = h->width, Kp = (size_t)h->patch * h->patch * 3, G2 = (size_t)h->grid * h->grid, L = h->L, B = batch;
    return al256(B * G2 * Kp * 2) + al256(B * G2 * D * 2) + al256(B * L * D * 2) + al256(B * L * 3 * D * 2) +
           al256(B * L * D * 2) + al256(B * L * 4 * D * 2) + al256(B * L * 8 * 16);   // (+ the rows' LayerNorm records: 8 x {sum, M2, count, -})
}

extern "C" int ec_vit_forward(const ec_vit_t* h, const float* rgb, int batch, void* workspace, size_t ws_bytes,
                              void* tokens_bf16, ec_stream_t stream) {
    if (!h || !rgb || !workspace || !tokens_bf16) return EC_ERR_ARG;
    if (batch <= 0) return EC_ERR_SHAPE;
    if (ws_bytes < ec_vit_workspace_bytes(h, batch)) return EC_ERR_WORKSPACE;
    const ec_min_tiles_scope mint_scope(h->conv8_min_tiles);   // this handle's dispatch threshold, for this call only
    hipStream_t s = (hipStream_t)stream;
    const int D = h->width, P = h->patch, G = h->grid, L = h->L, B = batch;
    const int Kp = P * P * 3, G2 = G * G;
    if ((long)B * G2 * Kp * 2 >= (1L << 31) || (long)B * L * 4 * D * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    unsigned char* p = (unsigned char*)workspace;
    uint16_t* patches = (uint16_t*)p; p += al256((size_t)B * G2 * Kp * 2);
    uint16_t* pemb = (uint16_t*)p;    p += al256((size_t)B * G2 * D * 2);
    uint16_t* hbuf = (uint16_t*)p;    p += al256((size_t)B * L * D * 2);
    uint16_t* qkv = (uint16_t*)p;     p += al256((size_t)B * L * 3 * D * 2);
    uint16_t* att = (uint16_t*)p;     p += al256((size_t)B * L * D * 2);
    uint16_t* mlp = (uint16_t*)p;     p += al256((size_t)B * L * 4 * D * 2);
    float* stats = (float*)p;
    uint16_t* x = (uint16_t*)tokens_bf16;   // residual stream lives in the output buffer
    const uint16_t* w = h->w;
    const float* f = h->f;
    const float *cls = f, *pos = f + D, *lnpre_w = pos + (size_t)L * D, *lnpre_b = lnpre_w + D;
    f = lnpre_b + D;
    {
        const long total4 = (long)B * G2 * Kp / 4;
        long blocks = (total4 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rgb, patches, h->res, P, G, total4);
    }
    // (through the folded GEMMs' tile chooser when the geometry suits the 8-wave kernel: 192-row tiles for N = D, §4.9)
    if (h->fold.ok && Kp % 64 == 0)
        RC(ec_gemm_bf16_ln8(patches, w, nullptr, nullptr, pemb, B * G2, D, Kp, EC_ACT_NONE, nullptr, nullptr, 0, nullptr, nullptr, stream));
    else
    RC(ec_gemm_bf16(patches, w, nullptr, nullptr, pemb, B * G2, D, Kp, EC_ACT_NONE, stream));
    w += (size_t)D * Kp;
    const long rows = (long)B * L;
    const unsigned lnb = (unsigned)((rows + 3) / 4);
    hipLaunchKernelGGL(layernorm_kernel<1>, dim3(lnb), dim3(256), 0, s, pemb, cls, pos, lnpre_w, lnpre_b, x, rows, D, L,
                       1e-5f, h->fold.ok ? (float4*)stats : nullptr);   // (+ the record of every output row: block 0's folded ln_1)
    RC(run_blocks(w, f, h->layers, h->heads, x, hbuf, qkv, att, mlp, B, L, D, false, &h->fold, stats, 1, stream));
    EC_CHECK_LAUNCH();
    return EC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// CLIP text tower == CLIP.encode_text ([U] openai/CLIP clip/model.py): the goal-embedding source of the zero-shot
// ObjectNav variant (readme_files/zeroshot_objectnav.md:3-8).  Only a handful of goal strings exist, so this runs
// once per experiment to build a [num_goals, embed_dim] table -- correctness path, not a throughput path.
// ---------------------------------------------------------------------------------------------------------------
struct ec_text {
    int width, layers, heads, ctx, vocab, embed;
    const uint16_t* w;
    const float* f;
    ec_lnfold fold;
};

extern "C" int ec_text_create(ec_text_t** out, int width, int layers, int heads, int context_length, int vocab_size,
                              int embed_dim, const void* w_bf16, size_t n_w, const float* params_f32, size_t n_f) {
    if (!out || !w_bf16 || !params_f32) return EC_ERR_ARG;
    if (width % 64 != 0 || width > 1024 || width / heads != 64 || context_length <= 0 ||
        context_length > MHA_GENERAL_MAX_TOKENS || vocab_size <= 0 || embed_dim % 32 != 0 || layers < 0)
        return EC_ERR_SHAPE;
    const size_t D = width;
    const size_t need_w = (size_t)layers * 12 * D * D + (size_t)embed_dim * D;
    const size_t need_f = (size_t)vocab_size * D + (size_t)context_length * D + (size_t)layers * 13 * D + 2 * D;
    if (n_w != need_w || n_f != need_f) return EC_ERR_SHAPE;
    ec_text* h = new (std::nothrow) ec_text();
    if (!h) return EC_ERR_ALLOC;
    h->width = width; h->layers = layers; h->heads = heads; h->ctx = context_length; h->vocab = vocab_size;
    h->embed = embed_dim; h->w = (const uint16_t*)w_bf16; h->f = params_f32;
    {
        const int rc = build_lnfold(h->fold, h->w, h->f + (size_t)vocab_size * D + (size_t)context_length * D, layers, width);
        if (rc != EC_OK) { delete h; return rc; }
    }
    *out = h;
    return EC_OK;
}
extern "C" void ec_text_destroy(ec_text_t* h) { delete h; }

extern "C" size_t ec_text_workspace_bytes(const ec_text_t* h, int batch) {
    if (!h || batch <= 0) return 0;
    const size_t D = h->width, L = h->ctx, B = batch;
    return al256(B * L * D * 2) * 3 + al256(B * L * 3 * D * 2) + al256(B * L * 4 * D * 2) + al256(B * D * 2) +
           al256(B * (size_t)h->embed * 2) + al256(B * L * 8 * 16);
}

extern "C" int ec_text_forward(const ec_text_t* h, const int32_t* tokens, int batch, void* workspace, size_t ws_bytes,
                               float* out, ec_stream_t stream) {
    if (!h || !tokens || !workspace || !out) return EC_ERR_ARG;
    if (batch <= 0) return EC_ERR_SHAPE;
    if (ws_bytes < ec_text_workspace_bytes(h, batch)) return EC_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = h->width, L = h->ctx, B = batch, E = h->embed;
    if ((long)B * L * 4 * D * 2 >= (1L << 31)) return EC_ERR_SHAPE;
    unsigned char* p = (unsigned char*)workspace;
    uint16_t* x = (uint16_t*)p;    p += al256((size_t)B * L * D * 2);
    uint16_t* hbuf = (uint16_t*)p; p += al256((size_t)B * L * D * 2);
    uint16_t* att = (uint16_t*)p;  p += al256((size_t)B * L * D * 2);
    uint16_t* qkv = (uint16_t*)p;  p += al256((size_t)B * L * 3 * D * 2);
    uint16_t* mlp = (uint16_t*)p;  p += al256((size_t)B * L * 4 * D * 2);
    uint16_t* eot = (uint16_t*)p;  p += al256((size_t)B * D * 2);
    uint16_t* emb = (uint16_t*)p;  p += al256((size_t)B * E * 2);
    float* stats = (float*)p;
    const float* f = h->f;
    const float *tok_emb = f, *pos = f + (size_t)h->vocab * D;
    f = pos + (size_t)L * D;
    const uint16_t* w = h->w;
    const long rows = (long)B * L;
    const long n2 = rows * (D / 2);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, tokens, tok_emb, pos, x, rows, L,
                       D, h->vocab);
    if (h->fold.ok)   // the embedding rows' LayerNorm records for block 0's folded ln_1
        hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, (float4*)stats, rows, D);
    RC(run_blocks(w, f, h->layers, h->heads, x, hbuf, qkv, att, mlp, B, L, D, true, &h->fold, stats, 1, stream));
    hipLaunchKernelGGL(eot_layernorm_kernel, dim3((unsigned)B), dim3(64), 0, s, tokens, x, f, f + D, eot, L, D, 1e-5f);
    RC(ec_gemm_bf16(eot, w, nullptr, nullptr, emb, B, E, D, EC_ACT_NONE, stream));          // @ text_projection
    RC(ec_bf16_to_f32(emb, out, B, E, E, stream));
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_bf16_to_f32(const void* in, float* out, long rows, long row_len, long in_stride, ec_stream_t stream) {
    if (!in || !out) return EC_ERR_ARG;
    if (rows <= 0 || row_len <= 0) return EC_ERR_SHAPE;
    const long n = rows * row_len;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)in, out, n, in_stride, row_len);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" size_t ec_attnpool_workspace_bytes(int batch, int HW, int C) {
    if (batch <= 0 || HW <= 0 || C <= 0) return 0;
    const size_t B = batch, L = HW + 1;
    return al256(B * L * C * 2) + al256(B * C * 2) + al256(B * L * 2 * C * 2) + al256(B * C * 2) + al256(B * C * 2) +
           al256(B * (size_t)C * 2);
}

extern "C" int ec_attnpool_forward(const void* feat, int batch, int HW, int C, int heads, int out_dim, const float* pos,
                                   const void* wq, const float* bq, const void* wkv, const float* bkv, const void* wc,
                                   const float* bc, void* workspace, size_t ws_bytes, float* out, ec_stream_t stream) {
    if (!feat || !pos || !wq || !bq || !wkv || !bkv || !wc || !bc || !workspace || !out) return EC_ERR_ARG;
    if (batch <= 0 || HW + 1 > ATTNPOOL_MAX_L || C / heads != 64 || C % 64 != 0 || out_dim % 32 != 0 || out_dim > C) return EC_ERR_SHAPE;
    if (ws_bytes < ec_attnpool_workspace_bytes(batch, HW, C)) return EC_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t B = batch, L = HW + 1;
    unsigned char* p = (unsigned char*)workspace;
    uint16_t* tok = (uint16_t*)p; p += al256(B * L * C * 2);
    uint16_t* cls = (uint16_t*)p; p += al256(B * C * 2);
    uint16_t* kv = (uint16_t*)p;  p += al256(B * L * 2 * C * 2);
    uint16_t* q = (uint16_t*)p;   p += al256(B * C * 2);
    uint16_t* att = (uint16_t*)p; p += al256(B * C * 2);
    uint16_t* ob = (uint16_t*)p;
    hipLaunchKernelGGL(attnpool_tokens_kernel, dim3((unsigned)B), dim3(256), 0, s, (const uint16_t*)feat, pos, tok, cls, HW, C);
    RC(ec_gemm_bf16(tok, wkv, bkv, nullptr, kv, (int)(B * L), 2 * C, C, EC_ACT_NONE, stream));
    RC(ec_gemm_bf16(cls, wq, bq, nullptr, q, (int)B, C, C, EC_ACT_NONE, stream));
    hipLaunchKernelGGL(attnpool_core_kernel, dim3((unsigned)(B * heads)), dim3(64), 0, s, q, kv, att, (int)L, C, heads,
                       0.125f);
    RC(ec_gemm_bf16(att, wc, bc, nullptr, ob, (int)B, out_dim, C, EC_ACT_NONE, stream));
    return ec_bf16_to_f32(ob, out, (long)B, out_dim, out_dim, stream);
}
