// GRU actor-critic policy: forward and hand-written backward.
//
// Replaces ``ActorCriticModel.forward`` as ``ResnetTensorObjectNavActorCritic``
// and the autograd graph torch builds behind ``total_loss.backward()``
// ([U] allenai/allenact ~v0.5.0: projects/objectnav_baselines/models/
// object_nav_models.py ResnetTensorGoalEncoder; allenact/embodiedai/models/
// basic_models.py RNNStateEncoder, LinearActorHead, LinearCriticHead;
// launched by the reference at readme_files/baselines_robothor_objectnav.md:48-51;
// SURVEY.md §8a a11-a14).
//
// Data layout (all fp32 except the frozen features):
//   feat  [T*N, S, C]      NHWC rows of the rollout feature buffer (bf16 or fp32)
//   params one flat fp32 buffer in AllenAct parameter order (ec_policy_param_offset)
//   grads  one flat fp32 buffer, same offsets (one all-reduce bucket)
// The goal embedding is folded into a [num_goals, 128] row-group bias table
// E1 = embed_class @ W3[:, 32:]^T + b3, so the 64-channel concat is never built.
#include <stdlib.h>

#include <mutex>
#include <new>
#include <unordered_map>

#include "common.h"

extern "C" int ec_dw_tn_x3_splits(long M, int NX);
extern "C" int ec_dw_tn_x3(const void* dYplanes, const void* X, float* part, float* dW, long M, int NX, ec_stream_t stream);
int ec_dw_tn_xp(const void* dYplanes, const void* X, float* part, float* dW, long M, int NX, int planes, ec_stream_t stream);   // dw_tn.hip
int ec_gemm_bf16a_xp(const void* A, const void* Wplanes, const float* bias, float* out, long M, int N, int K, int act, int planes,
                     ec_stream_t stream);                                                                                        // conv_igemm.hip
extern "C" int ec_split3_bf16(const float* W, void* planes, long rows, int K, ec_stream_t stream);
extern "C" int ec_gemm_bf16a_x3(const void* A, const void* Wplanes, const float* bias, float* out, long M, int N, int K,
                                int act, ec_stream_t stream);
extern "C" int ec_gemm_f32(const void* A, const void* B, float* Cp, int M, int N, int K, long sam, long sak, long sbk,
                           long sbn, int ldc, int flags, const float* bias, const float* gbias, const int* gidx,
                           int group, const float* dmask, const float* rowscale, int splitk, ec_stream_t stream);

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void goal_to_i32_kernel(const long long* g, int* o, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = (int)g[i];
}

// x[b, c*S + p] = x4[(b*S + p)*Cc + c]   (channel-major flatten of the combiner output)
// (ldx / xcol: row pitch and first column of this stream's block inside x -- the dual RGB + depth encoder concatenates
//  its two streams' channel-major blocks, [U] ResnetDualTensorGoalEncoder: torch.cat([rgb_x, depth_x], dim=1) then flatten)
__global__ void to_cmajor_kernel(const float* __restrict__ x4, float* __restrict__ x, int S, int Cc, long total, int ldx,
                                 int xcol) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int per = S * Cc;
    const long b = i / per;
    const int r = (int)(i - b * per);
    const int c = r / S, p = r - c * S;
    x[b * ldx + xcol + r] = x4[(b * S + p) * Cc + c];
}
__global__ void from_cmajor_kernel(const float* __restrict__ dx, float* __restrict__ dx4, int S, int Cc, long total, int ldx,
                                   int xcol) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int per = S * Cc;
    const long b = i / per;
    const int r = (int)(i - b * per);
    const int p = r / Cc, c = r - p * Cc;
    dx4[i] = dx[b * ldx + xcol + c * S + p];
}

// EC_WIH_PERM (learn pass): instead of re-ordering the [T*N x 1568] activations between the combiner's pixel-major rows and
// nn.Flatten's channel-major order (to_cmajor / from_cmajor: 2 x 206 MB of strided traffic per optimiser step, 0.5 ms),
// the 9.6-MB weight_ih is re-ordered once per step: mode 0  out[r, p*Cc + c] = in[r, c*S + p];  mode 1 (its gradient,
// back to the parameter's order)  out[r, c*S + p] += in[r, p*Cc + c].  One row per workgroup, staged through LDS.
__global__ __launch_bounds__(256) void permute_row_kernel(const float* __restrict__ in, float* __restrict__ out, int S, int Cc,
                                                         int mode) {
    extern __shared__ float prow[];
    const int flat = S * Cc;
    const long base = (long)blockIdx.x * flat;
    for (int i = threadIdx.x; i < flat; i += 256) prow[i] = in[base + i];
    __syncthreads();
    if (mode == 0) {
        for (int i = threadIdx.x; i < flat; i += 256) out[base + i] = prow[(i % Cc) * S + i / Cc];
    } else {
        for (int i = threadIdx.x; i < flat; i += 256) out[base + i] += prow[(i % S) * Cc + i / S];
    }
}

// torch.nn.GRU cell with the RNNStateEncoder episode mask:
//   hp = m*h_prev;  r = s(gi_r + m*gh_r + b_hr);  z = s(gi_z + m*gh_z + b_hz)
//   hn = m*gh_n + b_hn;  n = tanh(gi_n + r*hn);  h = (1-z)*n + z*hp
// (gh = h_prev W_hh^T without bias; the mask commutes with the row-wise GEMM).
__global__ void gru_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                     const float* __restrict__ bhh, const float* __restrict__ hprev,
                                     const float* __restrict__ mask, float* __restrict__ hout,
                                     float* __restrict__ gates, float* __restrict__ hn_s, float* __restrict__ hp_s,
                                     int N, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * H) return;
    const int n = i / H, j = i - n * H;
    const float m = mask[n];
    const float* gir = gi + (long)n * 3 * H;
    const float* ghr = gh + (long)n * 3 * H;
    const float hp = m * hprev[i];
    const float r = sigmoidf_(gir[j] + m * ghr[j] + bhh[j]);
    const float z = sigmoidf_(gir[H + j] + m * ghr[H + j] + bhh[H + j]);
    const float hn = m * ghr[2 * H + j] + bhh[2 * H + j];
    const float nn = tanhf(gir[2 * H + j] + r * hn);
    hout[i] = (1.f - z) * nn + z * hp;
    if (gates) {
        float* g = gates + (long)n * 3 * H;
        g[j] = r; g[H + j] = z; g[2 * H + j] = nn;
        hn_s[i] = hn;
        hp_s[i] = hp;
    }
}

// Fused GRU forward step (round 2): recurrent projection + gate math in ONE launch per time step, replacing the
// [N x 3H x H] GEMM launch + gru_gates_fwd_kernel pair (17 + 7 us per step at N = 128, 128 steps per direction).
// Workgroup (blockIdx.x, blockIdx.y) owns hidden units j0..j0+7 (24 gate columns of W_hh) of actors n0..n0+31:
//   * hp = m * h_prev tile [32 x H] and the 24 W_hh rows [24 x H] are staged in LDS (row pitch H + 4 floats: the
//     16-lane groups of a ds_read_b128 hit 16 different 16-byte slots);
//   * wave w contracts its quarter of K with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 == an fmaf chain, so the
//     result stays within fp32 rounding of the reference GRU): lane (i = l & 31, hh = l >> 5) feeds k = kb + 4 hh + q
//     to the q-th MFMA of a k-block of 8, one ds_read_b128 per operand per 4 MFMAs;
//   * the four partial 32 x 32 tiles are summed through LDS in a fixed order (actor-local, batch-invariant: act steps
//     stay bit-reproducible whatever the slicing), then thread (a, u) applies the gate math of unit j0 + u of actor a.
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                          const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                          const float* __restrict__ mask, float* __restrict__ hout,
                                                          float* __restrict__ gates, float* __restrict__ hn_s,
                                                          float* __restrict__ hp_s, int N, int H, int gi_parts,
                                                          long gi_pstride) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    // K is walked in chunks of KC = 256 (whole K when H is not a multiple of 256): the two tiles then take 66.5 KB, so two
    // workgroups fit a CU and the recurrences of the two actor slices (two HIP streams) overlap instead of queueing
    const int KC = ((H & 255) == 0) ? 256 : H;
    const int P = KC + 4;
    float* sA = gsm;
    float* sB = gsm + 32 * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * 8, n0 = blockIdx.y * 32;
    const int i = lane & 31, hh = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (KC == 256) {                                                    // W rows 24..31 of the B tile: zeros, once
        for (int idx = tid; idx < 8 * 64; idx += 256) {
            const int r = 24 + (idx >> 6), c4 = idx & 63;
            *reinterpret_cast<float4*>(sB + r * P + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // the episode mask is applied to the contraction result (m (h W^T) == (m h) W^T exactly for m in {0, 1})
    for (int k0 = 0; k0 < H; k0 += KC) {
        if (k0) __syncthreads();                                        // every wave is done reading the previous chunk
        if (KC == 256) {
            // LDS-DMA staging: one 1-KiB piece per tile row (256 floats); 56 rows round-robin over the 4 waves
            typedef __attribute__((address_space(3))) void lds_v;
            typedef const __attribute__((address_space(1))) void gbl_v;
            const int uwave = __builtin_amdgcn_readfirstlane(wave);
            for (int r = uwave; r < 56; r += 4) {
                const float* src;
                float* dst;
                if (r < 32) {
                    const int n = min(n0 + r, N - 1);                   // rows past N are never read back
                    src = hprev + (long)n * H + k0 + lane * 4;
                    dst = sA + r * P;
                } else {
                    const int c = r - 32;
                    src = Whh + ((long)(c >> 3) * H + j0 + (c & 7)) * H + k0 + lane * 4;
                    dst = sB + c * P;
                }
                __builtin_amdgcn_global_load_lds((gbl_v*)src, (lds_v*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int q4 = H >> 2;
            for (int idx = tid; idx < 32 * q4; idx += 256) {
                const int r = idx / q4, c4 = idx - r * q4;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                const int n = n0 + r;
                if (n < N) a = *reinterpret_cast<const float4*>(hprev + (long)n * H + c4 * 4);
                if (r < 24) b = *reinterpret_cast<const float4*>(Whh + ((long)(r >> 3) * H + j0 + (r & 7)) * H + c4 * 4);
                *reinterpret_cast<float4*>(sA + r * P + c4 * 4) = a;
                *reinterpret_cast<float4*>(sB + r * P + c4 * 4) = b;
            }
        }
        __syncthreads();
        const int kq = KC >> 2;
        const float* pa = sA + i * P + wave * kq + 4 * hh;
        const float* pb = sB + i * P + wave * kq + 4 * hh;
        for (int kb = 0; kb < kq; kb += 8) {
            const float4 a = *reinterpret_cast<const float4*>(pa + kb);
            const float4 b = *reinterpret_cast<const float4*>(pb + kb);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                       // the operand tiles are dead: LDS becomes the 4 partial [32 x 32] tiles
    float* part = gsm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;     // actor (C/D layout of the 32x32 MFMA), column = lane & 31
        part[(wave * 32 + row) * 32 + i] = acc[r];
    }
    __syncthreads();
    const int a_ = tid >> 3, u = tid & 7;
    const int n = n0 + a_, j = j0 + u;
    if (n >= N) return;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int c = g * 8 + u;
        gh[g] = ((part[(0 * 32 + a_) * 32 + c] + part[(1 * 32 + a_) * 32 + c]) + part[(2 * 32 + a_) * 32 + c]) +
                part[(3 * 32 + a_) * 32 + c];
    }
    const float* gir = gi + (long)n * 3 * H;
    float gi_r = gir[j], gi_z = gir[H + j], gi_n = gir[2 * H + j];
    for (int p2 = 1; p2 < gi_parts; ++p2) {        // act step: the input projection arrives as split-K parts (fixed order)
        const float* gp2 = gir + p2 * gi_pstride;
        gi_r += gp2[j]; gi_z += gp2[H + j]; gi_n += gp2[2 * H + j];
    }
    const float m = mask[n];
    const float hp = m * hprev[(long)n * H + j];
    const float r = sigmoidf_(gi_r + m * gh[0] + bhh[j]);
    const float z = sigmoidf_(gi_z + m * gh[1] + bhh[H + j]);
    const float hn = m * gh[2] + bhh[2 * H + j];
    const float nn = tanhf(gi_n + r * hn);
    const long o = (long)n * H + j;
    hout[o] = (1.f - z) * nn + z * hp;
    if (gates) {
        float* gp = gates + (long)n * 3 * H;
        gp[j] = r; gp[H + j] = z; gp[2 * H + j] = nn;
        hn_s[o] = hn;
        hp_s[o] = hp;
    }
}

// reverse step: dh = dhs + dh_carry;  outputs dgi (wrt gi), dghb (wrt m*gh + b_hh),
// dh_carry <- m * dh * z   (the GEMM m*(dghb W_hh) is accumulated on top afterwards)
__global__ void gru_gates_bwd_kernel(const float* __restrict__ dhs, float* __restrict__ dh_carry,
                                     const float* __restrict__ gates, const float* __restrict__ hn_s,
                                     const float* __restrict__ hp_s, const float* __restrict__ mask,
                                     float* __restrict__ dgi, float* __restrict__ dghb, int N, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * H) return;
    const int n = i / H, j = i - n * H;
    const float* g = gates + (long)n * 3 * H;
    const float r = g[j], z = g[H + j], nn = g[2 * H + j];
    const float hn = hn_s[i], hp = hp_s[i];
    const float dh = dhs[i] + dh_carry[i];
    const float dn_pre = dh * (1.f - z) * (1.f - nn * nn);
    const float dz_pre = dh * (hp - nn) * z * (1.f - z);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    float* a = dgi + (long)n * 3 * H;
    float* b = dghb + (long)n * 3 * H;
    a[j] = dr_pre; a[H + j] = dz_pre; a[2 * H + j] = dn_pre;
    b[j] = dr_pre; b[H + j] = dz_pre; b[2 * H + j] = dn_pre * r;
    dh_carry[i] = mask[n] * dh * z;
}

// out[c][r] = in[r][c] (fp32, R x C -> C x R): W_hh^T for the fused backward step, once per ec_policy_backward
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) out[(long)c * R + r] = t[tx][ty + 8 * i];
    }
}

// Fused GRU BACKWARD step (round 3): the recurrent back-projection of step t+1 and the gate math of step t in ONE launch,
// replacing the [N x H x 3H] split-K GEMM (fp32 atomics: 17 us) + gru_gates_bwd_kernel (5 us) pair of every step:
//   dh[n, j] = dhs_t[n, j] + carry[n, j] + m_{t+1}[n] * sum_k dghb_{t+1}[n, k] W_hh[k, j]
// (carry = the element-wise part m_{t+1} dh_{t+1} z_{t+1} the previous launch left), then the cell's gate gradients.
// Workgroup (blockIdx.x, blockIdx.y) owns hidden units j0..j0+31 of actors n0..n0+31; K = 3H is walked in 256-wide chunks
// through LDS-DMA exactly as gru_step_fwd_kernel does (A tile = dghb rows, B tile = rows j0.. of W_hh^T), the four waves
// contract a quarter of each chunk on the exact-fp32 MFMA and their partial tiles are summed through LDS in a fixed
// order: no atomics, so the policy gradients are now bit-reproducible run to run.
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dghb_next, const float* __restrict__ WhhT,
                                                          const float* __restrict__ mask_next, const float* __restrict__ dhs,
                                                          float* __restrict__ carry, const float* __restrict__ gates,
                                                          const float* __restrict__ hn_s, const float* __restrict__ hp_s,
                                                          const float* __restrict__ mask, float* __restrict__ dgi,
                                                          float* __restrict__ dghb, int N, int H) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    constexpr int KC = 256, P = KC + 4;
    const int K = 3 * H;
    float* sA = gsm;
    float* sB = gsm + 32 * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int i = lane & 31, hh = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0) __syncthreads();
        typedef __attribute__((address_space(3))) void lds_v;
        typedef const __attribute__((address_space(1))) void gbl_v;
        const int uwave = __builtin_amdgcn_readfirstlane(wave);
        for (int r = uwave; r < 64; r += 4) {                      // one 1-KiB LDS-DMA piece per tile row
            const float* src;
            float* dst;
            if (r < 32) {
                const int n = min(n0 + r, N - 1);                  // rows past N are never read back
                src = dghb_next + (long)n * K + k0 + lane * 4;
                dst = sA + r * P;
            } else {
                src = WhhT + (long)(j0 + r - 32) * K + k0 + lane * 4;
                dst = sB + (r - 32) * P;
            }
            __builtin_amdgcn_global_load_lds((gbl_v*)src, (lds_v*)dst, 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* pa = sA + i * P + wave * (KC >> 2) + 4 * hh;
        const float* pb = sB + i * P + wave * (KC >> 2) + 4 * hh;
#pragma unroll
        for (int kb = 0; kb < (KC >> 2); kb += 8) {
            const float4 a = *reinterpret_cast<const float4*>(pa + kb);
            const float4 b = *reinterpret_cast<const float4*>(pb + kb);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                       // the operand tiles are dead: LDS becomes the 4 partial [32 x 32] tiles
    float* part = gsm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;     // actor (C/D layout of the 32x32 MFMA), column = lane & 31
        part[(wave * 32 + row) * 32 + i] = acc[r];
    }
    __syncthreads();
    const int a_ = tid >> 3, u0 = tid & 7;
    const int n = n0 + a_;
    if (n >= N) return;
    const float mn = mask_next[n], m = mask[n];
    const float* g = gates + (long)n * 3 * H;
    float* ga = dgi + (long)n * 3 * H;
    float* gb = dghb + (long)n * 3 * H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = u0 + 8 * q, j = j0 + c;
        const float proj = ((part[(0 * 32 + a_) * 32 + c] + part[(1 * 32 + a_) * 32 + c]) + part[(2 * 32 + a_) * 32 + c]) +
                           part[(3 * 32 + a_) * 32 + c];
        const long o = (long)n * H + j;
        const float r = g[j], z = g[H + j], nn = g[2 * H + j];
        const float hn = hn_s[o], hp = hp_s[o];
        const float dh = dhs[o] + (carry[o] + mn * proj);
        const float dn_pre = dh * (1.f - z) * (1.f - nn * nn);
        const float dz_pre = dh * (hp - nn) * z * (1.f - z);
        const float dr_pre = dn_pre * hn * r * (1.f - r);
        ga[j] = dr_pre; ga[H + j] = dz_pre; ga[2 * H + j] = dn_pre;
        gb[j] = dr_pre; gb[H + j] = dz_pre; gb[2 * H + j] = dn_pre * r;
        carry[o] = m * dh * z;
    }
}

// ---- 16 x 16-tile GRU step kernels for the UPDATE's recurrences (H == 512, EC_GRU_FUSED >= 2, round 3) ----
// The 32 x 32-tile kernels above put 256 (forward) / 64 (backward) workgroups on the chip and wait for every K chunk with
// nothing else in flight: 15 us a step, 2 x 1024 dependent steps per update.  These two use the 16x16x4 fp32 MFMA so that
// 16 actors x 16 hidden units make a workgroup (256 of them for 128 actors, forward AND backward), and give every wave a
// PRIVATE quarter of K: a wave's LDS-DMA pieces land in its own LDS slice (no workgroup barrier between load and MFMA; the
// wave waits with counted s_waitcnt vmcnt while later pieces are still in flight).  Rows are stored unpadded, 512 B per
// row and wave, with the 16-B units XOR-swizzled by (row & 15) on the GLOBAL side of the DMA (LDS side is contiguous by
// construction), which makes the ds_read_b128 operand reads conflict-free.  The epilogue's operands are fetched before the
// first piece is issued.  Same fp32 arithmetic as above, K summed in a different (fixed) order.
typedef __attribute__((address_space(3))) void gru_lds_v;
typedef const __attribute__((address_space(1))) void gru_gbl_v;

__device__ __forceinline__ f32x4_t mfma16f(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Loads the compiler must neither sink to their first use (the epilogue operands: fetched under the pieces) nor fence with a
// vmcnt(0) of its own (it puts one in front of every ds_read that follows an LDS-DMA): issued as asm, waited for by the
// kernels' counted s_waitcnt, and tied to their consumers through gru_tie().
__device__ __forceinline__ float gru_gload(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x4_t gru_lread(unsigned lds_addr) {
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned gru_lds_addr(const float* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const float*)p;
}
template <int CNT> __device__ __forceinline__ void gru_lwait(f32x4_t& a, f32x4_t& b) {   // ds_reads up to here have returned
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT));
}

__global__ __launch_bounds__(256) void gru_step_fwd512_kernel(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                             const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                             const float* __restrict__ mask, float* __restrict__ hout,
                                                             float* __restrict__ gates, float* __restrict__ hn_s,
                                                             float* __restrict__ hp_s, int N) {
    constexpr int H = 512, KW = 128;                       // KW: the k range of one wave (512 B of every row)
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    float* wl = gsm + wave * (64 * KW);                    // 64 rows (16 actors + 3 x 16 W rows) x 128 floats = 32 KB / wave
    float* part = gsm + 4 * 64 * KW;                       // [wave][gate][16 actors][16 units]
    // epilogue operands first (oldest in the vmcnt order: complete before any piece that is waited for)
    const int a_ = tid >> 4, u_ = tid & 15;
    const int n = min(n0 + a_, N - 1), j = j0 + u_;
    // 32 pieces of 1 KiB: two rows each (lanes 0..31 / 32..63), 8 for the actors' h rows, then 8 per gate
    const int half = lane >> 5, unit = lane & 31;
    const int kw0 = wave * KW;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int r = 2 * p + half;                        // row of this lane
        const float* row;
        if (p < 8) row = hprev + (long)min(n0 + r, N - 1) * H;
        else {
            const int c = r - 16;
            row = Whh + ((long)(c >> 4) * H + j0 + (c & 15)) * H;
        }
        const float* src = row + kw0 + 4 * (unit ^ (r & 15));
        __builtin_amdgcn_global_load_lds((gru_gbl_v*)src, (gru_lds_v*)(wl + 2 * p * KW), 16, 0, 0);
    }
    // the epilogue's operands go LAST (gi is an HBM miss; vmcnt retires in order, so in front they would hold every piece back)
    const float* gir = gi + (long)n * 3 * H;
    float gi_r = gru_gload(gir + j), gi_z = gru_gload(gir + H + j), gi_n = gru_gload(gir + 2 * H + j);
    float m = gru_gload(mask + n);
    float hp_raw = gru_gload(hprev + (long)n * H + j);
    float b_r = gru_gload(bhh + j), b_z = gru_gload(bhh + H + j), b_n = gru_gload(bhh + 2 * H + j);
    const int i = lane & 15, q = lane >> 4;
    const unsigned la = gru_lds_addr(wl + i * KW);
    unsigned sw[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) sw[s2] = 16u * (unsigned)((4 * s2 + q) ^ i);
    f32x4_t av[8], bv[8];
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // 8 scalar loads + 24 pieces may still be in flight
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) av[s2] = gru_lread(la + sw[s2]);
    f32x4_t acc[3], acd[3];                                 // two chains per tile: the 16x16x4 MFMA issues every 32 cycles, depends at 40
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acd[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (g == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (g == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        const unsigned lb = la + (unsigned)((16 + 16 * g) * KW * 4);
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) bv[s2] = gru_lread(lb + sw[s2]);
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            switch (s2) {                                   // (8 reads in flight; the s2-th has returned when 7 - s2 remain)
                case 0: gru_lwait<7>(av[0], bv[0]); break;
                case 1: gru_lwait<6>(av[1], bv[1]); break;
                case 2: gru_lwait<5>(av[2], bv[2]); break;
                case 3: gru_lwait<4>(av[3], bv[3]); break;
                case 4: gru_lwait<3>(av[4], bv[4]); break;
                case 5: gru_lwait<2>(av[5], bv[5]); break;
                case 6: gru_lwait<1>(av[6], bv[6]); break;
                default: gru_lwait<0>(av[7], bv[7]); break;
            }
            acc[g] = mfma16f(av[s2][0], bv[s2][0], acc[g]);
            acd[g] = mfma16f(av[s2][1], bv[s2][1], acd[g]);
            acc[g] = mfma16f(av[s2][2], bv[s2][2], acc[g]);
            acd[g] = mfma16f(av[s2][3], bv[s2][3], acd[g]);
        }
        __builtin_amdgcn_sched_barrier(0);                  // (keep this tile's MFMAs in front of the next tile's piece wait)
    }
    // the scalar loads: landed after this wait; hand the epilogue operands to the compiler
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(gi_r), "+v"(gi_z), "+v"(gi_n), "+v"(m), "+v"(hp_raw), "+v"(b_r), "+v"(b_z), "+v"(b_n));
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((wave * 3 + g) * 16 + 4 * q + r) * 16 + i] = acc[g][r] + acd[g][r];   // (actor 4q + r, unit i)
    __syncthreads();
    if (n0 + a_ >= N) return;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int o = (g * 16 + a_) * 16 + u_;
        gh[g] = ((part[o] + part[3 * 256 + o]) + part[6 * 256 + o]) + part[9 * 256 + o];
    }
    const float hp = m * hp_raw;
    const float r = sigmoidf_(gi_r + m * gh[0] + b_r);
    const float z = sigmoidf_(gi_z + m * gh[1] + b_z);
    const float hn = m * gh[2] + b_n;
    const float nn = tanhf(gi_n + r * hn);
    const long o = (long)n * H + j;
    hout[o] = (1.f - z) * nn + z * hp;
    float* gp = gates + (long)n * 3 * H;
    gp[j] = r; gp[H + j] = z; gp[2 * H + j] = nn;
    hn_s[o] = hn;
    hp_s[o] = hp;
}

// Backward step, same scheme: dh[n, j] = dhs + carry + m_{t+1} sum_k dghb_{t+1}[n, k] W_hh^T[j, k] with K = 3H = 1536; a
// wave owns 384 k, walked as three 128-k sub-chunks through two private 16-KB buffers (the third sub-chunk's pieces are
// issued into buffer 0 as soon as the wave has read sub-chunk 0 out of it).  Two accumulators alternate so that the
// dependent-issue latency of the 16x16x4 MFMA (40 vs 32 cycles) stays hidden.
__global__ __launch_bounds__(256) void gru_step_bwd512_kernel(const float* __restrict__ dghb_next, const float* __restrict__ WhhT,
                                                             const float* __restrict__ mask_next, const float* __restrict__ dhs,
                                                             float* __restrict__ carry, const float* __restrict__ gates,
                                                             const float* __restrict__ hn_s, const float* __restrict__ hp_s,
                                                             const float* __restrict__ mask, float* __restrict__ dgi,
                                                             float* __restrict__ dghb, int N) {
    constexpr int H = 512, K = 3 * H, KS = 128;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    float* wl = gsm + wave * (2 * 32 * KS);                // two buffers of 32 rows x 128 floats
    float* part = gsm + 4 * 2 * 32 * KS;
    const int a_ = tid >> 4, u_ = tid & 15;
    const int n = min(n0 + a_, N - 1), j = j0 + u_;
    const long o = (long)n * H + j;
    const int half = lane >> 5, unit = lane & 31;
    const int kw0 = wave * (3 * KS);
    auto issue = [&](int sc, int buf) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int rr = 2 * p + half;
            const float* row = (p < 8) ? dghb_next + (long)min(n0 + rr, N - 1) * K : WhhT + (long)(j0 + rr - 16) * K;
            const float* src = row + kw0 + sc * KS + 4 * (unit ^ (rr & 15));
            __builtin_amdgcn_global_load_lds((gru_gbl_v*)src, (gru_lds_v*)(wl + buf * (32 * KS) + 2 * p * KS), 16, 0, 0);
        }
    };
    const int i = lane & 15, q = lane >> 4;
    const unsigned la = gru_lds_addr(wl + i * KS);
    unsigned sw[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) sw[s2] = 16u * (unsigned)((4 * s2 + q) ^ i);
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto contract = [&](int buf) {
        const unsigned pa = la + (unsigned)(buf * 32 * KS * 4), pb = pa + 16u * KS * 4;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                    // 2 x (4 A + 4 B reads in flight, then 16 MFMAs)
            f32x4_t av[4], bv[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                av[s2] = gru_lread(pa + sw[4 * hf + s2]);
                bv[s2] = gru_lread(pb + sw[4 * hf + s2]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                switch (s2) {
                    case 0: gru_lwait<6>(av[0], bv[0]); break;
                    case 1: gru_lwait<4>(av[1], bv[1]); break;
                    case 2: gru_lwait<2>(av[2], bv[2]); break;
                    default: gru_lwait<0>(av[3], bv[3]); break;
                }
                acc0 = mfma16f(av[s2][0], bv[s2][0], acc0);
                acc1 = mfma16f(av[s2][1], bv[s2][1], acc1);
                acc0 = mfma16f(av[s2][2], bv[s2][2], acc0);
                acc1 = mfma16f(av[s2][3], bv[s2][3], acc1);
            }
        }
    };
    issue(0, 0);
    issue(1, 1);
    // the epilogue's operands (HBM misses) behind the first two sub-chunks in the in-order vmcnt queue
    const float* g = gates + (long)n * 3 * H;
    float r = gru_gload(g + j), z = gru_gload(g + H + j), nn = gru_gload(g + 2 * H + j);
    float hn = gru_gload(hn_s + o), hp = gru_gload(hp_s + o);
    float dhs_v = gru_gload(dhs + o), carry_v = gru_gload(carry + o);
    float mn = gru_gload(mask_next + n), m = gru_gload(mask + n);
    asm volatile("s_waitcnt vmcnt(25)" ::: "memory");      // sub-chunk 1 (16 pieces) + 9 scalar loads may be in flight
    contract(0);                                           // (ends on lgkmcnt(0): this wave's reads of buffer 0 have returned)
    __builtin_amdgcn_sched_barrier(0);
    issue(2, 0);
    asm volatile("s_waitcnt vmcnt(25)" ::: "memory");      // 9 scalar loads + sub-chunk 2
    contract(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    contract(0);
    asm volatile("" : "+v"(r), "+v"(z), "+v"(nn), "+v"(hn), "+v"(hp), "+v"(dhs_v), "+v"(carry_v), "+v"(mn), "+v"(m));
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) part[(wave * 16 + 4 * q + rr) * 16 + i] = acc0[rr] + acc1[rr];
    __syncthreads();
    if (n0 + a_ >= N) return;
    const int po = a_ * 16 + u_;
    const float proj = ((part[po] + part[256 + po]) + part[512 + po]) + part[768 + po];
    const float dh = dhs_v + (carry_v + mn * proj);
    const float dn_pre = dh * (1.f - z) * (1.f - nn * nn);
    const float dz_pre = dh * (hp - nn) * z * (1.f - z);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    float* ga = dgi + (long)n * 3 * H;
    float* gb = dghb + (long)n * 3 * H;
    ga[j] = dr_pre; ga[H + j] = dz_pre; ga[2 * H + j] = dn_pre;
    gb[j] = dr_pre; gb[H + j] = dz_pre; gb[2 * H + j] = dn_pre * r;
    carry[o] = m * dh * z;
}

// part[blockIdx.y][n] = sum over the block's rows of Y[m*ld + n]   (no atomics: colsum_fold_kernel adds the row blocks in order)
__global__ void colsum_kernel(const float* __restrict__ Y, float* __restrict__ out, long M, int N, int ld,
                              int rows_per_block) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;   // 4 row lanes
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (long r = r0 + sub; r < r1; r += 4) s += Y[r * ld + n];
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && n < N) out[(long)blockIdx.y * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// The same for wide matrices (N % 4 == 0, ld % 4 == 0: the GRU's [T*N x 3H] gate gradients, 100 MB each per optimiser step):
// 16-B loads, 8 of them in flight per lane, 64 x 4 columns per wave and rows_per_block / 4 rows per wave -- HBM-bound
// instead of latency-bound (280 -> ~30 us per call).
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ Y, float* __restrict__ out, long M, int N, int ld,
                                                     int rows_per_block) {
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        const float* base = Y + c;
        long r = r0 + sub;
        for (; r + 28 < r1; r += 32) {
            f32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4_t*>(base + (r + 4 * k) * ld);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; r < r1; r += 4) s += *reinterpret_cast<const f32x4_t*>(base + r * ld);
    }
    __shared__ f32x4_t red[4][64];
    red[sub][lane] = s;
    __syncthreads();
    if (c < N) out[(long)blockIdx.y * N + c + sub] = ((red[0][lane][sub] + red[1][lane][sub]) + red[2][lane][sub]) + red[3][lane][sub];
}

// dE1[goal[b], n] += sum_p dm1[(b*S + p)*N + n]
__global__ void group_sum_scatter_kernel(const float* __restrict__ dm1, const int* __restrict__ goal,
                                         float* __restrict__ dE1, int S, int N, long B) {
    const long b = blockIdx.x;
    if (b >= B) return;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* p = dm1 + b * S * N + n;
        float s = 0.f;
        for (int q = 0; q < S; ++q) s += p[(long)q * N];
        atomicAdd(dE1 + (long)goal[b] * N + n, s);
    }
}

// zero-shot fusion: x[b, :] = feat[b, :] / |feat[b, :]| * table[goal[b], :]   (one wave per row; fp32 statistics)
template <bool BF16>
__global__ __launch_bounds__(256) void fuse_goal_kernel(const void* __restrict__ feat, const float* __restrict__ table,
                                                       const int* __restrict__ goal, float* __restrict__ x, long B, int E,
                                                       int num_goals) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float ss = 0.f;
    for (int k = lane; k < E; k += 64) {
        const float v = BF16 ? ec_bf2f(((const uint16_t*)feat)[row * E + k]) : ((const float*)feat)[row * E + k];
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    int g = goal[row];
    g = g < 0 ? 0 : (g >= num_goals ? num_goals - 1 : g);
    const float* t = table + (long)g * E;
    for (int k = lane; k < E; k += 64) {
        const float v = BF16 ? ec_bf2f(((const uint16_t*)feat)[row * E + k]) : ((const float*)feat)[row * E + k];
        x[row * E + k] = v * inv * t[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused forward "tail" of ResnetTensorGoalEncoder (round 2): for every feature-map pixel (row)
//     c2 = relu(W2 c1 + b2)            128 -> 32    (resnet_compressor.2)
//     m1 = relu(W3[:, :32] c2 + E1[goal])  32 -> 128   (target_obs_combiner.0; the goal half is the row-group bias E1)
//     x4 = W4 m1 + b4                  128 -> 32    (target_obs_combiner.2)
// in ONE pass over c1 instead of three GEMM launches that each stream a [T*N*49, 128] fp32 tensor through HBM with
// 32- or 128-long contractions (0.5-0.9 ms each per slice and epoch against ~0.2 ms of traffic for all three together).
// A wave owns 32-row tiles: the c1 tile is staged in a wave-private LDS image, each stage runs on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, an fmaf chain) with the weights resident in LDS as B operands, and a stage's output tile goes
// back to the wave's LDS image (bias + ReLU applied) as the next stage's A operand -- and to HBM, because the backward
// needs c2 and m1.  No workgroup barrier after the weights are staged.  Geometry fixed to the reference's
// (128, 32, 128, 32); other widths keep the GEMM path.
// Fragment-order copies of the act step's two weight operands (built with the other weight-derived tables, once per
// parameter update): a fragment = what one MFMA operand load of a wave fetches = 64 lanes x 16 B = ONE contiguous 1-KB unit.
// Out of the row-major tables a fragment load touched 32 rows x 32 B -- 64 quarter-used sectors per instruction -- and the
// act kernels ran at the resulting L2 -> L1 rate (gi 16-19 us, c1 19 us at 32 actors).
//   fp32 [N][K]  (gi):  unit (cb * K/8 + g) * 64 + l  =  row 32 cb + (l & 31), k = 8 g + 4 (l >> 5) .. + 3
//   bf16 planes [N][3][K] (c1):  unit ((cb * K/16 + ks) * 3 + pl) * 64 + l  =  row 32 cb + (l & 31), plane pl, k = 16 ks + 8 (l >> 5) .. + 7
__global__ void frag_pack_f32_kernel(const float4* __restrict__ W, float4* __restrict__ F, int N, int K) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)N * K / 4) return;
    const int l = (int)(t & 63);
    const long f = t >> 6;
    const int G = K / 8, g = (int)(f % G), cb = (int)(f / G);
    F[t] = W[((long)(cb * 32 + (l & 31)) * K + g * 8 + 4 * (l >> 5)) >> 2];
}
__global__ void frag_pack_planes_kernel(const uint4* __restrict__ P, uint4* __restrict__ F, int N, int K) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)N * 3 * K / 8) return;
    const int l = (int)(t & 63);
    const long f = t >> 6;
    const int pl = (int)(f % 3);
    const long f2 = f / 3;
    const int KS = K / 16, ks = (int)(f2 % KS), cb = (int)(f2 / KS);
    F[t] = P[(((long)(cb * 32 + (l & 31)) * 3 + pl) * K + ks * 16 + 8 * (l >> 5)) >> 3];
}

// Act step (T = 1): the GRU's input projection gi[N][3H] = x[N][K] W[3H][K]^T + b for a handful of actor rows (N = 32..128),
// K = 1568.  As a tiled GEMM this is 24-96 workgroups walking K in series behind a split-K fold (18 us at 32 actors);
// here a workgroup owns a 32 x 32 output tile and its NW waves SPLIT K (224 each for K = 1568 = 7 x 224), fragments straight
// from global memory as float4 (exact-fp32 MFMA 32x32x2: an operand is one element per lane, so a float4 feeds four MFMAs),
// the NW partial tiles folded through LDS in wave order + bias: deterministic, no atomics, no partial matrices in HBM.
template <int NW>
__global__ __launch_bounds__(NW * 64) void gi_act_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ gi, int N, int K,
                                                         int NO) {
    __shared__ float part[NW][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int kw = K / NW, k0 = wave * kw;
    const float* pa = x + (long)min(n0 + i, N - 1) * K + k0 + 4 * hh;        // (rows past N are computed on a clamped row, never stored)
    // W in fragment order (frag_pack_f32_kernel): group g of column block cb is the contiguous 1-KB unit (cb K/8 + g)
    const f32x4_t* pbf = reinterpret_cast<const f32x4_t*>(W) + ((long)blockIdx.x * (K / 8) + k0 / 8) * 64 + lane;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 7;                                          // groups of 8 k in flight
    for (int kb = 0; kb < kw; kb += 8 * U) {
        f32x4_t a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = min(kb + 8 * u, kw - 8);                // (kw is a multiple of 8; a clamped group is skipped below)
            a[u] = *reinterpret_cast<const f32x4_t*>(pa + kk);
            b[u] = pbf[(kk >> 3) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kb + 8 * u < kw) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][0], b[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][1], b[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][2], b[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][3], b[u][3], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * hh][i] = acc[r];     // C/D layout: row = actor, column = lane & 31
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += NW * 64) {
        const int row = e >> 5, col = e & 31;
        float v = part[0][row][col];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) v += part[w2][row][col];
        if (n0 + row < N) gi[(long)(n0 + row) * NO + j0 + col] = v + (bias ? bias[j0 + col] : 0.f);
    }
}

// Act step: the compressor's first 1x1 conv over the slice's bf16 feature rows, c1[M][NO] = relu(feat[M][K] W1[NO][K]^T + b1)
// (M = 49 n rows, K = 2048, NO = 128), exact-fp32 as everywhere in the policy: W1 arrives as three bf16 planes
// ([NO][3][K], ec_split3_bf16, cached in the act workspace with the other weight-derived tables) and each k-step runs the
// three plane products, lowest plane first, into one accumulator.  A workgroup owns a 32 x 32 output tile and its eight
// waves split K (fragments straight from global memory); the partial tiles are folded through LDS in wave order.  Replaces
// a split-K GEMM whose four partial matrices tail_fwd_kernel had to fold.
template <int MBLK, int U>
__global__ __launch_bounds__(512) void c1_act_kernel(const uint16_t* __restrict__ feat, const uint16_t* __restrict__ Wp,
                                                     const float* __restrict__ bias, float* __restrict__ c1, long M, int K, int NO) {
    // MBLK 32-row blocks per workgroup (the W slice is fetched once per workgroup: taller tiles for larger M), U k-steps of
    // fragment loads in flight per wave (the loop is L2-latency-bound: few, deep rounds)
    constexpr int NW = 8;
    __shared__ float part[NW][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * 32;
    const long m0 = (long)blockIdx.y * (32 * MBLK);
    const int kw = K / NW, k0 = wave * kw;
    const uint16_t* pa[MBLK];
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) pa[mb] = feat + min(m0 + mb * 32 + i, M - 1) * K + k0 + 8 * hh;
    // W1's planes in fragment order (frag_pack_planes_kernel): (k-step ks, plane pl) of column block cb is one 1-KB unit
    const s16x8_t* pbf = reinterpret_cast<const s16x8_t*>(Wp) + ((long)blockIdx.x * (K / 16) + k0 / 16) * 3 * 64 + lane;
    f32x16_t acc[MBLK];
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    for (int kb = 0; kb < kw; kb += 16 * U) {
        s16x8_t a[U][MBLK], b[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = min(kb + 16 * u, kw - 16);
#pragma unroll
            for (int mb = 0; mb < MBLK; ++mb) a[u][mb] = *reinterpret_cast<const s16x8_t*>(pa[mb] + kk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[u][pl] = pbf[((kk >> 4) * 3 + pl) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kb + 16 * u < kw) {
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int mb = 0; mb < MBLK; ++mb)
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[u][mb]),
                                                                          __builtin_bit_cast(bf16x8_t, b[u][pl]), acc[mb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) {
        if (mb) __syncthreads();                                  // the previous block's fold is done reading `part`
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * hh][i] = acc[mb][r];
        __syncthreads();
        for (int e = tid; e < 32 * 32; e += NW * 64) {
            const int row = e >> 5, col = e & 31;
            float v = part[0][row][col];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) v += part[w2][row][col];
            const long m = m0 + mb * 32 + row;
            if (m < M) c1[m * NO + j0 + col] = fmaxf(v + bias[j0 + col], 0.f);
        }
    }
}

constexpr int ACT_PARTS = 4, ACT_MAX_ROWS = 16384;   // act step: split-K factor of the two long-K GEMMs / row limit of that path
constexpr int TL_P128 = 132, TL_P32 = 36;     // LDS row pitches (floats): 16-byte slots of 16 consecutive rows differ

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tail_fwd_kernel(const float* __restrict__ c1, const float* __restrict__ W2,
                                                         const float* __restrict__ b2, const float* __restrict__ W3,
                                                         int w3_ld, const float* __restrict__ E1,
                                                         const int* __restrict__ goal, int S, int num_goals,
                                                         const float* __restrict__ W4, const float* __restrict__ b4,
                                                         float* __restrict__ c2, float* __restrict__ m1,
                                                         float* __restrict__ x4, long M, int c1_parts, long c1_pstride,
                                                         const float* __restrict__ b1, int gstride) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float* sW2 = tsm;                            // [32][132]
    float* sW3 = sW2 + 32 * TL_P128;             // [128][36]
    float* sW4 = sW3 + 128 * TL_P32;             // [32][132]
    float* sE1 = sW4 + 32 * TL_P128;             // [num_goals][128]
    float* sB = sE1 + num_goals * 128;           // b2[32], b4[32]
    float* wave_base = sB + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* T1 = wave_base + wave * (32 * TL_P128 + 32 * TL_P32);   // c1 tile, later the m1 tile
    float* T2 = T1 + 32 * TL_P128;                                 // c2 tile
    // (all of a thread's weight loads in flight before its first LDS store: ec_stage_all, common.h -- the act step's
    //  instance of this kernel is 13-25 workgroups whose prologue is on the rollout's critical chain)
    ec_stage_all<32 * 32, 256, float4>(tid,                        // W2: [32][128]
        [&](int idx) { return *reinterpret_cast<const float4*>(W2 + (idx >> 5) * 128 + (idx & 31) * 4); },
        [&](int idx, const float4& v) { *reinterpret_cast<float4*>(sW2 + (idx >> 5) * TL_P128 + (idx & 31) * 4) = v; });
    ec_stage_all<32 * 32, 256, float4>(tid,                        // W4: [32][128]
        [&](int idx) { return *reinterpret_cast<const float4*>(W4 + (idx >> 5) * 128 + (idx & 31) * 4); },
        [&](int idx, const float4& v) { *reinterpret_cast<float4*>(sW4 + (idx >> 5) * TL_P128 + (idx & 31) * 4) = v; });
    ec_stage_all<128 * 8, 256, float4>(tid,                        // W3[:, :32]: [128][32] out of rows of w3_ld floats
        [&](int idx) { return *reinterpret_cast<const float4*>(W3 + (long)(idx >> 3) * w3_ld + (idx & 7) * 4); },
        [&](int idx, const float4& v) { *reinterpret_cast<float4*>(sW3 + (idx >> 3) * TL_P32 + (idx & 7) * 4) = v; });
    for (int idx = tid; idx < num_goals * 32; idx += 256)
        *reinterpret_cast<float4*>(sE1 + idx * 4) = *reinterpret_cast<const float4*>(E1 + idx * 4);
    if (tid < 32) { sB[tid] = b2[tid]; sB[32 + tid] = b4[tid]; }
    __syncthreads();

    const int i = lane & 31, hh = lane >> 5;
    const long ntiles = (M + 31) / 32;
    const long tstep = (long)gridDim.x * 4;
    // c1 tiles are fetched one tile ahead into registers (16 float4 per lane: rows past M clamp to the last row)
    // (plain unrolled loops, no lambda: an array captured by a lambda lands in scratch memory)
    f32x4_t st[16];                                                // native vector type: HIP's float4 struct would be memcpy'd through scratch
    long tile = (long)blockIdx.x * 4 + wave;
#pragma unroll
    for (int q = 0; q < 16; ++q) {                                 // (unconditional, clamped: keeps st[] in registers)
        const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
        st[q] = *reinterpret_cast<const f32x4_t*>(c1 + min(tile * 32 + r, M - 1) * 128 + c4 * 4);
    }
    // act step (c1_parts > 1): c1 arrives as split-K partial sums WITHOUT bias / ReLU -- folded here in a fixed order
    // (unconditional code on both paths: with one part the loop below does nothing and b1v stays 0 / relu off)
    const bool fold = c1_parts > 1;
    f32x4_t b1v = {0.f, 0.f, 0.f, 0.f};
    if (fold) b1v = *reinterpret_cast<const f32x4_t*>(b1 + (lane & 31) * 4);
    auto fold_parts = [&](f32x4_t v, long row, int c4) {
        for (int p2 = 1; p2 < c1_parts; ++p2) v += *reinterpret_cast<const f32x4_t*>(c1 + p2 * c1_pstride + row * 128 + c4 * 4);
        v += b1v;
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        return v;
    };
    if (fold) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            st[q] = fold_parts(st[q], min(tile * 32 + r, M - 1), c4);
        }
    }
    for (; tile < ntiles; tile += tstep) {
        const long m0 = tile * 32;
        const long grp_base = m0 / S;                               // wave-uniform
        const int grp_rem = (int)(m0 - grp_base * S);
        const int last_row = (int)min((long)31, M - 1 - m0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<f32x4_t*>(T1 + r * TL_P128 + c4 * 4) = st[q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {                             // next tile: in flight under this tile's three contractions
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            st[q] = *reinterpret_cast<const f32x4_t*>(c1 + min((tile + tstep) * 32 + r, M - 1) * 128 + c4 * 4);
        }
        if (fold) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
                st[q] = fold_parts(st[q], min((tile + tstep) * 32 + r, M - 1), c4);
            }
        }
        // ---- c2 = relu(c1 W2^T + b2) ----
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kb = 0; kb < 128; kb += 8) {
            const float4 a = *reinterpret_cast<const float4*>(T1 + i * TL_P128 + kb + 4 * hh);
            const float4 b = *reinterpret_cast<const float4*>(sW2 + i * TL_P128 + kb + 4 * hh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        {
            const float bj = sB[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;     // C/D layout: column = lane & 31
                T2[row * TL_P32 + i] = fmaxf(acc[r] + bj, 0.f);
            }
        }
        // the tile leaves through the wave's LDS image as 16-byte row chunks (a lane of the C/D layout holds single floats)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
            if (m0 + r < M)
                *reinterpret_cast<float4*>(c2 + (m0 + r) * 32 + c4 * 4) = *reinterpret_cast<const float4*>(T2 + r * TL_P32 + c4 * 4);
        }
        // ---- m1 = relu(c2 W3a^T + E1[goal]) : four 32-column tiles, K = 32 ----
        f32x16_t am[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) am[jt][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 32; kb += 8) {
            const float4 a = *reinterpret_cast<const float4*>(T2 + i * TL_P32 + kb + 4 * hh);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const float4 b = *reinterpret_cast<const float4*>(sW3 + (jt * 32 + i) * TL_P32 + kb + 4 * hh);
                am[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, am[jt], 0, 0, 0);
                am[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, am[jt], 0, 0, 0);
                am[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, am[jt], 0, 0, 0);
                am[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, am[jt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            // row group (actor-step) of this row in 32-bit arithmetic: a 64-bit division per element would cost more than the MFMAs
            const unsigned trow = min((unsigned)(grp_rem + row), (unsigned)(grp_rem + last_row));
            int g = goal[(grp_base + (long)(trow / (unsigned)S)) * gstride];   // (gstride 2: the low words of the caller's int64 ids)
            g = g < 0 ? 0 : (g >= num_goals ? num_goals - 1 : g);
            const float* e = sE1 + g * 128 + i;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                T1[row * TL_P128 + jt * 32 + i] = fmaxf(am[jt][r] + e[jt * 32], 0.f);   // (the c1 tile is dead by now)
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            if (m0 + r < M)
                *reinterpret_cast<float4*>(m1 + (m0 + r) * 128 + c4 * 4) = *reinterpret_cast<const float4*>(T1 + r * TL_P128 + c4 * 4);
        }
        // ---- x4 = m1 W4^T + b4 ----
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kb = 0; kb < 128; kb += 8) {
            const float4 a = *reinterpret_cast<const float4*>(T1 + i * TL_P128 + kb + 4 * hh);
            const float4 b = *reinterpret_cast<const float4*>(sW4 + i * TL_P128 + kb + 4 * hh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        {
            const float bj = sB[32 + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                T2[row * TL_P32 + i] = acc[r] + bj;                   // (the c2 tile is dead: m1's MFMAs have read it)
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
            if (m0 + r < M)
                *reinterpret_cast<float4*>(x4 + (m0 + r) * 32 + c4 * 4) = *reinterpret_cast<const float4*>(T2 + r * TL_P32 + c4 * 4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward "tail" (round 2): the reverse of tail_fwd_kernel in ONE pass, for every 32-row tile
//     dm1 = (dx4 W4) * [m1 > 0]        32 -> 128      dW4 += dx4^T m1      db4 += colsum dx4     dE1[goal] += rowgroup-sum dm1
//     dc2 = (dm1 W3a) * [c2 > 0]       128 -> 32      dW3a += dm1^T c2
//     dc1 = (dc2 W2) * [c1 > 0]        32 -> 128      dW2 += dc2^T c1      db2 += colsum dc2     db1 += colsum dc1
// The GEMM path did this with three NN GEMMs, three TN GEMMs, three column sums and a row-group scatter, each streaming
// [T*N*49, 128|32] fp32 tensors through HBM (~9 ms per slice and epoch); this kernel reads dx4, m1, c2, c1 once and writes
// dc1 once (~1.4 GB, the floor).  A wave owns 32-row tiles; everything runs on the exact-fp32 MFMA (32x32x2): the data
// GEMMs read A rows as float4 from the wave's LDS image and B from transposed weight copies in LDS; the weight-gradient
// GEMMs contract over the tile's ROWS, so their operands are single floats of the same LDS images (lane = channel,
// k = row: the fp32 MFMA takes one element per lane, no transposed copy needed).  The weight gradients stay in the wave's
// accumulators for all of its tiles and leave as one partial set per wave; tail_bwd_reduce_kernel folds the sets into the
// gradient tensors.  dE1 is accumulated in an LDS table per workgroup (a 32-row tile touches at most two row groups: S >= 32).
constexpr int TB_MAX_WG = 256;
constexpr int TB_MAXG = 16;                       // goal rows of the per-wave dE1 register tables (num_goals <= 16 on the fused path)
constexpr int TB_W = 3 * 4096;                    // dW4 [32][128], dW3a [128][32], dW2 [32][128]
constexpr int TB_PART = TB_W + 64 + 64 + 256;     // + db4, db2 (two lane halves each), db1 (two halves)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tail_bwd_kernel(
    const float* __restrict__ dx4, const float* __restrict__ m1, const float* __restrict__ c2, const float* __restrict__ c1,
    const float* __restrict__ W2, const float* __restrict__ W3, int w3_ld, const float* __restrict__ W4,
    const int* __restrict__ goal, int S, int num_goals, float* __restrict__ dc1, uint16_t* __restrict__ dc1p,
    float* __restrict__ part, float* __restrict__ partE, long M) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float* sW4T = tsm;                            // [128 n][36]:  sW4T[n][k] = W4[k][n]
    float* sW3T = sW4T + 128 * TL_P32;            // [32 n][132]:  sW3T[n][k] = W3[k][n], n < 32
    float* sW2T = sW3T + 32 * TL_P128;            // [128 n][36]:  sW2T[n][k] = W2[k][n]
    float* sE = sW2T + 128 * TL_P32;              // [num_goals][128]
    float* wave_base = sE + (num_goals * 128 > 1024 ? num_goals * 128 : 1024);   // (== the host's tb_lds formula)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Tm = wave_base + wave * (32 * TL_P128 + 32 * TL_P32);   // m1 tile -> dm1 -> c1 tile -> dc1
    float* Tx = Tm + 32 * TL_P128;                                 // dx4 tile -> c2 tile -> dc2
    for (int idx = tid; idx < 4096; idx += 256) {
        const int k = idx >> 7, n = idx & 127;
        sW4T[n * TL_P32 + k] = W4[idx];
        sW2T[n * TL_P32 + k] = W2[idx];
        const int k3 = idx >> 5, n3 = idx & 31;
        sW3T[n3 * TL_P128 + k3] = W3[(long)k3 * w3_ld + n3];
    }
    __syncthreads();
    // dE1 (round 6: no float atomics): the tile's two row-group sums go through a wave-private 256-float line of the sE area
    // into PER-WAVE register tables regE[goal][2] (lane owns columns lane, lane + 64; the goal ids are wave-uniform, the
    // table update is 16 predicated adds) -- a wave folds its tiles in a fixed order, one table per wave leaves the kernel.
    float* sEw = sE + wave * 256;
    float regE[TB_MAXG][2];
#pragma unroll
    for (int g = 0; g < TB_MAXG; ++g) { regE[g][0] = 0.f; regE[g][1] = 0.f; }

    const int i = lane & 31, hh = lane >> 5;
    const long ntiles = (M + 31) / 32;
    const long tstep = (long)gridDim.x * 4;
    const long ngroups = (M + S - 1) / S;
    f32x16_t gW4[4], gW3[4], gW2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gW4[t][r] = 0.f; gW3[t][r] = 0.f; gW2[t][r] = 0.f; }
    float db4p = 0.f, db2p = 0.f, db1p[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4_t sm[16], sx[4];                          // register prefetch (native vectors; see tail_fwd_kernel)
    long tile = (long)blockIdx.x * 4 + wave;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
        sx[q] = *reinterpret_cast<const f32x4_t*>(dx4 + min(tile * 32 + r, M - 1) * 32 + c4 * 4);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
        sm[q] = *reinterpret_cast<const f32x4_t*>(m1 + min(tile * 32 + r, M - 1) * 128 + c4 * 4);
    }
    for (; tile < ntiles; tile += tstep) {
        const long m0 = tile * 32;
        const long grp_base = m0 / S;                                // wave-uniform
        const int nb = (int)((grp_base + 1) * S - m0);               // rows [0, nb) belong to group grp_base, the rest to the next
        // ---- stage dx4 (rows past M as zeros: they then contribute nothing anywhere) and m1 ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
            f32x4_t v = sx[q];
            if (m0 + r >= M) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4_t*>(Tx + r * TL_P32 + c4 * 4) = v;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<f32x4_t*>(Tm + r * TL_P128 + c4 * 4) = sm[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                // c2 and c1 of this tile: in flight under dW4 / dm1
            const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
            sx[q] = *reinterpret_cast<const f32x4_t*>(c2 + min(m0 + r, M - 1) * 32 + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            sm[q] = *reinterpret_cast<const f32x4_t*>(c1 + min(m0 + r, M - 1) * 128 + c4 * 4);
        }
        // ---- dW4[k][n] += sum_rows dx4[row][k] m1[row][n];  db4 ----
#pragma unroll 1
        for (int st = 0; st < 16; ++st) {
            const int row = 2 * st + hh;
            const float a = Tx[row * TL_P32 + i];
            db4p += a;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                gW4[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Tm[row * TL_P128 + jt * 32 + i], gW4[jt], 0, 0, 0);
        }
        // ---- dm1 = (dx4 W4) * [m1 > 0] -> Tm;  dE1 row-group sums (one 32-column tile at a time: 16 live accumulators) ----
        {
            int g0 = goal[min(grp_base, ngroups - 1)], g1 = goal[min(grp_base + 1, ngroups - 1)];
            g0 = g0 < 0 ? 0 : (g0 >= num_goals ? num_goals - 1 : g0);
            g1 = g1 < 0 ? 0 : (g1 >= num_goals ? num_goals - 1 : g1);
#pragma unroll 1
            for (int jt = 0; jt < 4; ++jt) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 32; kb += 8) {
                    const float4 a = *reinterpret_cast<const float4*>(Tx + i * TL_P32 + kb + 4 * hh);
                    const float4 b = *reinterpret_cast<const float4*>(sW4T + (jt * 32 + i) * TL_P32 + kb + 4 * hh);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                }
                float e0 = 0.f, e1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;     // C/D layout: column = lane & 31
                    float* pm = Tm + row * TL_P128 + jt * 32 + i;
                    const float v = (*pm > 0.f) ? acc[r] : 0.f;
                    *pm = v;
                    if (row < nb) e0 += v; else e1 += v;
                }
                e0 += __shfl_xor(e0, 32, 64);                        // the two lane halves hold the column's rows 4 hh .. : add them
                e1 += __shfl_xor(e1, 32, 64);
                if (hh == 0) { sEw[jt * 32 + i] = e0; sEw[128 + jt * 32 + i] = e1; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const float c0a = sEw[lane], c0b = sEw[64 + lane], c1a = sEw[128 + lane], c1b = sEw[192 + lane];
            const bool two = nb < 32;
#pragma unroll
            for (int g = 0; g < TB_MAXG; ++g) {
                regE[g][0] += (g == g0 ? c0a : 0.f) + ((two && g == g1) ? c1a : 0.f);
                regE[g][1] += (g == g0 ? c0b : 0.f) + ((two && g == g1) ? c1b : 0.f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the line is rewritten by the next tile)
            __builtin_amdgcn_wave_barrier();
        }
        // ---- stage c2 (dx4 is dead); fetch the next tile's dx4 ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4_t*>(Tx + r * TL_P32 + c4 * 4) = sx[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, r = idx >> 3, c4 = idx & 7;
            sx[q] = *reinterpret_cast<const f32x4_t*>(dx4 + min((tile + tstep) * 32 + r, M - 1) * 32 + c4 * 4);
        }
        // ---- dW3a[m][c] += sum_rows dm1[row][m] c2[row][c] ----
#pragma unroll 1
        for (int st = 0; st < 16; ++st) {
            const int row = 2 * st + hh;
            const float b = Tx[row * TL_P32 + i];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                gW3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Tm[row * TL_P128 + mt * 32 + i], b, gW3[mt], 0, 0, 0);
        }
        // ---- dc2 = (dm1 W3a) * [c2 > 0] -> Tx;  db2 ----
        {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int kb = 0; kb < 128; kb += 8) {
                const float4 a = *reinterpret_cast<const float4*>(Tm + i * TL_P128 + kb + 4 * hh);
                const float4 b = *reinterpret_cast<const float4*>(sW3T + i * TL_P128 + kb + 4 * hh);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                float* px = Tx + row * TL_P32 + i;
                const float v = (*px > 0.f) ? acc[r] : 0.f;
                *px = v;
                db2p += v;
            }
        }
        // ---- stage c1 (dm1 is dead); fetch the next tile's m1 ----
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<f32x4_t*>(Tm + r * TL_P128 + c4 * 4) = sm[q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            sm[q] = *reinterpret_cast<const f32x4_t*>(m1 + min((tile + tstep) * 32 + r, M - 1) * 128 + c4 * 4);
        }
        // ---- dW2[k][n] += sum_rows dc2[row][k] c1[row][n] ----
#pragma unroll 1
        for (int st = 0; st < 16; ++st) {
            const int row = 2 * st + hh;
            const float a = Tx[row * TL_P32 + i];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                gW2[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Tm[row * TL_P128 + jt * 32 + i], gW2[jt], 0, 0, 0);
        }
        // ---- dc1 = (dc2 W2) * [c1 > 0] -> Tm -> HBM;  db1 ----
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 32; kb += 8) {
                const float4 a = *reinterpret_cast<const float4*>(Tx + i * TL_P32 + kb + 4 * hh);
                const float4 b = *reinterpret_cast<const float4*>(sW2T + (jt * 32 + i) * TL_P32 + kb + 4 * hh);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            float d = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                float* pm = Tm + row * TL_P128 + jt * 32 + i;
                const float v = (*pm > 0.f) ? acc[r] : 0.f;
                *pm = v;
                d += v;
            }
            db1p[jt] += d;
        }
        if (dc1p) {                                       // dc1 leaves as three bf16 planes [row][3][128]: the operand of ec_dw_tn_x3
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
                const float4 v = *reinterpret_cast<const float4*>(Tm + r * TL_P128 + c4 * 4);
                const float x[4] = {v.x, v.y, v.z, v.w};
                uint2 p0, p1, p2;
                ec_split3x4(x, p0, p1, p2);
                if (m0 + r < M) {
                    uint16_t* d = dc1p + (m0 + r) * 384 + c4 * 4;
                    *reinterpret_cast<uint2*>(d) = p0;
                    *reinterpret_cast<uint2*>(d + 128) = p1;
                    *reinterpret_cast<uint2*>(d + 256) = p2;
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = lane + 64 * q, r = idx >> 5, c4 = idx & 31;
            if (m0 + r < M)
                *reinterpret_cast<float4*>(dc1 + (m0 + r) * 128 + c4 * 4) = *reinterpret_cast<const float4*>(Tm + r * TL_P128 + c4 * 4);
        }
        }
    }
    // ---- one partial set per wave; the dE1 table per workgroup ----
    float* pp = part + ((long)blockIdx.x * 4 + wave) * TB_PART;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            pp[row * 128 + t * 32 + i] = gW4[t][r];                  // dW4[k = row][n = t*32 + i]
            pp[4096 + (t * 32 + row) * 32 + i] = gW3[t][r];          // dW3a[m = t*32 + row][c = i]
            pp[8192 + row * 128 + t * 32 + i] = gW2[t][r];           // dW2[k = row][n = t*32 + i]
        }
    pp[TB_W + hh * 32 + i] = db4p;
    pp[TB_W + 64 + hh * 32 + i] = db2p;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) pp[TB_W + 128 + hh * 128 + jt * 32 + i] = db1p[jt];
    float* pe = partE + ((long)blockIdx.x * 4 + wave) * num_goals * 128;
#pragma unroll
    for (int g = 0; g < TB_MAXG; ++g)
        if (g < num_goals) { pe[g * 128 + lane] = regE[g][0]; pe[g * 128 + 64 + lane] = regE[g][1]; }
}

// folds the per-wave partial sets of tail_bwd_kernel into the gradient tensors (grads +=, dE1 +=).  Two stages, no atomics:
// y-slice blockIdx.y adds the sets p = y, y + ny, ... in increasing order into red[y][e]; tail_bwd_fold_kernel adds the ny
// slices in slice order onto the gradient.
__device__ __forceinline__ float* tail_dst(int e, float* gW4, float* gW3, int w3_ld, float* gW2, float* gb4, float* gb2, float* gb1,
                                           float* dE1) {
    if (e < 4096) return gW4 + e;
    if (e < 8192) { const int q = e - 4096; return gW3 + (long)(q >> 5) * w3_ld + (q & 31); }
    if (e < TB_W) return gW2 + (e - 8192);
    if (e < TB_W + 64) return gb4 + ((e - TB_W) & 31);
    if (e < TB_W + 128) return gb2 + ((e - TB_W - 64) & 31);
    if (e < TB_PART) return gb1 + ((e - TB_W - 128) & 127);
    return dE1 + (e - TB_PART);
}
__global__ __launch_bounds__(256) void tail_bwd_reduce_kernel(const float* __restrict__ part, int nsets,
                                                               const float* __restrict__ partE, int num_goals,
                                                               float* __restrict__ red) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int ne = TB_PART + num_goals * 128;
    if (e >= ne) return;
    float s = 0.f;
    if (e < TB_PART) {
        for (int p = blockIdx.y; p < nsets; p += gridDim.y) s += part[(long)p * TB_PART + e];
    } else {
        const int q = e - TB_PART;
        for (int p = blockIdx.y; p < nsets; p += gridDim.y) s += partE[(long)p * num_goals * 128 + q];
    }
    red[(long)blockIdx.y * ne + e] = s;
}
// red[ny][ne] -> gradient; a bias element, whose two lane halves sit in 2 slots of a set, is summed by ONE thread
__global__ __launch_bounds__(256) void tail_bwd_fold_kernel(const float* __restrict__ red, int ny, int num_goals,
                                                             float* __restrict__ gW4, float* __restrict__ gW3, int w3_ld,
                                                             float* __restrict__ gW2, float* __restrict__ gb4, float* __restrict__ gb2,
                                                             float* __restrict__ gb1, float* __restrict__ dE1) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int ne = TB_PART + num_goals * 128;
    if (e >= ne) return;
    auto slot = [&](int q) { float t = 0.f; for (int y = 0; y < ny; ++y) t += red[(long)y * ne + q]; return t; };
    if (e >= TB_W && e < TB_PART) {                        // bias slots: [db4 h0 | db4 h1 | db2 h0 | db2 h1 | db1 h0 (128) | db1 h1 (128)]
        const int q = e - TB_W;
        if (q < 32) gb4[q] += slot(e) + slot(e + 32);
        else if (q >= 64 && q < 96) gb2[q - 64] += slot(e) + slot(e + 32);
        else if (q >= 128 && q < 256) gb1[q - 128] += slot(e) + slot(e + 128);
        return;
    }
    *tail_dst(e, gW4, gW3, w3_ld, gW2, gb4, gb2, gb1, dE1) += slot(e);
}

// out[n] += sum_m Y[m*ld + n] WITHOUT atomics: row block blockIdx.y leaves its column sums in part[y][n] (colsum_part* below),
// colsum_fold_kernel adds the row blocks in order
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, int nby, int N, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int y = 0; y < nby; ++y) s += part[(long)y * N + n];
    out[n] += s;
}
// dW[m*ldc + n] += sum_k parts[k][m][n]: the ordered fold of a split-K weight-gradient GEMM's partial matrices (EC_GEMM_SPLIT_PARTS)
__global__ __launch_bounds__(256) void splitk_fold_kernel(const float* __restrict__ parts, int sk, int Mo, int No, int ldc,
                                                         float* __restrict__ dW) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)Mo * No) return;
    float s = 0.f;
    for (int k = 0; k < sk; ++k) s += parts[(long)k * Mo * No + q];
    const long m = q / No, n = q - m * No;
    dW[m * ldc + n] += s;
}

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

enum { P_EMB, P_W1, P_B1, P_W2, P_B2, P_W3, P_B3, P_W4, P_B4, P_WIH, P_WHH, P_BIH, P_BHH, P_WA, P_BA, P_WC, P_BC,
       // dual (RGB + depth) encoder: the depth stream's compressor / combiner, same shapes as P_W1 .. P_B4
       P_W1D, P_B1D, P_W2D, P_B2D, P_W3D, P_B3D, P_W4D, P_B4D, P_COUNT };
constexpr int P_COUNT_SINGLE = P_BC + 1;

}  // namespace

struct ec_policy {
    ec_policy_cfg c;
    size_t off[P_COUNT], num[P_COUNT], total;
    const float* goal_table = nullptr;   // fusion == 1: borrowed f32 [num_goals, in_channels]
    // EC_POLICY_INFER_REUSE bookkeeping: which (T, N, feature dtype) the weight-derived tables in a given act workspace were
    // built for.  The tables' offsets and WHICH of them exist (E1, W1's planes in fragment order, the re-ordered weight_ih and
    // its fragment-order copy) follow from exactly these three values, so a REUSE call with another geometry or dtype than the
    // call that built them (fp32 features first, bf16 next) rebuilds instead of reading uninitialised tables.
    struct Built { int T, N, bf16; };
    mutable std::mutex tables_mu;
    mutable std::unordered_map<const void*, Built> tables;
};

namespace {

struct Ws {   // float offsets into the workspace
    size_t E1, c1, c2, m1, x4, x, gi, gh, gates, hn, hp, hs, goal32, w1p;
    size_t E1d, c1d, c2d, m1d, x4d;   // the depth stream's copies (dual encoder)
    size_t wihA;                      // act step: weight_ih in pixel-major column order (valid while E1 is)
    size_t wihF, w1pF;                // ... and the same / W1's planes in MFMA-FRAGMENT order for gi_act_kernel / c1_act_kernel
    size_t dhs, dhc, dgi, dghb, dx, dx4, dm1, dc2, dc1, dE1, tpart, tpartE, whhT, wihP, gwihP, tA, tB, end;
};

Ws layout(const ec_policy* h, int T, int N, bool bwd) {
    const ec_policy_cfg& c = h->c;
    const size_t B = (size_t)T * N, S = (size_t)c.spatial * c.spatial, M49 = c.fusion ? 0 : B * S, H = c.hidden;
    const size_t nstream = (c.dual && !c.fusion) ? 2 : 1;
    const size_t flat = c.fusion ? (size_t)c.in_channels : nstream * c.comb_out * S;
    Ws w; size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += al(n * 4) / 4; return r; };
    w.E1 = take((size_t)c.num_goals * c.comb_hid);
    // small launches (the act step): room for the split-K partial matrices of c1 and gi (see ec_policy_forward)
    const size_t parts = (M49 > 0 && M49 <= (size_t)ACT_MAX_ROWS) ? ACT_PARTS : 1;
    w.c1 = take(M49 * c.compress_hid * parts);
    w.c2 = take(M49 * c.compress_out);
    w.m1 = take(M49 * c.comb_hid);
    w.x4 = take(M49 * c.comb_out);
    w.E1d = w.c1d = w.c2d = w.m1d = w.x4d = o;
    if (nstream == 2) {
        w.E1d = take((size_t)c.num_goals * c.comb_hid);
        w.c1d = take(M49 * c.compress_hid * parts);
        w.c2d = take(M49 * c.compress_out);
        w.m1d = take(M49 * c.comb_hid);
        w.x4d = take(M49 * c.comb_out);
    }
    w.x = take(B * flat);
    w.gi = take(B * 3 * H * parts);
    w.gh = take((size_t)N * 3 * H);
    w.gates = take(B * 3 * H);
    w.hn = take(B * H);
    w.hp = take(B * H);
    w.hs = take(B * H);
    w.goal32 = take(B);
    w.w1p = take(c.fusion ? 0 : ((size_t)c.compress_hid * 3 * c.in_channels + 1) / 2);   // W1 as three bf16 planes
    w.wihA = take((c.fusion || c.dual) ? 0 : 3 * H * flat);
    w.wihF = take((c.fusion || c.dual) ? 0 : 3 * H * flat);
    w.w1pF = take(c.fusion ? 0 : ((size_t)c.compress_hid * 3 * c.in_channels + 1) / 2);
    w.dhs = w.dhc = w.dgi = w.dghb = w.dx = w.dx4 = w.dm1 = w.dc2 = w.dc1 = w.dE1 = w.tpart = w.tpartE = w.whhT = w.wihP = w.gwihP = w.tA = w.tB = o;
    if (bwd) {
        w.dhs = take(B * H);
        w.dhc = take((size_t)N * H);
        w.dgi = take(B * 3 * H);
        w.dghb = take(B * 3 * H);
        w.dx = take(B * flat);
        w.dx4 = take(M49 * c.comb_out);
        w.dm1 = take(M49 * c.comb_hid);
        w.dc2 = take(M49 * c.compress_out);
        w.dc1 = take(M49 * c.compress_hid * 3 / 2 + 4);   // fp32 [M49][hid], or its three bf16 planes [M49][3][hid]
        w.dE1 = take((size_t)c.num_goals * c.comb_hid);
        w.tpart = take(c.fusion ? 0 : (size_t)TB_MAX_WG * 4 * TB_PART);              // tail_bwd_kernel's partial sets
        w.tpartE = take(c.fusion ? 0 : (size_t)TB_MAX_WG * 4 * c.num_goals * 128);   // ... and its per-wave dE1 tables
        w.whhT = take(H * 3 * H);                                                       // W_hh^T (fused backward step)
        w.wihP = take(c.fusion ? 0 : 3 * H * flat);                                     // weight_ih in pixel-major column order (EC_WIH_PERM)
        w.gwihP = take(c.fusion ? 0 : 3 * H * flat);                                    // ... and its gradient
        w.tA = take(B * 3 * H);                                                         // transposed operands of the GRU's weight-gradient GEMMs
        w.tB = take(B * (flat > H ? flat : H));
    }
    w.end = o;
    return w;
}

int pick_splitk(long M, long N, long K) {
    const long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    long s = 1024 / (tiles > 0 ? tiles : 1);
    const long nk = (K + 31) / 32;
    if (s > nk / 4) s = nk / 4;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// LinearActorHead + LinearCriticHead ([U] allenact basic_models.py): hv[b, :A] = hs[b] . Wa^T + ba, hv[b, A] = hs[b] . wc + bc.
// A+1 <= 8 outputs of a 512-long dot product per row: one wave per row (the tiled GEMM would run this as ONE
// workgroup -- 44 us for 128 rows).  Lane l owns elements l, l+64, ... of the row; fixed-order wave reduction.
// smp.actions != nullptr (ec_policy_act): the wave that owns a row also samples its action from the logits it has just reduced
// (CategoricalDistr.sample + log_prob, ec_sample_row: the arithmetic of sample_kernel) -- the act step's sixth launch folded into its fifth.
struct SampleArgs {
    long long* actions = nullptr;
    float* logp = nullptr;
    float* values = nullptr;
    uint64_t seed = 0, step = 0;
    int first_actor = 0;
};
template <int MAXO>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ hs, const float* __restrict__ Wa,
                                                       const float* __restrict__ ba, const float* __restrict__ Wc,
                                                       const float* __restrict__ bc, float* __restrict__ hv, long B,
                                                       int H, int A, SampleArgs smp) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.f;
    const float* x = hs + row * H;
    if ((H & 255) == 0) {
        // lane l owns the 4 elements [4 l, 4 l + 4) of every 256-wide chunk: one float4 of the row and one of each output's
        // weight row per chunk, all MAXO + 1 loads of a chunk in flight together (the scalar-per-lane loop waited for
        // 8 dependent rounds of 8 loads: 17 us for a 9-us GRU step in front of it); outputs past A read Wc (discarded)
        for (int k0 = 0; k0 < H; k0 += 256) {
            const float4 v = *reinterpret_cast<const float4*>(x + k0 + 4 * lane);
            float4 wv[MAXO];
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                wv[o] = *reinterpret_cast<const float4*>((o < A ? Wa + (long)o * H : Wc) + k0 + 4 * lane);
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                acc[o] = fmaf(v.w, wv[o].w, fmaf(v.z, wv[o].z, fmaf(v.y, wv[o].y, fmaf(v.x, wv[o].x, acc[o]))));
        }
    } else {
        for (int k = lane; k < H; k += 64) {
            const float v = x[k];
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o <= A) acc[o] = fmaf(v, (o < A ? Wa[(long)o * H + k] : Wc[k]), acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off, 64);
    }
    if (lane <= A) {
        float r = 0.f;
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
            if (o == lane) r = acc[o];
        hv[row * (A + 1) + lane] = r + (lane < A ? ba[lane] : bc[0]);
    }
    if (smp.actions) {   // (wave-uniform) every lane holds every reduced output: lane 0 samples from the final logits
        float lg[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) lg[o] = acc[o] + (o < A ? ba[o] : (o == A ? bc[0] : 0.f));
        if (lane == 0) {
            int a;
            float lp;
            ec_sample_row([&](int k) { float v = 0.f;
#pragma unroll
                              for (int o = 0; o < MAXO; ++o) if (o == k) v = lg[o];
                              return v; },
                          A, smp.seed, smp.step, (uint64_t)(row + smp.first_actor), a, lp);
            smp.actions[row] = a;
            smp.logp[row] = lp;
            float vv = 0.f;
#pragma unroll
            for (int o = 0; o < MAXO; ++o) if (o == A) vv = lg[o];
            if (smp.values) smp.values[row] = vv;
        }
    }
}

// split-K for the small per-step GEMMs that only ever ACCUMULATE (atomics are fine): aim at ~1 workgroup
// per CU on 64x64 tiles while keeping >= 4 K-tiles per slice.
int step_splitk(long M, long N, long K) {
    const long blocks = ((M + 63) / 64) * ((N + 63) / 64);
    long s = 256 / (blocks > 0 ? blocks : 1);
    const long nk = (K + 31) / 32;
    if (s > nk / 4) s = nk / 4;
    return (int)(s < 1 ? 1 : s);
}

#define RC(x) do { int rc__ = (x); if (rc__ != EC_OK) return rc__; } while (0)

}  // namespace

extern "C" int ec_policy_create(ec_policy_t** out, const ec_policy_cfg* cfg) {
    if (!out || !cfg) return EC_ERR_ARG;
    const ec_policy_cfg& c = *cfg;
    if (c.fusion != 0 && c.fusion != 1) return EC_ERR_ARG;
    if (c.dual != 0 && c.dual != 1) return EC_ERR_ARG;
    if (c.dual && c.fusion) return EC_ERR_ARG;
    if (c.in_channels <= 0 || c.spatial <= 0 || c.hidden <= 0 || c.num_goals <= 0 || c.num_actions <= 0) return EC_ERR_SHAPE;
    if ((c.in_channels & 3) || (c.hidden & 3)) return EC_ERR_SHAPE;
    if (c.fusion) {
        if (c.spatial != 1) return EC_ERR_SHAPE;
    } else {
        if (c.goal_dims <= 0 || c.compress_hid <= 0 || c.compress_out <= 0 || c.comb_hid <= 0 || c.comb_out <= 0)
            return EC_ERR_SHAPE;
        if ((c.compress_hid & 3) || (c.compress_out & 3) || (c.comb_hid & 3) || (c.comb_out & 3) || (c.goal_dims & 3))
            return EC_ERR_SHAPE;
    }
    ec_policy* h = new (std::nothrow) ec_policy();
    if (!h) return EC_ERR_ALLOC;
    h->c = c;
    const size_t S = (size_t)c.spatial * c.spatial, H = c.hidden;
    const size_t flat = c.fusion ? (size_t)c.in_channels : (size_t)(c.dual ? 2 : 1) * c.comb_out * S;
    size_t n[P_COUNT] = {(size_t)c.num_goals * c.goal_dims,
                               (size_t)c.compress_hid * c.in_channels, (size_t)c.compress_hid,
                               (size_t)c.compress_out * c.compress_hid, (size_t)c.compress_out,
                               (size_t)c.comb_hid * (c.compress_out + c.goal_dims), (size_t)c.comb_hid,
                               (size_t)c.comb_out * c.comb_hid, (size_t)c.comb_out,
                               3 * H * flat, 3 * H * H, 3 * H, 3 * H,
                               (size_t)c.num_actions * H, (size_t)c.num_actions, H, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c.fusion)
        for (int i = P_EMB; i <= P_B4; ++i) n[i] = 0;      // no goal embedding / compressor / combiner: GRU + heads only
    if (c.dual)
        for (int i = P_W1; i <= P_B4; ++i) n[i + (P_W1D - P_W1)] = n[i];
    size_t o = 0;
    for (int i = 0; i < P_COUNT; ++i) { h->off[i] = o; h->num[i] = n[i]; o += (n[i] + 3) / 4 * 4; }   // 16-B aligned
    h->total = o;
    *out = h;
    return EC_OK;
}
extern "C" void ec_policy_destroy(ec_policy_t* h) { delete h; }
extern "C" int ec_policy_set_goal_table(ec_policy_t* h, const float* table) {
    if (!h || !table || !h->c.fusion) return EC_ERR_ARG;
    h->goal_table = table;
    return EC_OK;
}
extern "C" int ec_policy_num_param_tensors(const ec_policy_t* h) { return (h && h->c.dual) ? P_COUNT : P_COUNT_SINGLE; }
extern "C" size_t ec_policy_flat_size(const ec_policy_t* h) { return h ? h->total : 0; }
extern "C" int ec_policy_param_offset(const ec_policy_t* h, int idx, size_t* off, size_t* numel) {
    if (!h || idx < 0 || idx >= P_COUNT || !off || !numel) return EC_ERR_ARG;
    *off = h->off[idx]; *numel = h->num[idx];
    return EC_OK;
}
extern "C" size_t ec_policy_workspace_bytes(const ec_policy_t* h, int T, int N, int for_backward) {
    if (!h || T <= 0 || N <= 0) return 0;
    return layout(h, T, N, for_backward != 0).end * 4;
}

extern "C" int ec_policy_forward(const ec_policy_t* h, const float* params, const void* feat, int feat_bf16,
                                 const int64_t* goal, const float* h0, const float* masks, int T, int N,
                                 void* workspace, size_t ws_bytes, int for_backward, float* hv, float* h_final,
                                 ec_stream_t stream) {
    return ec_policy_forward2(h, params, feat, nullptr, feat_bf16, goal, h0, masks, T, N, workspace, ws_bytes, for_backward, hv,
                              h_final, stream);
}

namespace {
int policy_forward_impl(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                        const int64_t* goal, const float* h0, const float* masks, int T, int N,
                        void* workspace, size_t ws_bytes, int for_backward, float* hv, float* h_final, const SampleArgs& smp,
                        ec_stream_t stream);
}
extern "C" int ec_policy_forward2(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                                  const int64_t* goal, const float* h0, const float* masks, int T, int N,
                                  void* workspace, size_t ws_bytes, int for_backward, float* hv, float* h_final,
                                  ec_stream_t stream) {
    return policy_forward_impl(h, params, feat, feat2, feat_bf16, goal, h0, masks, T, N, workspace, ws_bytes, for_backward, hv, h_final,
                               SampleArgs{}, stream);
}

// The act step in ONE call ([U] allenact OnPolicyRLEngine.act: actor_critic(...) then distributions.sample() / log_probs()):
// ec_policy_forward2(T = 1, EC_POLICY_INFER | EC_POLICY_INFER_REUSE) whose heads launch also samples -- exactly the results of
// ec_policy_forward2 followed by ec_sample_actions, one launch less on the act step's serial chain.
extern "C" int ec_policy_act(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                             const int64_t* goal, const float* h0, const float* masks, int N, void* workspace, size_t ws_bytes,
                             int reuse_tables, float* hv, float* h_final, int64_t* actions, float* logp, float* values,
                             uint64_t seed, uint64_t step, int first_actor, ec_stream_t stream) {
    if (!actions || !logp) return EC_ERR_ARG;
    if (!h || h->c.num_actions + 1 > 8) return EC_ERR_UNSUPPORTED;       // (the one-wave-per-row heads launch)
    SampleArgs smp;
    smp.actions = (long long*)actions; smp.logp = logp; smp.values = values; smp.seed = seed; smp.step = step; smp.first_actor = first_actor;
    return policy_forward_impl(h, params, feat, feat2, feat_bf16, goal, h0, masks, 1, N, workspace, ws_bytes,
                               reuse_tables ? EC_POLICY_INFER_REUSE : EC_POLICY_INFER, hv, h_final, smp, stream);
}

namespace {
int policy_forward_impl(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                        const int64_t* goal, const float* h0, const float* masks, int T, int N,
                        void* workspace, size_t ws_bytes, int for_backward, float* hv, float* h_final, const SampleArgs& smp,
                        ec_stream_t stream) {
    if (!h || !params || !feat || !goal || !h0 || !masks || !workspace || !hv) return EC_ERR_ARG;
    if (h->c.dual && !feat2) return EC_ERR_ARG;
    if (T <= 0 || N <= 0) return EC_ERR_SHAPE;
    const ec_policy_cfg& c = h->c;
    const Ws w = layout(h, T, N, false);
    if (for_backward < 0 || for_backward > EC_POLICY_INFER_REUSE) return EC_ERR_ARG;
    if (ws_bytes < layout(h, T, N, for_backward == EC_POLICY_LEARN).end * 4) return EC_ERR_WORKSPACE;
    float* ws = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const int B = T * N, S = c.spatial * c.spatial, H = c.hidden, A1 = c.num_actions + 1;
    const int M49 = B * S, C = c.in_channels, cat = c.compress_out + c.goal_dims;
    const int nstream = (c.dual && !c.fusion) ? 2 : 1, flat1 = c.comb_out * S;
    const int flat = c.fusion ? c.in_channels : nstream * flat1;
    const float* P = params;
    auto W = [&](int i) { return P + h->off[i]; };
    int* goal32 = (int*)(ws + w.goal32);
    // Act-step path (for_backward == 0, stated by the caller -- never inferred from the workspace size): inference only,
    // so c1 need not be materialised and the two long-K, small-M GEMMs (compressor conv 1, GRU input projection) run as
    // ACT_PARTS K slices whose partial matrices the consuming kernels fold in a fixed order.
    const bool infer_only = for_backward == 0 || for_backward == EC_POLICY_INFER_REUSE;
    // EC_POLICY_INFER_REUSE: the weight-derived table E1 that an EC_POLICY_INFER call left in THIS workspace is still valid
    // (same parameters: every act step of a rollout after the first) -- one GEMM launch less on the act step's chain
    bool reuse_tables = for_backward == EC_POLICY_INFER_REUSE;
    {
        std::lock_guard<std::mutex> lk(h->tables_mu);
        if (!infer_only) {
            h->tables.erase(workspace);                          // a learn pass overwrites the workspace
        } else {
            auto it = h->tables.find(workspace);
            if (reuse_tables && (it == h->tables.end() || it->second.T != T || it->second.N != N || it->second.bf16 != (feat_bf16 ? 1 : 0)))
                reuse_tables = false;                            // nothing valid for THIS geometry / dtype in this workspace: build
            if (!reuse_tables) {
                // bounded: a caller that allocates a fresh act workspace per call would otherwise add an entry per allocator
                // address for the life of the handle.  Dropping every record is always safe -- a REUSE call that
                // finds none rebuilds its tables.
                if (h->tables.size() >= 64 && h->tables.find(workspace) == h->tables.end()) h->tables.clear();
                h->tables[workspace] = ec_policy::Built{T, N, feat_bf16 ? 1 : 0};
            }
        }
    }
    // learn pass: the GRU's input projection reads the combiner output where it lies (pixel-major rows) against a re-ordered
    // weight_ih (permute_row_kernel) -- no activation transposes in either direction
    const bool wih_perm = ec_config().wih_perm && !infer_only && !c.fusion && !c.dual && (size_t)c.comb_out * S * 4 <= 64 * 1024;
    const Ws wb = layout(h, T, N, !infer_only);
    const bool small = !c.fusion && M49 > 0 && M49 <= ACT_MAX_ROWS;            // (== the condition in layout())
    const size_t tail_lds_ = ((size_t)2 * 32 * TL_P128 + 128 * TL_P32 + (size_t)c.num_goals * 128 + 64 +
                              4 * (size_t)(32 * TL_P128 + 32 * TL_P32)) * sizeof(float);
    const int act_parts_on = ec_config().act_split;
    const int tail_fused_ = ec_config().tail_fused;
    const int gru_fused_ = ec_config().gru_fused;
    const bool tail_ok = tail_fused_ && c.compress_hid == 128 && c.compress_out == 32 && c.comb_hid == 128 &&
                         c.comb_out == 32 && tail_lds_ <= 160 * 1024;
    const size_t gru_lds_ = std::max((size_t)2 * 32 * (((H & 255) == 0 ? 256 : H) + 4) * sizeof(float), (size_t)4 * 32 * 32 * sizeof(float));
    const bool step_ok = gru_fused_ && (H % 32) == 0 && gru_lds_ <= 160 * 1024;
    const bool act_split = act_parts_on && infer_only && small && tail_ok && (C % 32) == 0 && C / 32 >= ACT_PARTS;
    const bool gi_split = act_parts_on && infer_only && small && step_ok && (flat + 31) / 32 >= ACT_PARTS;
    // act step: the fused tail reads the low words of the int64 goal ids itself (== the conversion kernel's truncation): one
    // launch less on the act step's chain; and its GRU input projection runs against the re-ordered weight_ih (built by the
    // first act step after a parameter update, kept in the workspace like E1), so the channel-major transpose goes too
    const bool goal_direct = infer_only && !c.fusion && tail_ok;
    const bool wih_act = infer_only && !c.fusion && !c.dual && tail_ok && ec_config().wih_perm && (size_t)c.comb_out * S * 4 <= 64 * 1024;
    if (!goal_direct)
    hipLaunchKernelGGL(goal_to_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const long long*)goal, goal32, B);
    if (c.fusion) {
        if (!h->goal_table) return EC_ERR_ARG;
        if (feat_bf16)
            hipLaunchKernelGGL(fuse_goal_kernel<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, feat, h->goal_table,
                               goal32, ws + w.x, (long)B, flat, c.num_goals);
        else
            hipLaunchKernelGGL(fuse_goal_kernel<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, feat, h->goal_table,
                               goal32, ws + w.x, (long)B, flat, c.num_goals);
    } else {
    for (int sidx = 0; sidx < nstream; ++sidx) {   // dual encoder: the RGB stream, then the depth stream (own weights, own activations)
    const void* featS = sidx ? feat2 : feat;
    auto WS = [&](int i) { return P + h->off[(sidx && i >= P_W1 && i <= P_B4) ? i + (P_W1D - P_W1) : i]; };
    const size_t o_E1 = sidx ? w.E1d : w.E1, o_c1 = sidx ? w.c1d : w.c1, o_c2 = sidx ? w.c2d : w.c2, o_m1 = sidx ? w.m1d : w.m1,
                 o_x4 = sidx ? w.x4d : w.x4;
    // E1 = embed_class @ W3[:, co:]^T + b3
    if (!reuse_tables)
    RC(ec_gemm_f32(WS(P_EMB), WS(P_W3) + c.compress_out, ws + o_E1, c.num_goals, c.comb_hid, c.goal_dims, c.goal_dims, 1,
                   1, cat, c.comb_hid, 0, WS(P_B3), nullptr, nullptr, 0, nullptr, nullptr, 1, stream));
    // resnet_compressor
    // EC_C1_PINGPONG (default 1): bf16 features x fp32 W1 as three bf16 planes on the 8-wave ping-pong kernel
    // (conv_igemm8, X3 mode) once there are enough 256-row tiles to fill the chip; else the generic x3 GEMM
    const int c1_pp = ec_config().c1_pingpong;
    bool c1_final = false;                                       // c1 already carries bias + ReLU (no partial matrices to fold)
    if (c1_pp && feat_bf16 && c.compress_hid % 128 == 0 && C % 64 == 0 && M49 >= 256 * 128) {
        RC(ec_split3_bf16(WS(P_W1), ws + w.w1p, c.compress_hid, C, stream));
        // (EC_POLICY_FAST: the two leading planes of W1 -- 16 mantissa bits -- instead of all three: a third fewer MFMAs in a
        //  launch that is MFMA-bound at three products per feature byte; learn pass only, the act step stays exact)
        RC(ec_gemm_bf16a_xp(featS, ws + w.w1p, WS(P_B1), ws + o_c1, M49, c.compress_hid, C, 1 /* EC_ACT_RELU */,
                            (ec_config().policy_fast && !infer_only) ? 2 : 3, stream));
    } else if (act_split && feat_bf16 && !c.dual && (C % 128) == 0 && (c.compress_hid % 32) == 0) {
        // act step: one launch, K split over the waves of a workgroup, W1's three bf16 planes cached in the workspace with E1
        if (!reuse_tables) {
            RC(ec_split3_bf16(WS(P_W1), ws + w.w1p, c.compress_hid, C, stream));
            const long nu = (long)c.compress_hid * 3 * C / 8;
            hipLaunchKernelGGL(frag_pack_planes_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, (const uint4*)(ws + w.w1p),
                               (uint4*)(ws + w.w1pF), c.compress_hid, C);
        }
        {
            const uint16_t* fa_ = (const uint16_t*)featS;
            const uint16_t* wp_ = (const uint16_t*)(ws + w.w1pF);
            const unsigned nx = (unsigned)(c.compress_hid / 32);
            if (M49 <= 2048)
                hipLaunchKernelGGL((c1_act_kernel<1, 8>), dim3(nx, (unsigned)((M49 + 31) / 32)), dim3(512), 0, s, fa_, wp_, WS(P_B1),
                                   ws + o_c1, (long)M49, C, c.compress_hid);
            else if (M49 <= 4096)
                hipLaunchKernelGGL((c1_act_kernel<2, 4>), dim3(nx, (unsigned)((M49 + 63) / 64)), dim3(512), 0, s, fa_, wp_, WS(P_B1),
                                   ws + o_c1, (long)M49, C, c.compress_hid);
            else
                hipLaunchKernelGGL((c1_act_kernel<4, 4>), dim3(nx, (unsigned)((M49 + 127) / 128)), dim3(512), 0, s, fa_, wp_, WS(P_B1),
                                   ws + o_c1, (long)M49, C, c.compress_hid);
        }
        c1_final = true;
    } else if (act_split) {
        // act step: K = C is long and M small (196 workgroups walking 64 K-steps each): four K slices write four
        // partial matrices, tail_fwd_kernel folds them (+ b1, ReLU) in a fixed order -- no atomics, bit-reproducible,
        // and independent of how the actors are sliced
        RC(ec_gemm_f32(featS, WS(P_W1), ws + o_c1, M49, c.compress_hid, C, C, 1, 1, C, c.compress_hid,
                       EC_GEMM_SPLIT_PARTS | (feat_bf16 ? EC_GEMM_A_BF16 : 0), nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                       ACT_PARTS, stream));
    } else
    RC(ec_gemm_f32(featS, WS(P_W1), ws + o_c1, M49, c.compress_hid, C, C, 1, 1, C, c.compress_hid,
                   EC_GEMM_RELU | (feat_bf16 ? EC_GEMM_A_BF16 : 0), WS(P_B1), nullptr, nullptr, 0, nullptr, nullptr, 1,
                   stream));
    // EC_TAIL_FUSED (default 1): c2 / m1 / x4 in one pass over c1 for the reference's widths
    const int tail_fused = ec_config().tail_fused;
    const size_t tail_lds = ((size_t)2 * 32 * TL_P128 + 128 * TL_P32 + (size_t)c.num_goals * 128 + 64 +
                             4 * (size_t)(32 * TL_P128 + 32 * TL_P32)) * sizeof(float);
    if (tail_fused && c.compress_hid == 128 && c.compress_out == 32 && c.comb_hid == 128 && c.comb_out == 32 &&
        tail_lds <= 160 * 1024) {
        static std::atomic<uint64_t> attr_done{0};
        if (auto attr_g_ = ec_attr_needed(attr_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tail_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        const long ntiles = ((long)M49 + 31) / 32;
        long nwg = (ntiles + 3) / 4;
        if (nwg > 512) nwg = 512;
        hipLaunchKernelGGL(tail_fwd_kernel, dim3((unsigned)nwg), dim3(256), tail_lds, s, ws + o_c1, WS(P_W2), WS(P_B2), WS(P_W3), cat,
                           ws + o_E1, goal_direct ? (const int*)goal : goal32, S, c.num_goals, WS(P_W4), WS(P_B4), ws + o_c2, ws + o_m1, ws + o_x4, (long)M49,
                           (act_split && !c1_final) ? ACT_PARTS : 1, (long)M49 * c.compress_hid, WS(P_B1), goal_direct ? 2 : 1);
    } else {
    RC(ec_gemm_f32(ws + o_c1, WS(P_W2), ws + o_c2, M49, c.compress_out, c.compress_hid, c.compress_hid, 1, 1,
                   c.compress_hid, c.compress_out, EC_GEMM_RELU, WS(P_B2), nullptr, nullptr, 0, nullptr, nullptr, 1,
                   stream));
    // target_obs_combiner (goal half folded into the row-group bias E1[goal])
    RC(ec_gemm_f32(ws + o_c2, WS(P_W3), ws + o_m1, M49, c.comb_hid, c.compress_out, c.compress_out, 1, 1, cat,
                   c.comb_hid, EC_GEMM_RELU, nullptr, ws + o_E1, goal32, S, nullptr, nullptr, 1, stream));
    RC(ec_gemm_f32(ws + o_m1, WS(P_W4), ws + o_x4, M49, c.comb_out, c.comb_hid, c.comb_hid, 1, 1, c.comb_hid,
                   c.comb_out, 0, WS(P_B4), nullptr, nullptr, 0, nullptr, nullptr, 1, stream));
    }
    if (wih_perm) {
        hipLaunchKernelGGL(permute_row_kernel, dim3((unsigned)(3 * H)), dim3(256), flat * sizeof(float), s, W(P_WIH),
                           ws + wb.wihP, S, c.comb_out, 0);
    } else if (wih_act) {
        if (!reuse_tables)
            hipLaunchKernelGGL(permute_row_kernel, dim3((unsigned)(3 * H)), dim3(256), flat * sizeof(float), s, W(P_WIH),
                               ws + w.wihA, S, c.comb_out, 0);
    } else {
        const long total = (long)B * flat1;
        hipLaunchKernelGGL(to_cmajor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws + o_x4, ws + w.x,
                           S, c.comb_out, total, flat, sidx * flat1);
    }
    }   // streams
    }   // !fusion
    // GRU: input projection for all T at once, then the sequential recurrence
    // (split-K only as separate partial matrices folded by the step kernel: the act step stays free of float atomics, so
    //  rollouts are bit-reproducible)
    // act step with a handful of rows: one launch, K split over the waves of a workgroup, folded in-kernel (gi_act_kernel)
    const int gi_nw = (flat % 56 == 0) ? 7 : ((flat % 64 == 0) ? 8 : 0);
    const bool gi_small = infer_only && wih_act && T == 1 && B <= 256 && gi_nw != 0 && (3 * H) % 32 == 0 && ec_config().act_split;
    if (gi_small) {
        const float* xin = ws + w.x4;
        const float* win = ws + w.wihF;                              // the re-ordered weight_ih in fragment order
        if (!reuse_tables) {
            const long nu = (long)3 * H * flat / 4;
            hipLaunchKernelGGL(frag_pack_f32_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, (const float4*)(ws + w.wihA),
                               (float4*)(ws + w.wihF), 3 * H, flat);
        }
        const dim3 grid((unsigned)(3 * H / 32), (unsigned)((B + 31) / 32));
        if (gi_nw == 7) hipLaunchKernelGGL(gi_act_kernel<7>, grid, dim3(448), 0, s, xin, win, W(P_BIH), ws + w.gi, B, flat, 3 * H);
        else hipLaunchKernelGGL(gi_act_kernel<8>, grid, dim3(512), 0, s, xin, win, W(P_BIH), ws + w.gi, B, flat, 3 * H);
    } else
    RC(ec_gemm_f32((wih_perm || wih_act) ? ws + w.x4 : ws + w.x, wih_perm ? ws + wb.wihP : (wih_act ? ws + w.wihA : W(P_WIH)), ws + w.gi, B, 3 * H, flat, flat, 1, 1, flat, 3 * H,
                   gi_split ? EC_GEMM_SPLIT_PARTS : 0, W(P_BIH), nullptr, nullptr, 0, nullptr, nullptr, gi_split ? ACT_PARTS : 1,
                   stream));
    // EC_GRU_FUSED (default 1): one fused launch per step where the geometry allows (H % 32 == 0, tiles fit the LDS)
    const int gru_fused = ec_config().gru_fused;
    size_t gru_lds = (size_t)2 * 32 * (((H & 255) == 0 ? 256 : H) + 4) * sizeof(float);      // operand tiles (K chunk) ...
    if (gru_lds < 4 * 32 * 32 * sizeof(float)) gru_lds = 4 * 32 * 32 * sizeof(float);   // ... reused for the 4 partial tiles
    const bool fused_step = gru_fused && (H % 32) == 0 && gru_lds <= 160 * 1024;
    if (fused_step) {
        static std::atomic<uint64_t> attr_done{0};
        if (auto attr_g_ = ec_attr_needed(attr_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fwd_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // inference with T == 1 (the act step): the new state goes straight to h_final (no copy launch afterwards) and the
    // heads read it there
    float* hs_base = (infer_only && T == 1 && h_final && h_final != h0) ? h_final : ws + w.hs;
    // the update's recurrences (learn pass, H == 512): 16 x 16-tile kernel, one workgroup per CU with 140 KB of LDS; the act
    // step keeps the 66-KB kernel, which co-resides with the encoder launches of the other actor slice
    const bool tile16 = fused_step && gru_fused >= 2 && H == 512 && !infer_only;
    const size_t gru16_lds = (size_t)(4 * 64 * 128 + 4 * 3 * 256) * sizeof(float);
    if (tile16) {
        static std::atomic<uint64_t> attr16_done{0};
        if (auto attr_g_ = ec_attr_needed(attr16_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fwd512_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    for (int t = 0; t < T; ++t) {
        const float* hprev = (t == 0) ? h0 : hs_base + (size_t)(t - 1) * N * H;
        const size_t o3 = (size_t)t * N * 3 * H, o1 = (size_t)t * N * H;
        if (tile16) {
            hipLaunchKernelGGL(gru_step_fwd512_kernel, dim3(32u, (unsigned)((N + 15) / 16)), dim3(256), gru16_lds, s,
                               ws + w.gi + o3, W(P_WHH), W(P_BHH), hprev, masks + (size_t)t * N, hs_base + o1,
                               ws + w.gates + o3, ws + w.hn + o1, ws + w.hp + o1, N);
            continue;
        }
        if (fused_step) {
            hipLaunchKernelGGL(gru_step_fwd_kernel, dim3((unsigned)(H / 8), (unsigned)((N + 31) / 32)), dim3(256), gru_lds, s,
                               ws + w.gi + o3, W(P_WHH), W(P_BHH), hprev, masks + (size_t)t * N, hs_base + o1,
                               ws + w.gates + o3, ws + w.hn + o1, ws + w.hp + o1, N, H, (gi_split && !gi_small) ? ACT_PARTS : 1,
                               (long)B * 3 * H);
            continue;
        }
        RC(ec_gemm_f32(hprev, W(P_WHH), ws + w.gh, N, 3 * H, H, H, 1, 1, H, 3 * H, 0, nullptr, nullptr, nullptr, 0,
                       nullptr, nullptr, 1, stream));
        hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3((N * H + 255) / 256), dim3(256), 0, s, ws + w.gi + o3, ws + w.gh,
                           W(P_BHH), hprev, masks + (size_t)t * N, hs_base + o1, ws + w.gates + o3, ws + w.hn + o1,
                           ws + w.hp + o1, N, H);
    }
    // heads: hv[:, :A] = actor logits, hv[:, A] = critic value
    if (A1 <= 8) {
        hipLaunchKernelGGL(heads_fwd_kernel<8>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, hs_base, W(P_WA), W(P_BA),
                           W(P_WC), W(P_BC), hv, (long)B, H, c.num_actions, smp);
    } else {
        RC(ec_gemm_f32(hs_base, W(P_WA), hv, B, c.num_actions, H, H, 1, 1, H, A1, 0, W(P_BA), nullptr, nullptr, 0,
                       nullptr, nullptr, 1, stream));
        RC(ec_gemm_f32(hs_base, W(P_WC), hv + c.num_actions, B, 1, H, H, 1, 1, H, A1, 0, W(P_BC), nullptr, nullptr, 0,
                       nullptr, nullptr, 1, stream));
    }
    if (h_final && hs_base != h_final)
        (void)hipMemcpyAsync(h_final, ws + w.hs + (size_t)(T - 1) * N * H, (size_t)N * H * 4, hipMemcpyDeviceToDevice, s);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
}  // namespace

extern "C" int ec_policy_backward(const ec_policy_t* h, const float* params, const void* feat, int feat_bf16,
                                  const float* masks, int T, int N, void* workspace, size_t ws_bytes, const float* dhv,
                                  const float* dh_final, float* grads, ec_stream_t stream) {
    return ec_policy_backward2(h, params, feat, nullptr, feat_bf16, masks, T, N, workspace, ws_bytes, dhv, dh_final, grads, stream);
}

extern "C" int ec_policy_backward2(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                                   const float* masks, int T, int N, void* workspace, size_t ws_bytes, const float* dhv,
                                   const float* dh_final, float* grads, ec_stream_t stream) {
    return ec_policy_backward3(h, params, feat, feat2, feat_bf16, masks, T, N, workspace, ws_bytes, dhv, dh_final, grads, nullptr,
                               stream);
}

extern "C" int ec_policy_backward3(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                                   const float* masks, int T, int N, void* workspace, size_t ws_bytes, const float* dhv,
                                   const float* dh_final, float* grads, ec_event_t recurrent_grads_ready, ec_stream_t stream) {
    if (!h || !params || !feat || !masks || !workspace || !dhv || !grads) return EC_ERR_ARG;
    if (h->c.dual && !feat2) return EC_ERR_ARG;
    if (T <= 0 || N <= 0) return EC_ERR_SHAPE;
    const ec_policy_cfg& c = h->c;
    const Ws w = layout(h, T, N, true);
    if (ws_bytes < w.end * 4) return EC_ERR_WORKSPACE;
    float* ws = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const int B = T * N, S = c.spatial * c.spatial, H = c.hidden, A = c.num_actions, A1 = A + 1;
    const int M49 = B * S, C = c.in_channels, cat = c.compress_out + c.goal_dims;
    const int nstream = (c.dual && !c.fusion) ? 2 : 1, flat1 = c.comb_out * S;
    const int flat = c.fusion ? c.in_channels : nstream * flat1;
    auto W = [&](int i) { return params + h->off[i]; };
    auto G = [&](int i) { return grads + h->off[i]; };
    const int* goal32 = (const int*)(ws + w.goal32);
    // column sums (bias gradients) without atomics: row block y leaves its sums in cpart[y][Ncol] (the forward's per-step `gh`
    // scratch: N x 3H floats, idle in the backward), colsum_fold_kernel adds the row blocks in order onto the gradient
    float* cpart = ws + w.gh;
    const size_t cpart_cap = (size_t)N * 3 * H;
    auto colsum = [&](const float* Y, float* out, long M, int Ncol, int ld) {
        const bool wide = (Ncol & 3) == 0 && (ld & 3) == 0 && Ncol >= 256 && M >= 1024;
        long nby = (M + (wide ? 255 : 2047)) / (wide ? 256 : 2048);
        const long fit = (long)(cpart_cap / (size_t)Ncol);
        if (nby > fit) nby = fit;
        if (nby < 1) nby = 1;                                   // (Ncol <= 3H always: at least one row block fits)
        const int rpb = (int)(((M + nby - 1) / nby + 3) / 4 * 4);
        nby = (M + rpb - 1) / rpb;
        if (wide) hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)((Ncol + 255) / 256), (unsigned)nby), dim3(256), 0, s, Y, cpart, M, Ncol, ld, rpb);
        else hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((Ncol + 63) / 64), (unsigned)nby), dim3(256), 0, s, Y, cpart, M, Ncol, ld, rpb);
        hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((Ncol + 255) / 256)), dim3(256), 0, s, cpart, (int)nby, Ncol, out);
    };
    // weight-gradient GEMMs that split K: every K slice writes its own partial matrix (EC_GEMM_SPLIT_PARTS) into `sparts` and
    // splitk_fold_kernel adds the slices in order onto the gradient -- the fp32 atomics this replaces made two runs from one
    // seed differ in the last bits.  sparts: the dx area (free until dx / dx4 is computed, behind the weight gradients; with
    // the re-ordered weight_ih and in fusion mode it is never used at all).
    float* sparts = ws + w.dx;
    const size_t sparts_cap = (size_t)B * flat;
    auto gemm_acc_split = [&](const void* A_, const void* B_, float* dW, int Mo, int No, long K, long sam, long sak, long sbk, long sbn,
                              int ldc, int flags, int sk) {
        // every K slice costs a partial matrix written and read back: no more slices than fill the chip twice with 128 x 128 tiles
        const long tiles_ = (long)((Mo + 127) / 128) * ((No + 127) / 128);
        const int sk_cap = (int)std::max<long>(2, 512 / (tiles_ > 0 ? tiles_ : 1));
        if (sk > sk_cap) sk = sk_cap;
        while (sk > 1 && (size_t)sk * Mo * No > sparts_cap) --sk;
        if (sk <= 1)
            return ec_gemm_f32(A_, B_, dW, Mo, No, (int)K, sam, sak, sbk, sbn, ldc, EC_GEMM_ACCUMULATE | flags, nullptr, nullptr, nullptr, 0,
                               nullptr, nullptr, 1, stream);
        const int rc = ec_gemm_f32(A_, B_, sparts, Mo, No, (int)K, sam, sak, sbk, sbn, No, EC_GEMM_SPLIT_PARTS | flags, nullptr, nullptr,
                                   nullptr, 0, nullptr, nullptr, sk, stream);
        if (rc != EC_OK) return rc;
        const long nq = (long)Mo * No;
        hipLaunchKernelGGL(splitk_fold_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, sparts, sk, Mo, No, ldc, dW);
        return (int)EC_OK;
    };
    // ---- heads ----
    RC(ec_gemm_f32(dhv, W(P_WA), ws + w.dhs, B, H, A, A1, 1, H, 1, H, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                   1, stream));
    RC(ec_gemm_f32(dhv + A, W(P_WC), ws + w.dhs, B, H, 1, A1, 1, H, 1, H, EC_GEMM_ACCUMULATE, nullptr, nullptr, nullptr,
                   0, nullptr, nullptr, 1, stream));
    RC(gemm_acc_split(dhv, ws + w.hs, G(P_WA), A, H, B, 1, A1, H, 1, H, 0, pick_splitk(A, H, B)));
    RC(gemm_acc_split(dhv + A, ws + w.hs, G(P_WC), 1, H, B, 1, A1, H, 1, H, 0, pick_splitk(1, H, B)));
    colsum(dhv, G(P_BA), B, A, A1);
    colsum(dhv + A, G(P_BC), B, 1, A1);
    // ---- GRU, reverse time ----
    if (dh_final) (void)hipMemcpyAsync(ws + w.dhc, dh_final, (size_t)N * H * 4, hipMemcpyDeviceToDevice, s);
    else (void)hipMemsetAsync(ws + w.dhc, 0, (size_t)N * H * 4, s);
    // EC_GRU_FUSED (default 1) and H % 256 == 0: step T-1 is the plain gate kernel (its carry is dh_final), every earlier
    // step ONE fused launch (gru_step_bwd_kernel: back-projection of step t+1 + gates of step t); else GEMM + gate kernel
    const bool fused_bstep = ec_config().gru_fused && (H % 256) == 0 && T > 1;
    if (fused_bstep) {
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)(H / 32), (unsigned)(3 * H / 32)), dim3(256), 0, s, W(P_WHH),
                           ws + w.whhT, 3 * H, H);
        static std::atomic<uint64_t> attr_done{0};
        if (auto attr_g_ = ec_attr_needed(attr_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
    }
    const size_t gb_lds = (size_t)2 * 32 * (256 + 4) * sizeof(float);
    const bool btile16 = fused_bstep && ec_config().gru_fused >= 2 && H == 512;
    const size_t gb16_lds = (size_t)(4 * 2 * 32 * 128 + 4 * 256) * sizeof(float);
    if (btile16) {
        static std::atomic<uint64_t> attr16_done{0};
        if (auto attr_g_ = ec_attr_needed(attr16_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_bwd512_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    for (int t = T - 1; t >= 0; --t) {
        const size_t o3 = (size_t)t * N * 3 * H, o1 = (size_t)t * N * H;
        if (btile16 && t < T - 1) {
            hipLaunchKernelGGL(gru_step_bwd512_kernel, dim3(32u, (unsigned)((N + 15) / 16)), dim3(256), gb16_lds, s,
                               ws + w.dghb + o3 + (size_t)N * 3 * H, ws + w.whhT, masks + (size_t)(t + 1) * N, ws + w.dhs + o1,
                               ws + w.dhc, ws + w.gates + o3, ws + w.hn + o1, ws + w.hp + o1, masks + (size_t)t * N,
                               ws + w.dgi + o3, ws + w.dghb + o3, N);
            continue;
        }
        if (fused_bstep && t < T - 1) {
            hipLaunchKernelGGL(gru_step_bwd_kernel, dim3((unsigned)(H / 32), (unsigned)((N + 31) / 32)), dim3(256), gb_lds, s,
                               ws + w.dghb + o3 + (size_t)N * 3 * H, ws + w.whhT, masks + (size_t)(t + 1) * N, ws + w.dhs + o1,
                               ws + w.dhc, ws + w.gates + o3, ws + w.hn + o1, ws + w.hp + o1, masks + (size_t)t * N,
                               ws + w.dgi + o3, ws + w.dghb + o3, N, H);
            continue;
        }
        hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((N * H + 255) / 256), dim3(256), 0, s, ws + w.dhs + o1, ws + w.dhc,
                           ws + w.gates + o3, ws + w.hn + o1, ws + w.hp + o1, masks + (size_t)t * N, ws + w.dgi + o3,
                           ws + w.dghb + o3, N, H);
        if (fused_bstep) continue;          // (step T-1: its back-projection runs inside the next launch)
        // dh_carry += m * (dghb @ W_hh)
        RC(ec_gemm_f32(ws + w.dghb + o3, W(P_WHH), ws + w.dhc, N, H, 3 * H, 3 * H, 1, H, 1, H, EC_GEMM_ACCUMULATE, nullptr,
                       nullptr, nullptr, 0, nullptr, masks + (size_t)t * N, step_splitk(N, H, 3 * H), stream));
    }
    // EC_GEMM_BWD3: the large gradient GEMMs (weight gradients over all T*N rows, dx = dgi @ W_ih) on the three leading bf16
    // products of the bf16x3 split (relative product error 2^-16; the parameters' gradients then agree with the fp32 oracle to
    // ~1e-5 instead of ~1e-6)
    const int bwd3 = (ec_config().gemm_bwd3 || ec_config().policy_fast) ? EC_GEMM_3PRODUCTS : 0;
    // weight grads of the recurrence / input projection (TN over all T*N rows)
    bool parts_ok = true;       // (cleared once dx / dx4 occupies the partial-matrix area)
    auto tn = [&](const float* dY, int ldy, const void* X, int ldx, int x_bf16, float* dW, int Mo, int No, long K,
                  int ldc) {
        // grads += ...: K slices into partial matrices folded in order while the dx area is free (`parts_ok`); the tail's
        // unfused fallback GEMMs (EC_TAIL_FUSED=0, the dual encoder) run after dx exists and keep the atomic split
        const int sk = pick_splitk(Mo, No, K);
        if (parts_ok) return gemm_acc_split(dY, X, dW, Mo, No, K, 1, ldy, ldx, 1, ldc, (x_bf16 ? EC_GEMM_B_BF16 : 0) | bwd3, sk);
        return ec_gemm_f32(dY, X, dW, Mo, No, (int)K, 1, ldy, ldx, 1, ldc,
                           EC_GEMM_ACCUMULATE | (x_bf16 ? EC_GEMM_B_BF16 : 0) | bwd3, nullptr, nullptr, nullptr, 0, nullptr,
                           nullptr, sk, stream);
    };
    // EC_DW_TRANSPOSED (default 1): the GRU's two weight-gradient GEMMs contract over the T*N rows, i.e. BOTH operands are
    // strided in K -- the slowest staging of gemm_x3 (1.02 ms for dW_ih at 128 actors against 0.49 ms for the same-size input
    // projection).  Transposing both operands first (two bandwidth-bound passes, ~0.1 ms) makes it the K-contiguous case.
    const bool dw_t = ec_config().dw_transposed && B >= 1024;
    auto tn_t = [&](const float* dY, const float* X, float* dW, int Mo, int No, long K, int ldc) {   // ld(dY) == Mo, ld(X) == No
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((Mo + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0, s, dY,
                           ws + w.tA, (int)K, Mo);
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((No + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0, s, X,
                           ws + w.tB, (int)K, No);
        return gemm_acc_split(ws + w.tA, ws + w.tB, dW, Mo, No, K, K, 1, 1, K, ldc, bwd3, pick_splitk(Mo, No, K));
    };
    if (dw_t) RC(tn_t(ws + w.dghb, ws + w.hp, G(P_WHH), 3 * H, H, B, H));
    else
    RC(tn(ws + w.dghb, 3 * H, ws + w.hp, H, 0, G(P_WHH), 3 * H, H, B, H));
    colsum(ws + w.dghb, G(P_BHH), B, 3 * H, 3 * H);
    const bool wih_perm = ec_config().wih_perm && !c.fusion && !c.dual && (size_t)c.comb_out * S * 4 <= 64 * 1024;   // (== ec_policy_forward's)
    if (wih_perm) {   // gradient in the re-ordered weight's column order, then added back in the parameter's order
        (void)hipMemsetAsync(ws + w.gwihP, 0, (size_t)3 * H * flat * 4, s);
        if (dw_t) RC(tn_t(ws + w.dgi, ws + w.x4, ws + w.gwihP, 3 * H, flat, B, flat));
        else
        RC(tn(ws + w.dgi, 3 * H, ws + w.x4, flat, 0, ws + w.gwihP, 3 * H, flat, B, flat));
        hipLaunchKernelGGL(permute_row_kernel, dim3((unsigned)(3 * H)), dim3(256), flat * sizeof(float), s, ws + w.gwihP,
                           G(P_WIH), S, c.comb_out, 1);
    } else
    if (dw_t) RC(tn_t(ws + w.dgi, ws + w.x, G(P_WIH), 3 * H, flat, B, flat));
    else
    RC(tn(ws + w.dgi, 3 * H, ws + w.x, flat, 0, G(P_WIH), 3 * H, flat, B, flat));
    colsum(ws + w.dgi, G(P_BIH), B, 3 * H, 3 * H);
    // the gradients of the GRU and of both heads (tensors rnn.weight_ih_l0 .. critic.fc.bias: 92 % of the bucket) are final
    // here; what follows only writes the goal encoder's.  A caller that sums the bucket over ranks starts that section's
    // all-reduce behind this event, under the rest of this backward
    if (recurrent_grads_ready && hipEventRecord((hipEvent_t)recurrent_grads_ready, s) != hipSuccess) return EC_ERR_LAUNCH;
    if (c.fusion) {   // the image embedding and the goal table are frozen: nothing trainable upstream of the GRU
        EC_CHECK_LAUNCH();
        return EC_OK;
    }
    parts_ok = wih_perm;        // (the plain path writes dx into the partial-matrix area now; with the re-ordered weight_ih it stays free)
    // dx = dgi @ W_ih
    if (wih_perm) { // dx4 = dgi @ (re-ordered weight_ih): already pixel-major.  The weight is transposed first (9.6 MB, into
                    // the gradient staging buffer, free again by now) so that BOTH operands are K-contiguous: the GEMM's
                    // strided-B staging runs at half the rate (263 vs ~130 us at 32 actors, 770 vs ~540 at 128)
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((flat + 31) / 32), (unsigned)(3 * H / 32)), dim3(256), 0, s,
                           ws + w.wihP, ws + w.gwihP, 3 * H, flat);
        RC(ec_gemm_f32(ws + w.dgi, ws + w.gwihP, ws + w.dx4, B, flat, 3 * H, 3 * H, 1, 1, 3 * H, flat, bwd3, nullptr, nullptr,
                       nullptr, 0, nullptr, nullptr, 1, stream));
    }
    else
        RC(ec_gemm_f32(ws + w.dgi, W(P_WIH), ws + w.dx, B, flat, 3 * H, 3 * H, 1, flat, 1, flat, bwd3, nullptr, nullptr, nullptr,
                       0, nullptr, nullptr, 1, stream));
    for (int sidx = 0; sidx < nstream; ++sidx) {   // dual encoder: the depth stream re-uses the gradient temporaries (one HIP stream)
    const void* featS = sidx ? feat2 : feat;
    auto WS = [&](int i) { return params + h->off[(sidx && i >= P_W1 && i <= P_B4) ? i + (P_W1D - P_W1) : i]; };
    auto GS = [&](int i) { return grads + h->off[(sidx && i >= P_W1 && i <= P_B4) ? i + (P_W1D - P_W1) : i]; };
    const size_t o_c1 = sidx ? w.c1d : w.c1, o_c2 = sidx ? w.c2d : w.c2, o_m1 = sidx ? w.m1d : w.m1;
    if (!wih_perm) {
        const long total = (long)B * flat1;
        hipLaunchKernelGGL(from_cmajor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws + w.dx,
                           ws + w.dx4, S, c.comb_out, total, flat, sidx * flat1);
    }
    const int tail_fused = ec_config().tail_fused;
    const size_t tb_lds = ((size_t)2 * 128 * TL_P32 + 32 * TL_P128 + std::max<size_t>((size_t)c.num_goals * 128, 4 * 256) +
                           4 * (size_t)(32 * TL_P128 + 32 * TL_P32)) * sizeof(float);   // (the sE area: four wave-private 256-float lines)
    const bool fused_bwd = tail_fused && c.compress_hid == 128 && c.compress_out == 32 && c.comb_hid == 128 &&
                           c.comb_out == 32 && S >= 32 && tb_lds <= 160 * 1024 && c.num_goals <= TB_MAXG;
    // EC_DW1_TR (default 1): dW1 through the transpose-read kernel (dw_tn.hip); dc1 then only exists as bf16 planes
    const int dw1_tr = ec_config().dw1_tr;
    const bool dw1_planes = fused_bwd && dw1_tr && feat_bf16 && C % 256 == 0 && M49 >= 2048 &&
                            (size_t)ec_dw_tn_x3_splits(M49, C) * 128 * C <= (size_t)TB_MAX_WG * 4 * TB_PART;
    (void)hipMemsetAsync(ws + w.dE1, 0, (size_t)c.num_goals * c.comb_hid * 4, s);
    if (fused_bwd) {
        // EC_TAIL_FUSED (default 1): dm1 / dc2 / dc1 and all small weight gradients of the tail in one pass
        static std::atomic<uint64_t> attr_done{0};
        if (auto attr_g_ = ec_attr_needed(attr_done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        const long ntiles = ((long)M49 + 31) / 32;
        long nwg = (ntiles + 3) / 4;
        if (nwg > TB_MAX_WG) nwg = TB_MAX_WG;
        hipLaunchKernelGGL(tail_bwd_kernel, dim3((unsigned)nwg), dim3(256), tb_lds, s, ws + w.dx4, ws + o_m1, ws + o_c2, ws + o_c1,
                           WS(P_W2), WS(P_W3), cat, WS(P_W4), goal32, S, c.num_goals, ws + w.dc1,
                           dw1_planes ? (uint16_t*)(ws + w.dc1) : nullptr, ws + w.tpart, ws + w.tpartE,
                           (long)M49);
        const int ne = TB_PART + c.num_goals * 128;
        // the y-slices' sums: behind the used partial sets when there is room (nwg < TB_MAX_WG), else in the dm1 area, which
        // the fused path never materialises (nwg == TB_MAX_WG means M49 >= 32,768 rows: 4 M floats)
        float* red = (nwg < TB_MAX_WG) ? ws + w.tpart + (size_t)nwg * 4 * TB_PART : ws + w.dm1;
        const size_t red_cap = (nwg < TB_MAX_WG) ? (size_t)(TB_MAX_WG - nwg) * 4 * TB_PART : (size_t)M49 * c.comb_hid;
        int ny = (int)std::min<size_t>(8, red_cap / (size_t)ne);
        if (ny < 1) return EC_ERR_WORKSPACE;
        hipLaunchKernelGGL(tail_bwd_reduce_kernel, dim3((unsigned)((ne + 255) / 256), (unsigned)ny), dim3(256), 0, s, ws + w.tpart,
                           (int)nwg * 4, ws + w.tpartE, c.num_goals, red);
        hipLaunchKernelGGL(tail_bwd_fold_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, red, ny, c.num_goals, GS(P_W4),
                           GS(P_W3), cat, GS(P_W2), GS(P_B4), GS(P_B2), GS(P_B1), ws + w.dE1);
    } else {
    // ---- target_obs_combiner ----
    RC(tn(ws + w.dx4, c.comb_out, ws + o_m1, c.comb_hid, 0, GS(P_W4), c.comb_out, c.comb_hid, M49, c.comb_hid));
    colsum(ws + w.dx4, GS(P_B4), M49, c.comb_out, c.comb_out);
    RC(ec_gemm_f32(ws + w.dx4, WS(P_W4), ws + w.dm1, M49, c.comb_hid, c.comb_out, c.comb_out, 1, c.comb_hid, 1, c.comb_hid,
                   0, nullptr, nullptr, nullptr, 0, ws + o_m1, nullptr, 1, stream));
    RC(tn(ws + w.dm1, c.comb_hid, ws + o_c2, c.compress_out, 0, GS(P_W3), c.comb_hid, c.compress_out, M49, cat));
    hipLaunchKernelGGL(group_sum_scatter_kernel, dim3((unsigned)B), dim3(128), 0, s, ws + w.dm1, goal32, ws + w.dE1, S,
                       c.comb_hid, (long)B);
    // ---- resnet_compressor ----
    RC(ec_gemm_f32(ws + w.dm1, WS(P_W3), ws + w.dc2, M49, c.compress_out, c.comb_hid, c.comb_hid, 1, cat, 1,
                   c.compress_out, 0, nullptr, nullptr, nullptr, 0, ws + o_c2, nullptr, 1, stream));
    RC(tn(ws + w.dc2, c.compress_out, ws + o_c1, c.compress_hid, 0, GS(P_W2), c.compress_out, c.compress_hid, M49,
          c.compress_hid));
    colsum(ws + w.dc2, GS(P_B2), M49, c.compress_out, c.compress_out);
    RC(ec_gemm_f32(ws + w.dc2, WS(P_W2), ws + w.dc1, M49, c.compress_hid, c.compress_out, c.compress_out, 1,
                   c.compress_hid, 1, c.compress_hid, 0, nullptr, nullptr, nullptr, 0, ws + o_c1, nullptr, 1, stream));
    colsum(ws + w.dc1, GS(P_B1), M49, c.compress_hid, c.compress_hid);
    }
    // goal half of target_obs_combiner.0 (dE1 = row-group sums of dm1 scattered by goal id)
    colsum(ws + w.dE1, GS(P_B3), c.num_goals, c.comb_hid, c.comb_hid);
    RC(tn(ws + w.dE1, c.comb_hid, WS(P_EMB), c.goal_dims, 0, GS(P_W3) + c.compress_out, c.comb_hid, c.goal_dims,
          c.num_goals, cat));
    RC(ec_gemm_f32(ws + w.dE1, WS(P_W3) + c.compress_out, GS(P_EMB), c.num_goals, c.goal_dims, c.comb_hid, c.comb_hid, 1,
                   cat, 1, c.goal_dims, EC_GEMM_ACCUMULATE, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 1, stream));
    if (dw1_planes) RC(ec_dw_tn_xp(ws + w.dc1, featS, ws + w.tpart, GS(P_W1), M49, C, ec_config().policy_fast ? 2 : 3, stream));   // (tpart: free again after the reducer)
    else RC(tn(ws + w.dc1, c.compress_hid, featS, C, feat_bf16, GS(P_W1), c.compress_hid, C, M49, C));
    }   // streams
    EC_CHECK_LAUNCH();
    return EC_OK;
}
