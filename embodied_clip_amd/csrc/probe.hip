// Linear-probe heads of primitive_probing/train.py (BASELINE config 1).
//
// Replaces (reference file:line):
//   LinearEncoder.model heads           primitive_probing/train.py:19-49
//     Linear + Sigmoid | Linear + Softmax(dim=1) | AdaptiveAvgPool2d(3,3) + Conv1x1 + Flatten(2) + Sigmoid
//   LinearEncoder.compute_loss          primitive_probing/train.py:56-92
//     F.binary_cross_entropy on probabilities (:76), F.cross_entropy applied to the Softmax OUTPUT for
//     `free_space` (:35,78 -- the double softmax is reproduced, not fixed), the reachability column gather
//     (:61-63,72), the label clamp y[y > max_forward_steps] = max_forward_steps (:65), and the metric counts
//     behind MF.f1 / thresholded accuracy / argmax accuracy (:84-90)
//   and their autograd backward (loss.backward() under pytorch-lightning's training_step, :94-97).
//
// The Linear / Conv1x1 contraction itself is ec_gemm_f32 (bf16x3 MFMA); everything here is a few thousand
// elements, so each kernel is ONE workgroup with a fixed reduction order: results are bit-reproducible.
#include <math.h>

#include "common.h"

namespace {

constexpr int PT = 1024;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block reduction of NV doubles over PT threads; result valid on thread 0..NV-1 via `red`
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* out) {
    __shared__ double red[PT / 64][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum_d(v[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int w = 0; w < PT / 64; ++w) s += red[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// torch.binary_cross_entropy: -(y*max(log p,-100) + (1-y)*max(log(1-p),-100))
__device__ __forceinline__ float bce_term(float p, float y) {
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
    return -(y * lp + (1.f - y) * l1p);
}
// d loss / d logit through BCE(sigmoid): torch = (p - y) / max(p(1-p), 1e-12) * p(1-p)
__device__ __forceinline__ float bce_sigmoid_grad(float p, float y) {
    const float q = p * (1.f - p);
    return (p - y) / fmaxf(q, 1e-12f) * q;
}

// mode 0: dense sigmoid + BCE over [R, C]; out5 = {sum loss, tp, pred_pos, true_pos, correct(thr)}
__global__ __launch_bounds__(PT) void probe_bce_dense(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                      float* __restrict__ pred, float* __restrict__ dz, int R, int C,
                                                      double* __restrict__ out5) {
    double acc[5] = {0, 0, 0, 0, 0};
    const long n = (long)R * C;
    const float inv = 1.f / (float)n;
    for (long i = threadIdx.x; i < n; i += PT) {
        const float p = 1.f / (1.f + expf(-z[i]));
        const float t = (float)y[i];
        acc[0] += (double)bce_term(p, t);
        const bool pp = p > 0.5f, tt = y[i] != 0;
        acc[1] += (pp && tt) ? 1.0 : 0.0;
        acc[2] += pp ? 1.0 : 0.0;
        acc[3] += tt ? 1.0 : 0.0;
        acc[4] += (pp == tt) ? 1.0 : 0.0;
        if (pred) pred[i] = p;
        if (dz) dz[i] = bce_sigmoid_grad(p, t) * inv;
    }
    block_sum<5>(acc, out5);
}

// mode 1: sigmoid over [R, C], loss only on column idx[r] (reachability)
__global__ __launch_bounds__(PT) void probe_bce_gather(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                       const int64_t* __restrict__ idx, float* __restrict__ pred,
                                                       float* __restrict__ dz, int R, int C,
                                                       double* __restrict__ out5) {
    double acc[5] = {0, 0, 0, 0, 0};
    const float inv = 1.f / (float)R;
    for (long i = threadIdx.x; i < (long)R * C; i += PT) {
        const float p = 1.f / (1.f + expf(-z[i]));
        if (pred) pred[i] = p;
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        float g = 0.f;
        if ((int64_t)c == idx[r]) {
            const float t = (float)y[r];
            acc[0] += (double)bce_term(p, t);
            const bool pp = p > 0.5f, tt = y[r] != 0;
            acc[1] += (pp && tt) ? 1.0 : 0.0;
            acc[2] += pp ? 1.0 : 0.0;
            acc[3] += tt ? 1.0 : 0.0;
            acc[4] += (pp == tt) ? 1.0 : 0.0;
            g = bce_sigmoid_grad(p, t) * inv;
        }
        if (dz) dz[i] = g;
    }
    block_sum<5>(acc, out5);
}

// mode 2: p = softmax(z); loss = cross_entropy(p, y) = -log_softmax(p)[y]  (double softmax, train.py:35,78)
// one lane per row (C <= 64)
__global__ __launch_bounds__(PT) void probe_softmax_ce2(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                        float* __restrict__ pred, float* __restrict__ dz, int R, int C,
                                                        int label_clamp, double* __restrict__ out5) {
    double acc[5] = {0, 0, 0, 0, 0};
    const float inv = 1.f / (float)R;
    for (int r = threadIdx.x; r < R; r += PT) {
        const float* zr = z + (long)r * C;
        int64_t t = y[r];
        if (label_clamp >= 0 && t > label_clamp) t = label_clamp;
        t = t < 0 ? 0 : (t >= C ? C - 1 : t);   // torch raises on out-of-range targets; never index out of bounds
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, zr[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(zr[c] - m);
        // second softmax over the probabilities
        float pm = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float p = expf(zr[c] - m) / s;
            if (p > pm) { pm = p; arg = c; }
        }
        float s2 = 0.f;
        for (int c = 0; c < C; ++c) s2 += expf(expf(zr[c] - m) / s - pm);
        const float lse2 = pm + logf(s2);
        const float pt = expf(zr[t] - m) / s;
        acc[0] += (double)(lse2 - pt);
        acc[4] += ((int64_t)arg == t) ? 1.0 : 0.0;
        if (pred)
            for (int c = 0; c < C; ++c) pred[(long)r * C + c] = expf(zr[c] - m) / s;
        if (dz) {
            // dp_c = (softmax(p)_c - [c==t]) / R;  dz_c = p_c (dp_c - sum_k dp_k p_k)
            float dot = 0.f;
            for (int c = 0; c < C; ++c) {
                const float p = expf(zr[c] - m) / s;
                const float dp = (expf(p - lse2) - ((int64_t)c == t ? 1.f : 0.f)) * inv;
                dot += dp * p;
            }
            for (int c = 0; c < C; ++c) {
                const float p = expf(zr[c] - m) / s;
                const float dp = (expf(p - lse2) - ((int64_t)c == t ? 1.f : 0.f)) * inv;
                dz[(long)r * C + c] = p * (dp - dot);
            }
        }
    }
    block_sum<5>(acc, out5);
}

// db[c] = sum_r dz[r, c]   (fixed order)
__global__ void probe_bias_grad(const float* __restrict__ dz, float* __restrict__ db, int R, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dz[(long)r * C + c];
    db[c] = s;
}

// AdaptiveAvgPool2d((3,3)) of fp32 NCHW [B, C, H, W] -> rows [B*9, C] (cell-major within a frame), so that the
// following Conv1x1 is a plain row GEMM whose [B*9, 52] output IS `y_pred.permute(0,2,1).flatten(1)` (train.py:70).
// bin i = [floor(i*H/3), ceil((i+1)*H/3))  -> 7x7: [0,3) [2,5) [4,7)
__global__ __launch_bounds__(256) void probe_pool3(const float* __restrict__ x, float* __restrict__ out, int C, int H,
                                                   int W) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* p = x + ((long)b * C + c) * H * W;
    for (int i = 0; i < 3; ++i) {
        const int y0 = (i * H) / 3, y1 = ((i + 1) * H + 2) / 3;
        for (int j = 0; j < 3; ++j) {
            const int x0 = (j * W) / 3, x1 = ((j + 1) * W + 2) / 3;
            float s = 0.f;
            for (int yy = y0; yy < y1; ++yy)
                for (int xx = x0; xx < x1; ++xx) s += p[yy * W + xx];
            out[((long)b * 9 + i * 3 + j) * C + c] = s / (float)((y1 - y0) * (x1 - x0));
        }
    }
}

}  // namespace

extern "C" int ec_probe_pool3(const float* conv_nchw, float* rows, int B, int C, int H, int W, ec_stream_t stream) {
    if (!conv_nchw || !rows) return EC_ERR_ARG;
    if (B <= 0 || C <= 0 || H < 3 || W < 3) return EC_ERR_SHAPE;
    hipLaunchKernelGGL(probe_pool3, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, conv_nchw, rows, C, H, W);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_probe_head(int mode, const float* logits, const int64_t* labels, const int64_t* idx, int R, int C,
                             int label_clamp, float* pred, float* dlogits, float* dbias, double* out5,
                             ec_stream_t stream) {
    if (!logits || !labels || !out5) return EC_ERR_ARG;
    if (R <= 0 || C <= 0) return EC_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case EC_PROBE_SIGMOID_BCE:
            hipLaunchKernelGGL(probe_bce_dense, dim3(1), dim3(PT), 0, s, logits, labels, pred, dlogits, R, C, out5);
            break;
        case EC_PROBE_SIGMOID_BCE_GATHER:
            if (!idx) return EC_ERR_ARG;
            hipLaunchKernelGGL(probe_bce_gather, dim3(1), dim3(PT), 0, s, logits, labels, idx, pred, dlogits, R, C, out5);
            break;
        case EC_PROBE_SOFTMAX_CE2:
            if (C > 64) return EC_ERR_SHAPE;
            hipLaunchKernelGGL(probe_softmax_ce2, dim3(1), dim3(PT), 0, s, logits, labels, pred, dlogits, R, C, label_clamp,
                               out5);
            break;
        default:
            return EC_ERR_ARG;
    }
    EC_CHECK_LAUNCH();
    if (dbias) {
        if (!dlogits) return EC_ERR_ARG;
        hipLaunchKernelGGL(probe_bias_grad, dim3((C + 63) / 64), dim3(64), 0, s, dlogits, dbias, R, C);
        EC_CHECK_LAUNCH();
    }
    return EC_OK;
}
