// GAE returns, advantage normalisation, PPO loss (forward + analytic backward),
// categorical sampling, global-norm clip + Adam on the flat parameter bucket.
//
// Replaces ([U] allenai/allenact ~v0.5.0):
//   RolloutStorage.compute_returns(use_gae=True)      (onpolicy_sync/storage.py)
//   PPO.loss_per_step / PPO.loss                       (onpolicy_sync/losses/ppo.py)
//   CategoricalDistr.sample / log_prob / entropy       (base_abstractions/distributions.py)
//   clip_grad_norm_(max_grad_norm) + Adam.step()       (onpolicy_sync/engine.py backprop_step)
// All fp32; reductions accumulate in fp64 so the result does not depend on
// the launch geometry beyond fp32 rounding of the inputs.
#include <math.h>

#include "common.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// Block reduction of up to NV doubles in a FIXED order (wave butterflies, then the waves' sums in wave order): thread 0 returns
// the totals.  No atomics anywhere in this file's reductions (round 6): a floating-point atomicAdd per block made the advantage
// statistics, the loss sums and the gradient norm depend on the order the blocks happened to retire in -- two runs from one
// seed differed in the last bits, and Adam's sign-like steps amplify that.
template <int NV, int NWAVES>
__device__ __forceinline__ void block_sum(double (&v)[NV], double (*red)[NV]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double t = 0.0;
            for (int w = 0; w < NWAVES; ++w) t += red[w][i];
            v[i] = t;
        }
    }
}

// One lane per sampler: reverse scan over T.
//   delta = r[t] + g*V[t+1]*m[t+1] - V[t];  gae = delta + g*tau*m[t+1]*gae;  R[t] = gae + V[t]
// adv[t] = R[t] - V[t]; also accumulates sum(adv), sum(adv^2) for the normalisation.
__global__ __launch_bounds__(1024) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ msk,
                                                  float* __restrict__ ret, float* __restrict__ adv, double* __restrict__ stats, int T, int N,
                                                  float gamma, float tau) {
    // ONE workgroup: every sampler's scan is a serial chain over T anyway, and the statistics then need no cross-block sum
    double acc[2] = {0.0, 0.0};
    for (int n = threadIdx.x; n < N; n += 1024) {
        float gae = 0.f;
        ret[(long)T * N + n] = val[(long)T * N + n];
        for (int t = T - 1; t >= 0; --t) {
            const float m1 = msk[(long)(t + 1) * N + n];
            const float v = val[(long)t * N + n];
            const float delta = rew[(long)t * N + n] + gamma * val[(long)(t + 1) * N + n] * m1 - v;
            gae = delta + gamma * tau * m1 * gae;
            ret[(long)t * N + n] = gae + v;
            adv[(long)t * N + n] = gae;
            acc[0] += (double)gae;
            acc[1] += (double)gae * (double)gae;
        }
    }
    __shared__ double red[16][2];
    block_sum<2, 16>(acc, red);
    if (threadIdx.x == 0) { stats[0] = acc[0]; stats[1] = acc[1]; }
}

// norm_adv = (adv - mean) / (std_unbiased + eps)
__global__ void adv_norm_kernel(const float* __restrict__ adv, const double* __restrict__ stats, float* __restrict__ out,
                                long n, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double mean = stats[0] / (double)n;
    double var = (stats[1] - (double)n * mean * mean) / (double)(n > 1 ? n - 1 : 1);
    if (var < 0) var = 0;
    out[i] = (float)(((double)adv[i] - mean) / (sqrt(var) + (double)eps));
}

// PPO loss, forward + backward in one pass over the [B] steps.
// hv[b, 0:A] = logits, hv[b, A] = value.  sums[0..3] += {action, value, entropy(-H), ratio}
// dhv = d(total)/d(hv) * grad_scale with total = mean(La) + vc*mean(Lv) + ec*mean(Le).
template <int MAXA>
__global__ __launch_bounds__(1024) void ppo_loss_kernel(const float* __restrict__ hv, const long long* __restrict__ actions,
                                const float* __restrict__ old_logp, const float* __restrict__ old_val,
                                const float* __restrict__ returns, const float* __restrict__ nadv,
                                float* __restrict__ dhv, double* __restrict__ sums, long B, int A, float clip,
                                float vclip, float vcoef, float ecoef, float grad_scale) {
    // ONE workgroup of 1024 threads walks the B steps (a few thousand per actor slice: ~10 us): the four loss sums are then
    // folded in a fixed order inside it -- no cross-block atomics, bit-identical run to run
    double acc[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (long i = threadIdx.x; i < B; i += 1024) {                   // (unrolled: four elements' loads and exp / log chains in flight per thread)
        const float* row = hv + i * (A + 1);
        float lg[MAXA];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) {
            lg[k] = (k < A) ? row[k] : -INFINITY;
            mx = fmaxf(mx, lg[k]);
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) se += (k < A) ? expf(lg[k] - mx) : 0.f;
        const float lse = mx + logf(se);
        const int a = (int)actions[i];
        float ent = 0.f, lp_a = 0.f;
        float pr[MAXA], lpk[MAXA];
#pragma unroll
        for (int k = 0; k < MAXA; ++k) {
            lpk[k] = (k < A) ? lg[k] - lse : 0.f;
            pr[k] = (k < A) ? expf(lpk[k]) : 0.f;
            ent -= pr[k] * lpk[k];
            if (k == a) lp_a = lpk[k];
        }
        const float adv = nadv[i];
        const float ratio = expf(lp_a - old_logp[i]);
        const float surr1 = ratio * adv;
        const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const float surr2 = rc * adv;
        const bool use2 = surr2 < surr1;
        const float la = -(use2 ? surr2 : surr1);
        // d(-min)/d(ratio): surr1 branch -> -adv; surr2 branch -> -adv inside the clamp range, 0 outside
        const bool in_rng = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
        const float dla_dratio = use2 ? (in_rng ? -adv : 0.f) : -adv;
        const float dla_dlp = dla_dratio * ratio;
        const float v = row[A], vo = old_val[i], R = returns[i];
        const float dv = v - vo;
        // vclip < 0: use_clipped_value_loss=False -> 0.5 * (R - V)^2
        const bool vc_on = vclip >= 0.f;
        const float vcl = vc_on ? vo + fminf(fmaxf(dv, -vclip), vclip) : v;
        const float l1 = (v - R) * (v - R), l2 = (vcl - R) * (vcl - R);
        const float lv = 0.5f * fmaxf(l1, l2);
        const bool v_in = !vc_on || ((dv >= -vclip) && (dv <= vclip));
        const float g1 = (v - R), g2 = v_in ? (vcl - R) : 0.f;
        const float dlv_dv = (l1 > l2) ? g1 : ((l2 > l1) ? g2 : 0.5f * (g1 + g2));   // torch.max splits ties evenly
        const float invB = grad_scale / (float)B;
        float* drow = dhv + i * (A + 1);
#pragma unroll
        for (int k = 0; k < MAXA; ++k)
            if (k < A) {
                const float dlp = ((k == a) ? 1.f : 0.f) - pr[k];
                const float dent = pr[k] * (lpk[k] + ent);          // d(-H)/dlogit_k
                drow[k] = invB * (dla_dlp * dlp + ecoef * dent);
            }
        drow[A] = invB * vcoef * dlv_dv;
        acc[0] += la; acc[1] += lv; acc[2] += -ent; acc[3] += ratio;
        if (a < 0 || a >= A) acc[0] = (double)NAN;   // an out-of-range action id poisons the loss instead of passing silently
    }
    __shared__ double red[16][4];
    block_sum<4, 16>(acc, red);
    if (threadIdx.x == 0) { sums[0] = acc[0]; sums[1] = acc[1]; sums[2] = acc[2]; sums[3] = acc[3]; }
}

// CategoricalDistr.sample() + log_prob(): inverse-CDF on a counter-based uniform (ec_sample_row, common.h).
template <int MAXA>
__global__ void sample_kernel(const float* __restrict__ hv, long long* __restrict__ actions, float* __restrict__ logp,
                              float* __restrict__ values, int N, int A, uint64_t seed, uint64_t step, int first_actor) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* row = hv + (long)n * (A + 1);
    int a;
    float lp;
    ec_sample_row([&](int k) { return row[k]; }, A, seed, step, (uint64_t)(n + first_actor), a, lp);   // keyed by GLOBAL actor id
    actions[n] = a;
    logp[n] = lp;
    if (values) values[n] = row[A];
}

// sum of squares of the flat gradient bucket: every block folds its stride of the bucket in a fixed order and stores its partial
// in scratch[1 + block]; the block that draws the last ticket (an INTEGER atomic: order-free) adds the partials in block order
// into scratch[0].  scratch = EC_CLIP_ADAM_SCRATCH_DOUBLES doubles: [0] total, [1 .. 1024] partials, [1025] the ticket counter.
constexpr int SUMSQ_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, double* __restrict__ scratch, long n) {
    double acc[1] = {0.0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc[0] += (double)g[i] * (double)g[i];
    __shared__ double red[4][1];
    __shared__ unsigned last;
    block_sum<1, 4>(acc, red);
    if (threadIdx.x == 0) {
        scratch[1 + blockIdx.x] = acc[0];
        __threadfence();                                             // the partial is visible device-wide before the ticket is
        unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 1 + SUMSQ_MAX_BLOCKS);
        last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();                                                 // acquire: the other blocks' partials
    // thread t adds partials t, t + 256, ... (independent loads in flight), then the same fixed-order block fold: the order
    // depends on the grid size only
    double f[1] = {0.0};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) f[0] += __hip_atomic_load(scratch + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                                 // (red is reused)
    block_sum<1, 4>(f, red);
    if (threadIdx.x == 0) {
        scratch[0] = f[0];
        *reinterpret_cast<unsigned*>(scratch + 1 + SUMSQ_MAX_BLOCKS) = 0u;
    }
}

// clip_grad_norm_ (coef = min(1, max_norm/(norm+1e-6))) fused into torch.optim.Adam's update
__global__ void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, const double* __restrict__ sumsq, long n, float max_norm,
                                 float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float norm = (float)sqrt(sumsq[0]);
        coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    }
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace

extern "C" int ec_gae(const float* rewards, const float* values, const float* masks, float* returns, float* adv,
                      float* norm_adv, double* stats2, int T, int N, float gamma, float tau, float eps,
                      ec_stream_t stream) {
    if (!rewards || !values || !masks || !returns || !adv || !stats2) return EC_ERR_ARG;
    if (T <= 0 || N <= 0) return EC_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gae_kernel, dim3(1), dim3(1024), 0, s, rewards, values, masks, returns, adv, stats2, T, N, gamma, tau);
    if (norm_adv) {
        const long n = (long)T * N;
        hipLaunchKernelGGL(adv_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, adv, stats2, norm_adv, n,
                           eps);
    }
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_ppo_loss(const float* hv, const int64_t* actions, const float* old_logp, const float* old_values,
                           const float* returns, const float* norm_adv, float* dhv, double* sums4, long B, int A,
                           float clip, float vcoef, float ecoef, float grad_scale, ec_stream_t stream) {
    return ec_ppo_loss_ex(hv, actions, old_logp, old_values, returns, norm_adv, dhv, sums4, B, A, clip, clip, vcoef, ecoef,
                          grad_scale, stream);
}

extern "C" int ec_ppo_loss_ex(const float* hv, const int64_t* actions, const float* old_logp, const float* old_values,
                              const float* returns, const float* norm_adv, float* dhv, double* sums4, long B, int A,
                              float clip, float value_clip, float vcoef, float ecoef, float grad_scale,
                              ec_stream_t stream) {
    if (!hv || !actions || !old_logp || !old_values || !returns || !norm_adv || !dhv || !sums4) return EC_ERR_ARG;
    if (B <= 0 || A <= 0 || A > 16) return EC_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (A <= 8)
        hipLaunchKernelGGL(ppo_loss_kernel<8>, dim3(1), dim3(1024), 0, s, hv, (const long long*)actions, old_logp, old_values,
                           returns, norm_adv, dhv, sums4, B, A, clip, value_clip, vcoef, ecoef, grad_scale);
    else
        hipLaunchKernelGGL(ppo_loss_kernel<16>, dim3(1), dim3(1024), 0, s, hv, (const long long*)actions, old_logp, old_values,
                           returns, norm_adv, dhv, sums4, B, A, clip, value_clip, vcoef, ecoef, grad_scale);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_sample_actions(const float* hv, int64_t* actions, float* logp, float* values, int N, int A,
                                 uint64_t seed, uint64_t step, int first_actor, ec_stream_t stream) {
    if (!hv || !actions || !logp) return EC_ERR_ARG;
    if (N <= 0 || A <= 0) return EC_ERR_SHAPE;
    hipLaunchKernelGGL(sample_kernel<16>, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, hv,
                       (long long*)actions, logp, values, N, A, seed, step, first_actor);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_clip_adam_scratch_doubles(void) { return 1 + SUMSQ_MAX_BLOCKS + 1; }

extern "C" int ec_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, double* sumsq1,
                                 long n, float max_grad_norm, float lr, float beta1, float beta2, float eps, int step,
                                 ec_stream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sumsq1) return EC_ERR_ARG;
    if (n <= 0 || step <= 0) return EC_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(sumsq1 + 1 + SUMSQ_MAX_BLOCKS, 0, sizeof(double), s);      // the ticket counter
    long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)(blocks > SUMSQ_MAX_BLOCKS ? SUMSQ_MAX_BLOCKS : blocks)), dim3(256), 0, s, grads, sumsq1, n);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq,
                       sumsq1, n, max_grad_norm, lr, beta1, beta2, eps, bc1, bc2s);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
