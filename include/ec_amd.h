/*
 * ec_amd.h -- C-ABI of the MI355X-native embodied-clip hot path.
 *
 * The reference (allenai/embodied-clip) is pure Python over torch.nn and has
 * NO FFI / operator-registration boundary of its own.  This
 * header is therefore the boundary a maintainer would bind (ctypes stubs are
 * shown in INTEGRATION.md); every entry point cites the reference-side
 * interface it replaces.  "[U]" = upstream module that the reference depends
 * on but does not vendor (openai/CLIP @40f5484c, allenai/allenact ~v0.5.0).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in signatures
 *     (ec_stream_t is a hipStream_t passed as void*; NULL = default stream).
 *   - every pointer is a DEVICE pointer unless the name starts with h_.
 *   - functions return EC_OK (0) or a negative EC_ERR_* code; they never throw
 *     and never allocate behind the caller's back: workspaces are sized by an
 *     ec_*_workspace_bytes() query and passed in.
 *   - kernels are enqueued on the given stream and are asynchronous.
 *   - one handle per process/GPU; handles are immutable after creation and may
 *     be used from one thread at a time.
 *   - "bf16" buffers are raw uint16 bfloat16; activations are NHWC
 *     (channels-last), conv weights are [Cout][KH][KW][Cin].
 */
#ifndef EC_AMD_H
#define EC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ec_stream_t;

enum {
    EC_OK = 0,
    EC_ERR_ARG = -1,         /* null pointer / bad enum */
    EC_ERR_SHAPE = -2,       /* unsupported shape for this kernel */
    EC_ERR_LAUNCH = -3,      /* HIP launch failed */
    EC_ERR_WORKSPACE = -4,   /* workspace too small */
    EC_ERR_ALLOC = -5,       /* host allocation failed */
    EC_ERR_UNSUPPORTED = -6
};

enum { EC_ACT_NONE = 0, EC_ACT_RELU = 1, EC_ACT_QUICKGELU = 2 };

int ec_version(void);
const char* ec_strerror(int code);

/* ------------------------------------------------------------------------
 * Encoder building blocks (bf16 storage, fp32 accumulate on MFMA).
 * Replace the cuDNN/cuBLAS kernels the reference triggers through
 * `clip_model(clip_input)` (primitive_probing/generate_data/thor_image_features.py:109).
 * ---------------------------------------------------------------------- */

/* Conv (1x1 or 3x3, stride 1, pad (k-1)/2) with BatchNorm folded into
 * (w,bias) per freeze_model (thor_image_features.py:26-33), fused
 * + bias, + residual, activation, and optional fused AvgPool2d(2) epilogue
 * ([U] clip/model.py Bottleneck.forward: conv-bn-relu / avgpool / add-relu).
 * in  bf16 [B,H,W,Cin]; w bf16 [Cout][k*k*Cin]; bias f32 [Cout];
 * res bf16 [B,H,W,Cout] or NULL (not with pool); out bf16 [B,H',W',Cout]
 * (H'=H/2 when pool).  Cin power of two >= 8; Cout multiple of 32. */
int ec_conv_bf16(const void* in, const void* w, const float* bias, const void* res, void* out,
                 int B, int H, int W, int Cin, int Cout, int ksize, int pool, int act,
                 ec_stream_t stream);

/* Plain GEMM view of the same kernel: out[M,N] = act(A[M,K] W[N,K]^T + bias (+res)).
 * Replaces nn.Linear / nn.MultiheadAttention projections of [U] clip/model.py
 * ResidualAttentionBlock and AttentionPool2d.  K multiple of 8, N multiple of 32. */
int ec_gemm_bf16(const void* A, const void* W, const float* bias, const void* res, void* out,
                 int M, int N, int K, int act, ec_stream_t stream);

/* Stem conv1: 3x3 stride 2 pad 1 on the fp32 NHWC frame the RGB sensor hands
 * over ([U] ClipResNetPreprocessor.process: obs[rgb].permute(0,3,1,2)), folded
 * BN + ReLU, LDS-staged image tiles.  w f32 [3*3*3][Cout] (ky,kx,ci major),
 * out bf16 [B,H/2,W/2,Cout].  Cout in {32,48}. */
int ec_stem_conv1(const float* rgb_nhwc, const float* w, const float* bias, void* out,
                  int B, int H, int W, int Cout, ec_stream_t stream);

/* AvgPool2d(2) on bf16 NHWC ([U] Bottleneck downsample "-1"). C multiple of 8. */
int ec_avgpool2_bf16(const void* in, void* out, int B, int H, int W, int C, ec_stream_t stream);

/* bf16 NHWC [B,HW,C] -> fp32 NCHW [B,C,HW]: the `.float()` + layout the
 * reference exposes (thor_image_features.py:111; observation_space (2048,7,7)). */
int ec_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int HW, int C, ec_stream_t stream);

/* AdaptiveAvgPool2d(1)+Flatten on the fp32-cast features
 * (thor_image_features.py:63-66,113; preprocessor pool=True). bf16 [B,HW,C] -> f32 [B,C]. */
int ec_spatial_mean_bf16(const void* in, float* out, int B, int HW, int C, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP ModifiedResNet trunk (stem + layer1..4, attnpool detached):
 * `clip_model.visual` with `attnpool = Identity` (thor_image_features.py:59-67,109)
 * == [U] ClipResNetEmbedder.forward.
 * ---------------------------------------------------------------------- */
typedef struct ec_rn50 ec_rn50_t;

/* Weight order contract (execution order): stem conv1 (f32, [27][w/2]) is
 * passed separately; `w_bf16` holds, concatenated, stem conv2, conv3, then per
 * block conv1, conv2, conv3, [downsample] as [Cout][k*k*Cin] bf16;
 * `bias` holds stem conv1..3 then the same per-block order, f32.
 * The handle borrows the device pointers (caller keeps them alive). */
int ec_rn50_create(ec_rn50_t** out, int width, const int* layers4, int input_resolution,
                   const float* stem_w_f32, const void* w_bf16, size_t n_w, const float* bias, size_t n_bias);
void ec_rn50_destroy(ec_rn50_t* h);
size_t ec_rn50_workspace_bytes(const ec_rn50_t* h, int batch);
int ec_rn50_out_channels(const ec_rn50_t* h);
int ec_rn50_out_spatial(const ec_rn50_t* h);
/* rgb f32 NHWC [B,R,R,3] -> feat bf16 NHWC [B,R/32,R/32,32w].  `feat` may be a
 * slice of the rollout feature buffer.  chunk>0 runs the trunk in sub-batches
 * of `chunk` frames so intermediates stay Infinity-Cache resident. */
int ec_rn50_forward(const ec_rn50_t* h, const float* rgb_nhwc, int batch, void* workspace, size_t ws_bytes,
                    void* feat_bf16_nhwc, int chunk, ec_stream_t stream);
/* debugging / parity: copy of an intermediate stage of the LAST forward is not
 * kept; instead run only the first `n_ops` ops and return the op's output dims. */
int ec_rn50_num_ops(const ec_rn50_t* h);

#ifdef __cplusplus
}
#endif
#endif /* EC_AMD_H */
