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
typedef void* ec_event_t;    /* hipEvent_t */

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

/* Stream plumbing for callers that keep several launches in flight (no reference counterpart: the reference runs on one
 * stream).  The HIP runtime binds a stream to one of its few hardware queues at the stream's FIRST submission; two streams on
 * the same queue run one after the other.  ec_bind_streams gives each of `n` freshly created streams its first work (a
 * spin kernel of `spin_us`) while the others are busy, so that each gets a queue of its own; ec_stream_pair_overlap
 * measures a pair: *ratio ~ 1 = concurrent, ~ 2 = serialised.  Both block (device synchronise): set-up time only. */
int ec_bind_streams(ec_stream_t* streams, int n, int spin_us);
int ec_stream_pair_overlap(ec_stream_t a, ec_stream_t b, int spin_us, float* ratio);

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
/* ec_conv_bf16 writing a COLUMN BLOCK of a wider tensor: output row m starts at out + m * out_row_stride (elements;
 * >= Cout, multiple of 8; == Cout is ec_conv_bf16).  The trunk lays a stride-2 Bottleneck's pooled conv2 output
 * [M, planes] and its pooled block input [M, inplanes] (ec_avgpool2_bf16_ld) side by side, so that conv3 and the
 * downsample conv of [U] clip/model.py Bottleneck.forward (`out = relu(bn3(conv3(out)) + downsample(x))`) are ONE GEMM
 * over the concatenated K axis: the downsample output is never written or re-read. */
int ec_conv_bf16_ld(const void* in, const void* w, const float* bias, const void* res, void* out,
                    int B, int H, int W, int Cin, int Cout, int ksize, int pool, int act, int out_row_stride,
                    ec_stream_t stream);
/* Plain GEMM view of the same kernel: out[M,N] = act(A[M,K] W[N,K]^T + bias (+res)).
 * Replaces nn.Linear / nn.MultiheadAttention projections of [U] clip/model.py
 * ResidualAttentionBlock and AttentionPool2d.  K multiple of 8, N multiple of 32. */
int ec_gemm_bf16(const void* A, const void* W, const float* bias, const void* res, void* out,
                 int M, int N, int K, int act, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP image preprocessing on raw uint8 frames: Resize(n_px, BICUBIC) + CenterCrop(n_px), bit-exact with Pillow.
 * Replaces the PIL/torchvision half of `clip_preprocess(frame)` (primitive_probing/generate_data/
 * thor_image_features.py:108, :36-44; frames are 300x300: thor_frames.py:33-34).  ToTensor + Normalize are fused into
 * the stem (ec_rn50_forward_u8).
 *   ec_clip_resize_table_ints / ec_clip_resize_table: HOST ONLY -- torchvision's resize / center-crop geometry and
 *     Pillow's two 22-bit fixed-point coefficient tables (libImaging/Resample.c precompute_coeffs +
 *     normalize_coeffs_8bpc) for H x W frames; table[6] = the LDS rows the kernel needs (pass as table_max_rows).
 *     The caller copies the table to the device once per frame geometry.
 *   ec_clip_resize_crop_u8: frames uint8 [B, H, W, 3] -> uint8 [B, n_px, n_px, 3].
 * ---------------------------------------------------------------------- */
size_t ec_clip_resize_table_ints(int H, int W, int n_px);
int ec_clip_resize_table(int H, int W, int n_px, int* host_table, size_t n_ints);
int ec_clip_resize_crop_u8(const uint8_t* frames_u8, const int* table_dev, int table_max_rows, uint8_t* out_u8, int B,
                           int H, int W, int n_px, ec_stream_t stream);


/* out[M, N] (fp32) = act(A[M, K] (bf16) @ W^T + bias), W an fp32 [N, K] matrix handed over as the three bf16 planes
 * ec_split3_bf16 writes ([N][3][K]): the exact-fp32 product of stored bf16 features with fp32 weights -- the first
 * 1x1 conv of ResnetTensorGoalEncoder.resnet_compressor (allenact_plugins/robothor_plugin/.../resnet_tensor encoders;
 * SURVEY.md section 8 row a) -- on the 8-wave ping-pong kernel.  N % 128 == 0, K % 64 == 0. */
int ec_split3_bf16(const float* W, void* planes, long rows, int K, ec_stream_t stream);
int ec_gemm_bf16a_x3(const void* A_bf16, const void* W_planes, const float* bias, float* out, long M, int N, int K,
                     int act, ec_stream_t stream);

/* dW[128, NX] += dY^T X over M token rows: dY as three bf16 planes [M][3][128] (ec_split3_bf16 layout), X bf16 [M][NX]
 * (NX % 256 == 0) -- the weight gradient of the compressor's first 1x1 conv over the stored features.  `part` is scratch
 * of ec_dw_tn_x3_splits(M, NX) * 128 * NX floats (per-split partial tiles, folded deterministically). */
int ec_dw_tn_x3_splits(long M, int NX);
int ec_dw_tn_x3(const void* dY_planes, const void* X_bf16, float* part, float* dW, long M, int NX, ec_stream_t stream);

/* Stem conv1: 3x3 stride 2 pad 1 on the fp32 NHWC frame the RGB sensor hands
 * over ([U] ClipResNetPreprocessor.process: obs[rgb].permute(0,3,1,2)), folded
 * BN + ReLU, LDS-staged image tiles.  w f32 [3*3*3][Cout] (ky,kx,ci major),
 * out bf16 [B,H/2,W/2,Cout].  Cout in {32,48}. */
int ec_stem_conv1(const float* rgb_nhwc, const float* w, const float* bias, void* out,
                  int B, int H, int W, int Cout, ec_stream_t stream);

/* Fused pair of 1x1 convolutions across a Bottleneck boundary of layer1 (bandwidth-bound at 56x56):
 *   y = relu(a0 . w0^T + b0 [+ a1 . w1^T + b1] [+ res])   [M,256]  == relu(bn3(conv3(out)) + identity), the identity
 *                                                                    being `res` or the fused downsample conv (a1,w1,b1)
 *   z = relu(y . w2^T + b2)                               [M,N2]   == the next block's relu(bn1(conv1(x)))
 * ([U] openai/CLIP clip/model.py Bottleneck.forward; call site thor_image_features.py:109).
 * a0,a1 bf16 [M,64]; w0,w1 bf16 [256,64]; res,y bf16 [M,256]; w2 bf16 [N2,256]; z bf16 [M,N2].
 * K0 must be 64, N 256, N2 64 or 128 (64 only with a1), M a multiple of 32; else EC_ERR_SHAPE. */
int ec_conv1x1_pair_bf16(const void* a0, const void* w0, const float* b0, const void* a1, const void* w1,
                         const float* b1, const void* res, void* y, const void* w2, const float* b2, void* z,
                         long M, int K0, int N, int N2, ec_stream_t stream);

/* Same fused pair at the layer-1 -> layer-2 boundary (res given, N2 = 128), which ALSO emits y_pooled =
 * AvgPool2d(2)(y) bf16 [B, H/2, W/2, 256] -- the input of the next block's downsample path ([U] Bottleneck:
 * `downsample = Sequential(AvgPool2d(stride), Conv2d 1x1, BatchNorm2d)`) -- so that no separate pooling pass re-reads y.
 * Tiles are 4 x 8 pixel blocks: H % 4 == 0 and W % 8 == 0. */
int ec_conv1x1_pair_pool_bf16(const void* a0, const void* w0, const float* b0, const void* res, void* y,
                              void* y_pooled, const void* w2, const float* b2, void* z, int B, int H, int W, int K0,
                              int N, int N2, ec_stream_t stream);

/* Same, on the RAW uint8 HWC frame (thor_frames.py:33-34,96 writes uint8 frames): ToTensor (/255) and
 * Normalize(mean, std) of `clip_preprocess` (thor_image_features.py:108) are fused into the LDS staging.
 * h_mean3 / h_std3 are HOST pointers to 3 floats (CLIP_RGB_MEANS / CLIP_RGB_STDS). */
int ec_stem_conv1_u8(const uint8_t* rgb_u8_nhwc, const float* h_mean3, const float* h_std3, const float* w,
                     const float* bias, void* out, int B, int H, int W, int Cout, ec_stream_t stream);

/* Bottleneck-level fusion (conv_bneck.hip): conv2 (3x3) + bn2 + ReLU and conv3 (1x1) + bn3 + identity + ReLU of ONE
 * stride-1 Bottleneck in one launch, one workgroup per image, the image's 14 x 14 x C map resident in LDS -- replaces the
 * last two convs of [U] clip/model.py Bottleneck.forward for CLIP-RN50 layer3.1 .. layer3.5 (planes C = 256), reached
 * through `clip_model(clip_input)` (primitive_probing/generate_data/thor_image_features.py:109).
 * c1 bf16 [B,14,14,C] = relu(bn1(conv1(x))); b2 / b3 f32 (folded BatchNorm); x (identity) and y bf16 [B,14,14,4C].
 * The weights come in the kernel's streaming order: ec_bneck_pack_weights(w2 bf16 [C][3*3*C], w3 bf16 [4C][C]) ->
 * packed (ec_bneck_packed_elems(C) bf16 elements, conv2 then conv3; once per set of weights).
 * Bit-identical to ec_conv_bf16(3x3) followed by ec_conv_bf16(1x1, res = x).  EC_ERR_SHAPE unless H = W = 14, C = 256. */
size_t ec_bneck_packed_elems(int C);
int ec_bneck_pack_weights(const void* w2, const void* w3, void* packed, int C, ec_stream_t stream);
int ec_bneck_conv23_bf16(const void* c1, const void* packed, const float* b2, const float* b3,
                         const void* x, void* y, int B, int H, int W, int C, ec_stream_t stream);

/* The WHOLE stride-1 Bottleneck in one launch (conv1 + bn1 + ReLU in front of the two above; c1 never exists in HBM: the
 * block input x streams through a shared LDS ring): y = relu(conv3(relu(conv2(relu(conv1(x)))) + x).  packed =
 * ec_bneck3_pack_weights(w1 bf16 [C][4C], w2, w3) (ec_bneck3_packed_elems(C) elements).  Bit-identical to the three ec_conv_bf16
 * calls.  Same geometry restriction. */
size_t ec_bneck3_packed_elems(int C);
int ec_bneck3_pack_weights(const void* w1, const void* w2, const void* w3, void* packed, int C, ec_stream_t stream);
int ec_bneck_conv123_bf16(const void* x, const void* packed, const float* b1, const float* b2, const float* b3,
                          void* y, int B, int H, int W, int C, ec_stream_t stream);

/* relu(bn2(conv2(x))) of a late Bottleneck for SMALL launches (<= 64 frames: the per-GPU batches of strong scaling,
 * readme_files/baselines_habitat.md:63-73): one workgroup per (image, 32/64-channel slice), the image's map resident in
 * LDS, the eight waves split K and their partial tiles are folded through LDS in a fixed order (deterministic; equal to
 * ec_conv_bf16 up to fp32-accumulation rounding).  Geometries: (H = W = 14, C = 256), (H = W = 7, C = 512) and, with
 * pool = 1 (CLIP's anti-aliased stride: ReLU then AvgPool2d(2), out [B,7,7,C]; the 200-KB map resident in two channel
 * chunks), (H = W = 14, C = 512); else EC_ERR_SHAPE.  in / out bf16 [B,H,W,C]; packed = ec_conv3x3_img_pack(w bf16
 * [C][3*3*C]) (C * 9C elements). */
int ec_conv3x3_img_pack(const void* w, void* packed, int C, ec_stream_t stream);
int ec_conv3x3_img_bf16(const void* in, const void* packed, const float* bias, void* out, int B, int H, int W, int C,
                        int pool, ec_stream_t stream);
/* ... its pooled geometry with an output row stride (see ec_conv_bf16_ld); out_row_stride != C needs pool = 1. */
int ec_conv3x3_img_bf16_ld(const void* in, const void* packed, const float* bias, void* out, int B, int H, int W, int C,
                           int pool, int out_row_stride, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * torchvision ResNet-50 pieces: the ImageNet half of the feature scripts,
 * `resnet_model = Sequential(*list(models.resnet50(pretrained=True).children())[:-2])`
 * (primitive_probing/generate_data/thor_image_features.py:46-49,102-106; reachable_image_features.py:48-51,81-85).
 * ---------------------------------------------------------------------- */
/* Stride-2 convolution (1x1, or 3x3 with pad 1) + folded BatchNorm (+ residual) + activation: ResNet v1.5 strides
 * inside Bottleneck.conv2 and in the downsample conv, where CLIP's ModifiedResNet pools.  in bf16 [B,H,W,Cin] (H, W
 * even); w bf16 [Cout][k*k*Cin]; res / out bf16 [B,H/2,W/2,Cout].  Cin % 8 == 0, Cout % 64 == 0; act NONE or RELU. */
int ec_conv_bf16_s2(const void* in, const void* w, const float* bias, const void* res, void* out, int B, int H, int W,
                    int Cin, int Cout, int ksize, int act, ec_stream_t stream);
/* conv1 (7x7, stride 2, pad 3, 3 -> 64) + bn1 (folded) + ReLU + MaxPool2d(3, 2, 1) in one launch (the conv output never
 * touches HBM).  rgb: fp32 NHWC [B,H,W,3], ImageNet-normalised (u8 == 0), or raw uint8 NHWC with ToTensor +
 * Normalize(mean, std) of `resnet_preprocess` (thor_image_features.py:36-44) fused (u8 == 1; h_mean3 / h_std3 HOST
 * pointers to 3 floats).  w bf16 [64][176]: K = 7 rows (ky) of 24 slots, slot kx*3+ci < 21 real, the rest and the
 * last 8 zero; bias f32 [64]; out bf16 [B,H/4,W/4,64].  H, W multiples of 4. */
int ec_stem7_pool(const void* rgb, int u8, const float* h_mean3, const float* h_std3, const void* w, const float* bias,
                  void* out, int B, int H, int W, ec_stream_t stream);

/* AvgPool2d(2) on bf16 NHWC ([U] Bottleneck downsample "-1"). C multiple of 8. */
int ec_avgpool2_bf16(const void* in, void* out, int B, int H, int W, int C, ec_stream_t stream);
/* ... writing a column block of a wider tensor: pooled pixel q goes to out + q * out_row_stride (elements; see
 * ec_conv_bf16_ld).  `out` 16-B aligned. */
int ec_avgpool2_bf16_ld(const void* in, void* out, int B, int H, int W, int C, int out_row_stride, ec_stream_t stream);

/* bf16 NHWC [B,HW,C] -> fp32 NCHW [B,C,HW]: the `.float()` + layout the
 * reference exposes (thor_image_features.py:111; observation_space (2048,7,7)). */
int ec_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int HW, int C, ec_stream_t stream);
/* The inverse for a consumer that holds the reference's fp32 NCHW tensors ([U] ResnetTensorObjectNavActorCritic.forward's
 * observations): fp32 [B,C,HW] -> bf16 [B,HW,C]; *inexact (device int, zeroed by the caller) is OR-ed with 1 when some
 * value is not exactly representable in bf16 -- the caller then keeps its fp32 path.  C % 64 == 0. */
int ec_nchw_f32_to_nhwc_bf16(const float* in, void* out, int B, int HW, int C, int* inexact, ec_stream_t stream);

/* AdaptiveAvgPool2d(1)+Flatten on the fp32-cast features
 * (thor_image_features.py:63-66,113; preprocessor pool=True). bf16 [B,HW,C] -> f32 [B,C]. */
int ec_spatial_mean_bf16(const void* in, float* out, int B, int HW, int C, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP ModifiedResNet trunk (stem + layer1..4, attnpool detached):
 * `clip_model.visual` with `attnpool = Identity` (thor_image_features.py:59-67,109)
 * == [U] ClipResNetEmbedder.forward.
 * ---------------------------------------------------------------------- */
typedef struct ec_rn50 ec_rn50_t;

/* Weight order contract (execution order): stem conv1 (f32, [27][sc]) is
 * passed separately; `w_bf16` holds, concatenated, stem conv2, conv3, then per
 * block conv1, conv2, conv3, [downsample] as [Cout][k*k*Cin] bf16;
 * `bias` holds stem conv1..3 then the same per-block order, f32.
 * `width` is a multiple of 32 (64 = RN50, 96 = RN50x16).  The stem's w/2 channels are carried at
 * sc = roundup(w/2, 32) (RN50x16: 48 -> 64): stem conv1 is [27][sc], conv2 [sc][9*sc], conv3 [w][9*sc], with zero
 * weights / biases in the padding (exactly neutral after ReLU).
 * The handle borrows the device pointers (caller keeps them alive). */
int ec_rn50_create(ec_rn50_t** out, int width, const int* layers4, int input_resolution,
                   const float* stem_w_f32, const void* w_bf16, size_t n_w, const float* bias, size_t n_bias);
/* The torchvision ResNet trunk (width 64) behind the same handle type: every ec_rn50_* function below applies.
 * stem_w_bf16 = conv1 in ec_stem7_pool's layout; `w_bf16` / `bias` hold per block conv1, conv2, conv3, [downsample]
 * (bias: stem first) exactly as for ec_rn50_create.  ec_rn50_forward takes the ImageNet-normalised fp32 frame,
 * ec_rn50_forward_u8 the raw frame + IMAGENET mean / std.  Output bf16 NHWC [B,R/32,R/32,2048] == `imagenet_conv`
 * (thor_image_features.py:105,130) before its `.float()` / NCHW view. */
int ec_rn50tv_create(ec_rn50_t** out, const int* layers4, int input_resolution, const void* stem_w_bf16,
                     const void* w_bf16, size_t n_w, const float* bias, size_t n_bias);
void ec_rn50_destroy(ec_rn50_t* h);
size_t ec_rn50_workspace_bytes(const ec_rn50_t* h, int batch);
int ec_rn50_out_channels(const ec_rn50_t* h);
int ec_rn50_out_spatial(const ec_rn50_t* h);
/* rgb f32 NHWC [B,R,R,3] -> feat bf16 NHWC [B,R/32,R/32,32w].  `feat` may be a
 * slice of the rollout feature buffer.  chunk>0 runs the trunk in sub-batches
 * of `chunk` frames so intermediates stay Infinity-Cache resident. */
int ec_rn50_forward(const ec_rn50_t* h, const float* rgb_nhwc, int batch, void* workspace, size_t ws_bytes,
                    void* feat_bf16_nhwc, int chunk, ec_stream_t stream);
/* uint8 frames in (fused input pipeline; 4x fewer input bytes, no fp32 frame tensor). */
int ec_rn50_forward_u8(const ec_rn50_t* h, const uint8_t* rgb_u8_nhwc, const float* h_mean3, const float* h_std3,
                       int batch, void* workspace, size_t ws_bytes, void* feat_bf16_nhwc, int chunk,
                       ec_stream_t stream);
/* Number of ops in the handle's launch plan (from 128 frames on one kernel launch each: 38 for CLIP RN50 at 224 x 224;
 * smaller launches run the five fused layer-3 blocks as three launches each).  Tests and tools/ key on it. */
int ec_rn50_num_ops(const ec_rn50_t* h);
/* 64-bit hash of the handle's launch plan (op kinds, shapes, buffer routing) and the library version: identifies
 * what a profiler summary under profiles/ was measured on (bench.py rejects a stale one). */
uint64_t ec_rn50_plan_hash(const ec_rn50_t* h);
/* Tuning property of the handle (no reference counterpart): fewest 256-row output tiles for which this trunk's conv
 * launches take the 8-wave kernel (0 = library default, 150).  A caller that keeps two encoder launches in flight on
 * two streams (engine.Worker with 128-frame slices) sets 50 on both handles: each launch then occupies fewer, fully
 * used CUs.  Part of the plan hash; affects only this handle's forwards (it is not process state). */
int ec_rn50_set_conv8_min_tiles(ec_rn50_t* h, int n);


/* ------------------------------------------------------------------------
 * General fp32 GEMM on the exact-fp32 MFMA:  C (+)= epi(A B),
 * A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]  (NT / NN / TN by strides).
 * Replaces the cuBLAS calls behind nn.Conv2d(1x1) / nn.Linear / nn.GRU and their
 * autograd backward in the policy ([U] allenact ResnetTensorGoalEncoder,
 * RNNStateEncoder, LinearActorHead/CriticHead).
 * Epilogue order: + bias[n] + gbias[gidx[m/group]][n] -> relu -> * rowscale[m]
 * -> * (dmask[m,n] > 0) -> store / += / atomicAdd (splitk > 1; C pre-initialised).
 * ---------------------------------------------------------------------- */
enum {
    EC_GEMM_A_BF16 = 1,      /* A stored as bf16 (the frozen CLIP features) */
    EC_GEMM_B_BF16 = 2,
    EC_GEMM_RELU = 4,
    EC_GEMM_ACCUMULATE = 8,  /* C += ... */
    EC_GEMM_SPLIT_PARTS = 16,/* splitk > 1: slice z writes its partial sums to C + z * M * ldc (no atomics; the caller folds
                              * the parts in a fixed order); bias goes into part 0; no ReLU / mask / row scale */
    EC_GEMM_3PRODUCTS = 32   /* bf16x3 path: only the three leading bf16 products a0 b0 + a0 b1 + a1 b0 of an fp32 x fp32 product
                              * (relative error 2^-16 per product instead of 2^-24; half the MFMAs).  Meant for GRADIENT GEMMs
                              * (ec_policy_backward with EC_GEMM_BWD3=1); ignored on the exact-fp32 MFMA path */
};
int ec_gemm_f32(const void* A, const void* B, float* C, int M, int N, int K, long sam, long sak, long sbk, long sbn,
                int ldc, int flags, const float* bias, const float* gbias, const int* gidx, int group,
                const float* dmask, const float* rowscale, int splitk, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * GRU actor-critic policy == ActorCriticModel.forward as
 * ResnetTensorObjectNavActorCritic ([U] allenact; launched by
 * readme_files/baselines_robothor_objectnav.md:48-51) and its backward.
 * params/grads: ONE flat fp32 buffer each, tensors in AllenAct order
 * (embed_class, compressor.0.{w,b}, compressor.2.{w,b}, combiner.0.{w,b},
 * combiner.2.{w,b}, rnn.weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0,
 * actor.linear.{w,b}, critic.fc.{w,b}) at ec_policy_param_offset() (16-B aligned).
 * ---------------------------------------------------------------------- */
typedef struct {
    int in_channels;   /* 2048 */
    int spatial;       /* 7    */
    int hidden;        /* 512  */
    int goal_dims;     /* 32   */
    int num_goals;     /* 12   */
    int num_actions;   /* 6    */
    int compress_hid;  /* 128  */
    int compress_out;  /* 32   */
    int comb_hid;      /* 128  */
    int comb_out;      /* 32   */
    int fusion;        /* 0: ResnetTensorGoalEncoder (goal-embedding fusion of the RoboTHOR/Habitat configs);
                        * 1: zero-shot dual-encoder fusion (readme_files/zeroshot_objectnav.md:3-8; builder-defined,
                        *    parity-unpinned): x = feat/|feat| (*) goal_table[goal], feat = [T*N, 1, in_channels] CLIP
                        *    image embeddings, goal_table = frozen CLIP text embeddings (ec_policy_set_goal_table);
                        *    spatial must be 1, the compressor/combiner fields are ignored and those nine parameter
                        *    tensors have zero elements (the trainable part is GRU + heads) */
    int dual;          /* 1: RGB + depth, [U] ResnetDualTensorGoalEncoder (readme_files/baselines_habitat.md:75 "replace rgb
                        *    with rgbd"): two feature tensors, each with its own compressor / combiner (the goal embedding is
                        *    shared), rgb_x and depth_x concatenated along channels before nn.Flatten -> GRU input
                        *    2 * comb_out * spatial^2.  Eight more parameter tensors (the depth stream's, after the 17 of
                        *    the single encoder); use ec_policy_forward2 / ec_policy_backward2.  Not combinable with fusion */
} ec_policy_cfg;
typedef struct ec_policy ec_policy_t;

int ec_policy_create(ec_policy_t** out, const ec_policy_cfg* cfg);
/* fusion == 1 only: borrow the device goal table f32 [num_goals, in_channels] (e.g. ec_text_forward over the goal
 * prompts, L2-normalised); frozen -- it receives no gradient. */
int ec_policy_set_goal_table(ec_policy_t* h, const float* table);
void ec_policy_destroy(ec_policy_t* h);
int ec_policy_num_param_tensors(const ec_policy_t* h);
size_t ec_policy_flat_size(const ec_policy_t* h);                      /* floats, incl. alignment padding */
int ec_policy_param_offset(const ec_policy_t* h, int idx, size_t* off, size_t* numel);
size_t ec_policy_workspace_bytes(const ec_policy_t* h, int T, int N, int for_backward);
/* feat [T*N, S*S, C] NHWC (bf16 if feat_bf16 else f32); goal int64 [T*N]; h0 f32 [N,H];
 * masks f32 [T*N] (h is multiplied by masks[t] before step t: episode reset);
 * hv f32 [T*N, A+1] out: logits in cols 0..A-1, value in col A; h_final f32 [N,H] or NULL.
 * for_backward (the kernel plan depends on this flag alone, never on the size of the buffer handed in):
 *   EC_POLICY_LEARN (1)        the workspace (ec_policy_workspace_bytes(.., 1)) keeps every activation ec_policy_backward needs;
 *   EC_POLICY_INFER (0)        inference only (act step; a workspace of ec_policy_workspace_bytes(.., 0) suffices);
 *   EC_POLICY_INFER_REUSE (2)  inference, and the weight-derived tables an EC_POLICY_INFER call left in THIS workspace are
 *                              still valid (unchanged parameters: the later act steps of a rollout) -- they are not rebuilt. */
enum { EC_POLICY_INFER = 0, EC_POLICY_LEARN = 1, EC_POLICY_INFER_REUSE = 2 };
int ec_policy_forward(const ec_policy_t* h, const float* params, const void* feat, int feat_bf16,
                      const int64_t* goal, const float* h0, const float* masks, int T, int N,
                      void* workspace, size_t ws_bytes, int for_backward, float* hv, float* h_final, ec_stream_t stream);
/* dual == 1: feat = the RGB preprocessor's features, feat2 = the depth preprocessor's (same shape and dtype); with
 * dual == 0 feat2 is ignored (ec_policy_forward == ec_policy_forward2 with feat2 = NULL). */
int ec_policy_forward2(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                       const int64_t* goal, const float* h0, const float* masks, int T, int N,
                       void* workspace, size_t ws_bytes, int for_backward, float* hv, float* h_final, ec_stream_t stream);
/* The act step in ONE call ([U] allenact OnPolicyRLEngine.act: `actor_critic(...)`, then `distributions.sample()` and
 * `log_probs(actions)`): ec_policy_forward2 with T = 1 and EC_POLICY_INFER (reuse_tables = 0) / EC_POLICY_INFER_REUSE (1), whose
 * heads launch also samples -- actions int64 [N], logp f32 [N], values f32 [N] or NULL, keyed as in ec_sample_actions.  Bit-for-bit
 * the results of ec_policy_forward2 followed by ec_sample_actions; one launch less on the act step's serial chain.
 * EC_ERR_UNSUPPORTED for more than 7 actions (use the two calls). */
int ec_policy_act(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                  const int64_t* goal, const float* h0, const float* masks, int N, void* workspace, size_t ws_bytes,
                  int reuse_tables, float* hv, float* h_final, int64_t* actions, float* logp, float* values,
                  uint64_t seed, uint64_t step, int first_actor, ec_stream_t stream);
/* dhv f32 [T*N, A+1] = dLoss/dhv; dh_final [N,H] or NULL; grads += dLoss/dparams.
 * `workspace` must be the one the matching ec_policy_forward filled (for_backward size). */
int ec_policy_backward(const ec_policy_t* h, const float* params, const void* feat, int feat_bf16,
                       const float* masks, int T, int N, void* workspace, size_t ws_bytes, const float* dhv,
                       const float* dh_final, float* grads, ec_stream_t stream);
int ec_policy_backward2(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                        const float* masks, int T, int N, void* workspace, size_t ws_bytes, const float* dhv,
                        const float* dh_final, float* grads, ec_stream_t stream);
/* ec_policy_backward2 that also records `recurrent_grads_ready` (a hipEvent_t, or NULL) on `stream` at the point where the
 * gradients of rnn.weight_ih_l0 ... critic.fc.bias (one contiguous section of the flat bucket, 92 % of its bytes) are final
 * and only the goal encoder's remain to be written: the data-parallel caller starts that section's SUM all-reduce behind the
 * event, under the rest of the backward.  Replaces the `async_op=True` per-parameter all-reduces of [U] allenact
 * OnPolicyTrainer.backprop_step (§8e "can be fully overlapped with the GRU-backward tail"). */
int ec_policy_backward3(const ec_policy_t* h, const float* params, const void* feat, const void* feat2, int feat_bf16,
                        const float* masks, int T, int N, void* workspace, size_t ws_bytes, const float* dhv,
                        const float* dh_final, float* grads, ec_event_t recurrent_grads_ready, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * Rollout post-processing and the PPO update ([U] allenact onpolicy_sync:
 * storage.py RolloutStorage.compute_returns, losses/ppo.py PPO.loss,
 * distributions.py CategoricalDistr, engine.py backprop_step).
 * ---------------------------------------------------------------------- */
/* rewards [T,N]; values, masks [T+1,N]; returns [T+1,N]; adv [T,N]; norm_adv [T,N] or NULL
 * ((adv-mean)/(unbiased std + eps) over the T*N local steps); stats2: 2 doubles scratch. */
int ec_gae(const float* rewards, const float* values, const float* masks, float* returns, float* adv,
           float* norm_adv, double* stats2, int T, int N, float gamma, float tau, float eps, ec_stream_t stream);
/* hv [B,A+1]; actions int64 [B]; old_logp, old_values, returns, norm_adv f32 [B];
 * dhv [B,A+1] out = grad_scale * d(total)/d(hv); sums4 out (doubles): sum over B of
 * {action loss, value loss, -entropy, ratio}; total = (s0 + vcoef*s1 + ecoef*s2)/B. */
int ec_ppo_loss(const float* hv, const int64_t* actions, const float* old_logp, const float* old_values,
                const float* returns, const float* norm_adv, float* dhv, double* sums4, long B, int A,
                float clip, float vcoef, float ecoef, float grad_scale, ec_stream_t stream);
/* As ec_ppo_loss with the value clip given separately: value_clip >= 0 is PPO(use_clipped_value_loss=True) with
 * that clip (AllenAct uses the same, possibly decayed, clip_param for both); value_clip < 0 is
 * use_clipped_value_loss=False, value loss = 0.5 (returns - values)^2.  An action id outside [0, A) makes
 * sums4[0] NaN (the loss fails loudly instead of scoring log-prob 0). */
int ec_ppo_loss_ex(const float* hv, const int64_t* actions, const float* old_logp, const float* old_values,
                   const float* returns, const float* norm_adv, float* dhv, double* sums4, long B, int A,
                   float clip, float value_clip, float vcoef, float ecoef, float grad_scale, ec_stream_t stream);
/* actions ~ Categorical(logits = hv[:, :A]); logp = log_prob(actions); values = hv[:, A] (or NULL).
 * The counter-based uniform of row n is keyed by (seed, step, first_actor + n), so a batch may be sampled in
 * slices (e.g. one per HIP stream) with identical results. */
int ec_sample_actions(const float* hv, int64_t* actions, float* logp, float* values, int N, int A,
                      uint64_t seed, uint64_t step, int first_actor, ec_stream_t stream);
/* clip_grad_norm_(max_grad_norm) (<=0 disables) then Adam (1-based `step`) over n floats;
 * sumsq1: device scratch of ec_clip_adam_scratch_doubles() doubles; sumsq1[0] receives ||grads||^2 (the blocks' partial sums
 * and a ticket counter live behind it: the norm is folded in a fixed order, bit-identical run to run). */
int ec_clip_adam_scratch_doubles(void);
int ec_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, double* sumsq1,
                      long n, float max_grad_norm, float lr, float beta1, float beta2, float eps, int step,
                      ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP VisionTransformer embedder == [U] allenact ClipViTEmbedder.forward over
 * [U] openai/CLIP VisionTransformer: patch-embed, class +
 * positional embedding, ln_pre, then `layers_run` ResidualAttentionBlocks
 * (ClipViTPreprocessor runs all but the LAST block; no ln_post / proj).
 * w_bf16 (concatenated): conv1 as [D][P*P*3] with K ordered (ky,kx,c); then per block
 *   in_proj_weight [3D,D], out_proj.weight [D,D], mlp.c_fc.weight [4D,D], mlp.c_proj.weight [D,4D].
 * params_f32: class_embedding [D], positional_embedding [L,D], ln_pre.{w,b}; then per block
 *   ln_1.{w,b}, in_proj_bias [3D], out_proj.bias [D], ln_2.{w,b}, c_fc.bias [4D], c_proj.bias [D].
 * ---------------------------------------------------------------------- */
typedef struct ec_vit ec_vit_t;
int ec_vit_create(ec_vit_t** out, int width, int layers_run, int heads, int patch, int input_resolution,
                  const void* w_bf16, size_t n_w, const float* params_f32, size_t n_f);
void ec_vit_destroy(ec_vit_t* h);
int ec_vit_tokens(const ec_vit_t* h);                       /* L = (R/P)^2 + 1 */
int ec_vit_set_conv8_min_tiles(ec_vit_t* h, int n);          /* as ec_rn50_set_conv8_min_tiles */
uint64_t ec_vit_plan_hash(const ec_vit_t* h);                /* as ec_rn50_plan_hash */
size_t ec_vit_workspace_bytes(const ec_vit_t* h, int batch);
/* rgb f32 NHWC [B,R,R,3] -> tokens bf16 [B, L, D] */
int ec_vit_forward(const ec_vit_t* h, const float* rgb_nhwc, int batch, void* workspace, size_t ws_bytes,
                   void* tokens_bf16, ec_stream_t stream);
/* out[r, c] = (float) in[r*in_stride + c]  -- the `.float()` at the API edge (e.g. class_emb_only: in_stride = L*D) */
int ec_bf16_to_f32(const void* in, float* out, long rows, long row_len, long in_stride, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * [U] openai/CLIP AttentionPool2d.forward, the module the reference detaches and calls on the conv
 * features (primitive_probing/generate_data/thor_image_features.py:62,112): mean token + positional
 * embedding, q/k/v projections, 64-wide heads, softmax, c_proj; returns token 0 only, so only the
 * CLS query is computed.  feat bf16 NHWC [B,HW,C]; wkv = [k_proj.weight; v_proj.weight] ([2C,C]),
 * bkv likewise; out f32 [B,out_dim].
 * ---------------------------------------------------------------------- */
size_t ec_attnpool_workspace_bytes(int batch, int HW, int C);
int ec_attnpool_forward(const void* feat, int batch, int HW, int C, int heads, int out_dim, const float* pos,
                        const void* wq, const float* bq, const void* wkv, const float* bkv, const void* wc,
                        const float* bc, void* workspace, size_t ws_bytes, float* out, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP text tower == CLIP.encode_text ([U] openai/CLIP clip/model.py; the goal embedding of the zero-shot
 * ObjectNav variant, readme_files/zeroshot_objectnav.md:3-8): token + positional embedding, `layers` causal
 * ResidualAttentionBlocks, ln_final, features at the EOT token (arg-max id) @ text_projection.
 * w_bf16: per block in_proj [3D][D], out_proj [D][D], c_fc [4D][D], c_proj [D][4D]; then text_projection^T [E][D].
 * params_f32: token_embedding [vocab][D], positional_embedding [ctx][D], per block ln_1 w,b, in_proj_bias [3D],
 * out_proj.bias, ln_2 w,b, c_fc.bias [4D], c_proj.bias; then ln_final w,b.  tokens int32 [B][ctx] (the BPE tokenizer
 * is host string processing, not part of this path); out f32 [B][E].  width/heads must be 64, ctx <= 512.
 * ---------------------------------------------------------------------- */
typedef struct ec_text ec_text_t;
int ec_text_create(ec_text_t** out, int width, int layers, int heads, int context_length, int vocab_size,
                   int embed_dim, const void* w_bf16, size_t n_w, const float* params_f32, size_t n_f);
void ec_text_destroy(ec_text_t* h);
size_t ec_text_workspace_bytes(const ec_text_t* h, int batch);
int ec_text_forward(const ec_text_t* h, const int32_t* tokens, int batch, void* workspace, size_t ws_bytes,
                    float* out, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * Linear probe heads (BASELINE config 1) == LinearEncoder of
 * primitive_probing/train.py:14-113.  The Linear / Conv1x1 contraction is ec_gemm_f32; these are the
 * activation + loss + metric counts + analytic backward of compute_loss (train.py:56-92):
 *   EC_PROBE_SIGMOID_BCE         Sigmoid + F.binary_cross_entropy over [R,C]  (object_presence; object_localization
 *                                with R = B*9 rows from ec_probe_pool3)                     train.py:29,31,44-49,76
 *   EC_PROBE_SIGMOID_BCE_GATHER  Sigmoid, loss on column idx[r] only (reachability)         train.py:61-63,72,76
 *   EC_PROBE_SOFTMAX_CE2         Softmax(dim=1) then F.cross_entropy ON THE PROBABILITIES   train.py:35,78
 *                                (double softmax reproduced); labels > label_clamp are clamped (train.py:65)
 * labels int64: [R,C] (mode 0) or [R] (modes 1,2); idx int64 [R] (mode 1).
 * pred [R,C] (probabilities), dlogits [R,C] = d(mean loss)/d(logits), dbias [C] = column sums of dlogits:
 * each may be NULL.  out5 (doubles): {sum of per-element losses, tp, pred_pos, true_pos, correct}; the mean
 * loss is out5[0] / (R*C) (mode 0) or / R (modes 1,2); micro-F1 = 2 tp / (pred_pos + true_pos) (train.py:86),
 * accuracy = correct / count (train.py:88,90). */
enum { EC_PROBE_SIGMOID_BCE = 0, EC_PROBE_SIGMOID_BCE_GATHER = 1, EC_PROBE_SOFTMAX_CE2 = 2 };
int ec_probe_head(int mode, const float* logits, const int64_t* labels, const int64_t* idx, int R, int C,
                  int label_clamp, float* pred, float* dlogits, float* dbias, double* out5, ec_stream_t stream);
/* nn.AdaptiveAvgPool2d((3,3)) (train.py:45) of the cached fp32 NCHW conv features [B,C,H,W] ->
 * rows [B*9, C] so the Conv1x1 (train.py:46) is a row GEMM whose [B*9,52] output equals
 * y_pred.permute(0,2,1).flatten(1) (train.py:70). */
int ec_probe_pool3(const float* conv_nchw, float* rows, int B, int C, int H, int W, ec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EC_AMD_H */
