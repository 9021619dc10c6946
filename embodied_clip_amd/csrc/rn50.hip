// CLIP ModifiedResNet trunk executor (stem + layer1..4, attnpool detached).
//
// Replaces `clip_features = clip_model(clip_input)` with
// `clip_model.attnpool = nn.Identity()`
// (primitive_probing/generate_data/thor_image_features.py:59-67,109) ==
// [U] allenact_plugins/clip_plugin ClipResNetEmbedder.forward.
//
// The op list is derived from the architecture ([U] openai/CLIP clip/model.py
// ModifiedResNet.__init__/_make_layer, Bottleneck): it is a straight line of
// fused conv launches over five ping-pong NHWC bf16 buffers in the caller's
// workspace.  CLIP's anti-aliased stride (3x3 conv at full resolution, then
// AvgPool2d(2)) is fused into the 3x3 conv's epilogue; the residual add + ReLU
// is fused into conv3's epilogue.
#include <stdlib.h>

#include <new>
#include <vector>

#include "common.h"

// (common.h includes include/ec_amd.h, which declares the conv_bneck.hip entry points used below)

namespace {

enum OpKind { OP_STEM1, OP_CONV, OP_POOL, OP_PAIR, OP_BNECK, OP_STEM7 };

struct Op {
    OpKind kind;
    int src, dst, res;       // buffer ids; -1 = none; src -2 = rgb input; dst -3 = final output
    int H, W, Cin, Cout, ks, pool, act;
    size_t w_off, b_off;     // element offsets into w_bf16 / bias
    // OP_PAIR (fused layer-1 block boundary, conv_pair.hip): y = relu(src.w + [src1.w1] + [res]) -> dst;
    // z = relu(y.w2) -> dst2 (the next block's conv1 output, N2 channels)
    int src1 = -1, dst2 = -1, N2 = 0;
    int dst3 = -1;           // OP_PAIR at the layer-1 -> layer-2 boundary: AvgPool2d(2)(y) for the downsample path
    size_t w1_off = 0, b1_off = 0, w2_off = 0, b2_off = 0;
    long wc1_off = -1, bc1_off = -1;   // OP_BNECK with conv1 folded in (whole block in one launch): conv1's weight / bias offsets; its input is `res`
    int stride = 1;          // OP_CONV: 2 = torchvision's strided conv (ec_conv_bf16_s2); H, W are the INPUT dims
    long wimg_off = -1;      // offset (elements, into wbneck) of this 3x3 conv's streaming-order weights for the small-launch kernel
    int ldo = 0, ocol = 0;   // OP_CONV / OP_POOL writing a column block of a wider tensor: row stride (0 = dense) and first column (elements)
    long wcat_off = -1;      // OP_CONV over a concatenated K axis (conv3 | downsample conv of a stride-2 block): its [Cout][K1 + K2] weights in wbneck,
    long bcat_off = -1;      // ... its summed bias in bias_cat; w_off / b_off = conv3's, w1_off / b1_off = the downsample conv's, Cin = K1 + K2, N2 = K1    // OP_CONV (a block's conv3) that may also emit AvgPool2d(2) of its output for the NEXT block's K-concatenated GEMM (buffer pdst, row stride
    // pld, first column pcol): taken when the launch runs on conv1x1_regw_kernel<.., PL>; the OP_POOL that follows the next block's conv2 carries
    // pool_of = 1 and is skipped then
    int pdst = -1, pld = 0, pcol = 0;
    int pool_of = 0;
};

}  // namespace

struct ec_rn50 {
    int width, res, out_c, out_sp;
    int tv = 0;                   // 1: torchvision ResNet (7x7 stem + max-pool, stride inside conv2 / the downsample conv): ec_rn50tv_create
    std::vector<Op> ops;
    size_t max_elems_per_frame;   // largest activation (bf16 elements) per frame
    const float* stem_w;
    const uint16_t* w;
    const float* bias;
    size_t n_w, n_b;
    int conv8_min_tiles = 0;      // 0 = library default (ec_rn50_set_conv8_min_tiles)
    uint16_t* wbneck = nullptr;   // streaming-order weights of the fused bottleneck launches (ec_bneck_pack_weights), one block per OP_BNECK
    float* bias_cat = nullptr;    // summed biases of the K-concatenated convs (Op::bcat_off)
    ~ec_rn50() {
        if (wbneck) (void)hipFree(wbneck);
        if (bias_cat) (void)hipFree(bias_cat);
    }
};

namespace {
constexpr int NBUF = 5;
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

extern "C" int ec_stem7_pool(const void* rgb, int u8, const float* h_mean3, const float* h_std3, const void* w,
                             const float* bias, void* out, int B, int H, int W, ec_stream_t stream);
extern "C" int ec_conv_bf16_s2(const void* in, const void* w, const float* bias, const void* res, void* out, int B, int H, int W,
                               int Cin, int Cout, int ksize, int act, ec_stream_t stream);

namespace {
int rn50_build(ec_rn50_t** out, bool tv, int width, const int* layers4, int input_resolution, const void* stem_w,
               const void* w_bf16, size_t n_w, const float* bias, size_t n_bias);
}

extern "C" int ec_rn50_create(ec_rn50_t** out, int width, const int* layers4, int input_resolution,
                              const float* stem_w_f32, const void* w_bf16, size_t n_w, const float* bias,
                              size_t n_bias) {
    return rn50_build(out, false, width, layers4, input_resolution, stem_w_f32, w_bf16, n_w, bias, n_bias);
}

// torchvision ResNet (v1.5) trunk == Sequential(*list(resnet50.children())[:-2])
// (primitive_probing/generate_data/thor_image_features.py:46-49): the same executor and, for every stride-1 conv, the same
// launches as the CLIP trunk; the stem is ONE launch (7x7 s2 conv + bn + relu + 3x3 s2 max-pool, stem7.hip) and the first
// block of layers 2-4 strides inside its 3x3 conv and its 1x1 downsample conv (ec_conv_bf16_s2) instead of pooling.
extern "C" int ec_rn50tv_create(ec_rn50_t** out, const int* layers4, int input_resolution, const void* stem_w_bf16,
                                const void* w_bf16, size_t n_w, const float* bias, size_t n_bias) {
    return rn50_build(out, true, 64, layers4, input_resolution, stem_w_bf16, w_bf16, n_w, bias, n_bias);
}

namespace {
int rn50_build(ec_rn50_t** out, bool tv, int width, const int* layers4, int input_resolution, const void* stem_w_f32,
               const void* w_bf16, size_t n_w, const float* bias, size_t n_bias) {
    if (!out || !layers4 || !stem_w_f32 || !w_bf16 || !bias) return EC_ERR_ARG;
    if (width % 32 != 0 || width < 32 || input_resolution % 32 != 0) return EC_ERR_SHAPE;
    // stem channels: width/2, rounded up to the 32-channel granule of the conv kernels (RN50x16: 48 -> 64; the
    // packer zero-pads the weights, so the padded channels are exactly 0 after ReLU and contribute nothing)
    const int sc = (width / 2 + 31) / 32 * 32;
    ec_rn50* h = new (std::nothrow) ec_rn50();
    if (!h) return EC_ERR_ALLOC;
    h->width = width; h->res = input_resolution; h->tv = tv ? 1 : 0;
    h->stem_w = (const float*)stem_w_f32; h->w = (const uint16_t*)w_bf16; h->bias = bias;
    size_t wo = 0, bo = 0, mx = 0;
    auto track = [&](int H, int W, int C) { mx = std::max(mx, (size_t)H * W * C); };
    auto conv = [&](int src, int dst, int res, int H, int W, int Cin, int Cout, int ks, int pool, int act, int stride = 1) {
        Op o{OP_CONV, src, dst, res, H, W, Cin, Cout, ks, pool, act, wo, bo};
        o.stride = stride;
        wo += (size_t)Cout * ks * ks * Cin;
        bo += Cout;
        h->ops.push_back(o);
        track((pool || stride == 2) ? H / 2 : H, (pool || stride == 2) ? W / 2 : W, Cout);
    };
    int R = input_resolution / 2;
    // buffers: 0 = X (block input / output), 1,2 = temporaries, 3 = identity path, 4 = Y
    if (tv) {   // conv1 7x7 s2 + bn1 + relu + maxpool 3x3 s2 in one launch: frame -> buffer 0 at R/2 x R/2 x 64
        Op o{OP_STEM7, -2, 0, -1, input_resolution, input_resolution, 3, width, 7, 0, EC_ACT_RELU, 0, bo};
        bo += width;
        h->ops.push_back(o);
        R /= 2;
        track(R, R, width);
    } else {   // stem
        Op o{OP_STEM1, -2, 1, -1, input_resolution, input_resolution, 3, sc, 3, 0, EC_ACT_RELU, 0, bo};
        bo += sc;
        h->ops.push_back(o);
        track(R, R, sc);
        conv(1, 2, -1, R, R, sc, sc, 3, 0, EC_ACT_RELU);
        conv(2, 0, -1, R, R, sc, width, 3, 1, EC_ACT_RELU);   // + fused AvgPool2d(2)
        R /= 2;
    }
    // Layer-1 block boundaries (56x56, bandwidth-bound 1x1 convs) run as ONE fused launch per boundary:
    // conv3 (+ the block-0 downsample conv) + identity + ReLU, chained in registers into the next block's conv1.
    const bool fuse_env = ec_config().rn50_fuse != 0;
    const bool fuse_l1 = fuse_env && width == 64 && (R % 8) == 0;   // K = 64, N = 256; 32-pixel tiles divide R*R
    bool conv1_done = false;   // the previous boundary launch already produced this block's conv1 output in buffer 1
    int pooled_in = -1;        // ... and (layer-1 -> layer-2) the pooled block input for the downsample path, in this buffer
    int c1_buf = 1;            // buffer holding that conv1 output
    int inplanes = width, x = 0;
    for (int li = 0; li < 4; ++li) {
        const int planes = width << li;
        for (int b = 0; b < layers4[li]; ++b) {
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const bool ds = stride > 1 || inplanes != planes * 4;
            const int y = (x == 0) ? 4 : 0;
            const int Ro = R / stride;
            int c1 = 1;
            if (conv1_done) {   // weights are still laid out conv1, conv2, conv3, downsample: skip the slot
                wo += (size_t)planes * inplanes;
                bo += planes;
                conv1_done = false;
                c1 = c1_buf;
            } else {
                conv(x, 1, -1, R, R, inplanes, planes, 1, 0, EC_ACT_RELU);
            }
            if (tv && stride > 1) {   // torchvision: the 3x3 conv itself strides (no pool)
                conv(c1, 2, -1, R, R, planes, planes, 3, 0, EC_ACT_RELU, 2);
            } else
            conv(c1, 2, -1, R, R, planes, planes, 3, stride > 1 ? 1 : 0, EC_ACT_RELU);
            int idt = x;
            // weights are laid out conv1, conv2, conv3, downsample; the downsample conv
            // has to run BEFORE conv3 (conv3 consumes its output as the residual), so
            // reserve conv3's weight slot first.
            const size_t w_c3 = wo, b_c3 = bo;
            wo += (size_t)planes * 4 * planes; bo += planes * 4;
            // boundary fusion applies to every block of layer 1 that is followed by a 1x1 conv1 on its output
            const bool last_of_layer = (b + 1 == layers4[li]);
            const bool pair = fuse_l1 && li == 0 && stride == 1 && (!ds || (inplanes == planes && !last_of_layer)) &&
                              (!last_of_layer || li + 1 < 4);   // (downsample + a 128-wide conv1 exceeds the LDS)
            if (pair) {
                Op o{OP_PAIR, 2, y, ds ? -1 : x, R, R, planes, planes * 4, 1, 0, EC_ACT_RELU, w_c3, b_c3};
                if (ds) {   // block 0: the downsample conv (x -> 256) is folded in as a second K = 64 operand
                    o.src1 = x; o.w1_off = wo; o.b1_off = bo;
                    wo += (size_t)planes * 4 * inplanes; bo += planes * 4;
                }
                o.dst2 = 1;
                o.N2 = last_of_layer ? planes * 2 : planes;          // next conv1: 256 -> planes (same layer) | 2*planes
                if (!tv && last_of_layer && !ds && (R % 8) == 0) {   // the next block pools its input: emit it here
                    o.dst3 = 3;
                    pooled_in = 3;
                    track(R / 2, R / 2, planes * 4);
                }
                o.w2_off = wo; o.b2_off = bo;                        // == the next block's conv1 slot
                h->ops.push_back(o);
                track(R, R, planes * 4);
                conv1_done = true;
                c1_buf = 1;
                x = y;
                inplanes = planes * 4;
                continue;
            }
            // Stride-2 blocks of layers 3-4 (CLIP: AvgPool2d(2) after conv2 and in front of the downsample conv): the pooled conv2
            // output [M, planes] and the pooled block input [M, inplanes] are laid side by side in buffer 2, and
            //   relu(conv3(c2) + b3 + downsample(xp) + bd) == relu([c2 | xp] . [W3 | Wd]^T + (b3 + bd))
            // is ONE GEMM with K = planes + inplanes: the downsample output (M x 4 planes) is never written or re-read, one
            // launch less, and the sum is rounded to bf16 once instead of twice.  (layer 2's first block gets its pooled
            // input from the layer-1 boundary launch and chains conv3 into the next conv1: left as is.)
            if (ds && !tv && stride > 1 && li >= 2 && pooled_in < 0 && ec_config().rn50_dscat && (planes % 8) == 0) {
                const int Kc = planes + inplanes;
                // The concatenated operand lives in buffer 3 (free in layers 3-4), not in buffer 2: the conv3 that PRODUCES this block's
                // input reads its own conv2 output from buffer 2 and may write the pooled columns in the same launch (below).
                const int cat = 3;
                Op& c2op = h->ops.back();                       // conv2 (+ pool) -> buffer `cat`, columns [0, planes)
                c2op.ldo = Kc;
                c2op.dst = cat;
                Op pl{OP_POOL, x, cat, -1, R, R, inplanes, inplanes, 0, 0, 0, 0, 0};
                pl.ldo = Kc; pl.ocol = planes;                 // pooled block input -> buffer `cat`, columns [planes, Kc)
                // layer 2's last conv3 (128 -> 512 + identity @28x28; three ops back: behind it came this block's conv1 and conv2) can emit
                // the pooled copy itself (conv1x1_regw_kernel<.., PL>); whether it did is known per launch (rn50_run)
                if (ec_config().rn50_poolout && h->ops.size() >= 3) {
                    Op& pr = h->ops[h->ops.size() - 3];
                    if (pr.kind == OP_CONV && pr.ks == 1 && pr.dst == x && pr.res >= 0 && pr.Cin == 128 && pr.Cout == inplanes && inplanes == 512 &&
                        pr.H == R && pr.W == R && pr.act == EC_ACT_RELU && !pr.pool && !pr.ldo && pr.wcat_off < 0) {
                        pr.pdst = cat; pr.pld = Kc; pr.pcol = planes;
                        pl.pool_of = 1;
                    }
                }
                h->ops.push_back(pl);
                track(Ro, Ro, Kc);
                Op o{OP_CONV, cat, y, -1, Ro, Ro, Kc, planes * 4, 1, 0, EC_ACT_RELU, w_c3, b_c3};
                o.w1_off = wo; o.b1_off = bo;                   // the downsample conv's slot
                o.N2 = planes;
                o.wcat_off = 0;                                 // (assigned below, with the other packed weights)
                wo += (size_t)planes * 4 * inplanes; bo += planes * 4;
                h->ops.push_back(o);
                track(Ro, Ro, planes * 4);
                x = y;
                inplanes = planes * 4;
                R = Ro;
                continue;
            }
            if (ds) {
                int dsrc = x, ddst = 3;
                if (tv) {
                    // torchvision: downsample = Conv2d(1x1, stride) + BatchNorm2d straight on the block input
                } else
                if (stride > 1 && pooled_in >= 0) {   // pooled input came with the previous boundary launch (buffer 3);
                    dsrc = pooled_in;                 // buffer 1 (this block's conv1 output) is free after conv2
                    ddst = 1;
                    pooled_in = -1;
                } else if (stride > 1) {
                    // the pooled block input goes to the block's OUTPUT buffer y (free until conv3 writes it), not to the
                    // conv1 / conv2 temporaries
                    Op o{OP_POOL, x, y, -1, R, R, inplanes, inplanes, 0, 0, 0, 0, 0};
                    h->ops.push_back(o);
                    track(Ro, Ro, inplanes);
                    dsrc = y;
                }
                if (tv && stride > 1) {
                    conv(dsrc, ddst, -1, R, R, inplanes, planes * 4, 1, 0, EC_ACT_NONE, 2);
                } else
                conv(dsrc, ddst, -1, Ro, Ro, inplanes, planes * 4, 1, 0, EC_ACT_NONE);
                idt = ddst;
            }
            // Layer-2 block boundaries (28x28, 128 -> 512 -> 128): conv3 + identity + ReLU and the next block's conv1 in
            // one launch with the weights in registers (conv_pair.hip, layer-2 geometry).  Frame counts whose row count
            // is not a multiple of 32 run the two convs separately (see rn50_run).
            if (fuse_l1 && li == 1 && planes == 128 && !last_of_layer) {
                Op o{OP_PAIR, 2, y, idt, Ro, Ro, planes, planes * 4, 1, 0, EC_ACT_RELU, w_c3, b_c3};
                o.dst2 = (idt == 1) ? 3 : 1;      // block 0 with a pooled input keeps its identity in buffer 1
                o.N2 = planes;
                o.w2_off = wo; o.b2_off = bo;     // == the next block's conv1 slot
                h->ops.push_back(o);
                track(Ro, Ro, planes * 4);
                conv1_done = true;
                c1_buf = o.dst2;
            } else if (ec_config().rn50_bneck > 0 && !ds && stride == 1 && planes == 256 && Ro == 14 && !h->ops.empty() &&
                       h->ops.back().kind == OP_CONV && h->ops.back().ks == 3 && h->ops.back().dst == 2) {
                // Bottleneck-level fusion (conv_bneck.hip): conv2 + conv3 + identity + ReLU of layer3.1 .. layer3.5 in one
                // launch, one workgroup per image with the 14 x 14 x 256 map resident in LDS.  The conv2 op just planned
                // is folded in: src = conv1's output, res = the block input, w / b = conv2's, w1 / b1 = conv3's.
                // Launches below EC_RN50_BNECK frames run the two convs separately (rn50_run): a workgroup per image
                // only fills the chip from ~128 images on.
                const Op c2 = h->ops.back();
                h->ops.pop_back();
                Op o{OP_BNECK, c2.src, y, idt, Ro, Ro, planes, planes * 4, 3, 0, EC_ACT_RELU, c2.w_off, c2.b_off};
                o.w1_off = w_c3; o.b1_off = b_c3;
                // EC_RN50_BNECK3 (default 1): conv1 too -- the whole Bottleneck is ONE launch (bneck23_kernel<.., F1>): the op
                // just before conv2 is this block's conv1 (x -> buffer 1); it is folded in and its output never leaves the LDS
                if (ec_config().rn50_bneck3 && !h->ops.empty() && h->ops.back().kind == OP_CONV && h->ops.back().ks == 1 &&
                    h->ops.back().src == idt && h->ops.back().dst == c2.src && h->ops.back().Cin == planes * 4 && h->ops.back().Cout == planes) {
                    o.wc1_off = (long)h->ops.back().w_off; o.bc1_off = (long)h->ops.back().b_off;
                    h->ops.pop_back();
                }
                h->ops.push_back(o);
                track(Ro, Ro, planes * 4);
            } else {
                Op o{OP_CONV, 2, y, idt, Ro, Ro, planes, planes * 4, 1, 0, EC_ACT_RELU, w_c3, b_c3};
                h->ops.push_back(o);
                track(Ro, Ro, planes * 4);
            }
            x = y;
            inplanes = planes * 4;
            R = Ro;
        }
    }
    h->ops.back().dst = -3;
    h->out_c = inplanes; h->out_sp = R;
    h->max_elems_per_frame = mx;
    h->n_w = wo; h->n_b = bo;
    if (n_w != wo || n_bias != bo) { delete h; return EC_ERR_SHAPE; }
    {   // fused bottleneck launches: their conv2 + conv3 weights in streaming order, one packed block per op (w2_off = its offset)
        size_t tot = 0, btot = 0;
        for (Op& o : h->ops)
            if (o.wcat_off >= 0) {   // K-concatenated conv3 | downsample conv: [Cout][K1 + K2] weights, summed bias
                o.wcat_off = (long)tot; tot += (size_t)o.Cout * o.Cin;
                o.bcat_off = (long)btot; btot += (size_t)o.Cout;
            }
        if (btot && hipMalloc(&h->bias_cat, btot * sizeof(float)) != hipSuccess) { h->bias_cat = nullptr; delete h; return EC_ERR_LAUNCH; }
        for (Op& o : h->ops) {
            if (o.kind == OP_BNECK) { o.w2_off = tot; o.wimg_off = (long)tot; tot += ec_bneck3_packed_elems(o.Cin); }   // (packed conv2 comes first; room for conv1 too)
            // the un-pooled 3x3 convs of the 7x7 stage: streaming-order weights for the small-launch kernel (conv3x3_img_kernel)
            if (o.kind == OP_CONV && o.ks == 3 && o.Cin == 512 && o.Cout == 512 && ec_config().rn50_img3 &&
                ((!o.pool && o.H == 7 && o.W == 7) || (o.pool && o.H == 14 && o.W == 14))) {   // (layer4.0's conv2 + AvgPool2d: the chunked variant)
                o.wimg_off = (long)tot;
                tot += (size_t)o.Cout * 9 * o.Cin;
            }
        }
        if (tot) {
            if (hipMalloc(&h->wbneck, tot * sizeof(uint16_t)) != hipSuccess) { h->wbneck = nullptr; delete h; return EC_ERR_LAUNCH; }
            for (const Op& o : h->ops) {
                int rc = EC_OK;
                if (o.wcat_off >= 0) {
                    const int K1 = o.N2, K2 = o.Cin - o.N2;
                    uint16_t* wc = h->wbneck + o.wcat_off;
                    if (hipMemcpy2D(wc, (size_t)o.Cin * 2, h->w + o.w_off, (size_t)K1 * 2, (size_t)K1 * 2, (size_t)o.Cout,
                                    hipMemcpyDeviceToDevice) != hipSuccess ||
                        hipMemcpy2D(wc + K1, (size_t)o.Cin * 2, h->w + o.w1_off, (size_t)K2 * 2, (size_t)K2 * 2, (size_t)o.Cout,
                                    hipMemcpyDeviceToDevice) != hipSuccess) { delete h; return EC_ERR_LAUNCH; }
                    std::vector<float> b3((size_t)o.Cout), bd((size_t)o.Cout);
                    if (hipMemcpy(b3.data(), h->bias + o.b_off, (size_t)o.Cout * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(bd.data(), h->bias + o.b1_off, (size_t)o.Cout * 4, hipMemcpyDeviceToHost) != hipSuccess) { delete h; return EC_ERR_LAUNCH; }
                    for (int i = 0; i < o.Cout; ++i) b3[(size_t)i] += bd[(size_t)i];
                    if (hipMemcpy(h->bias_cat + o.bcat_off, b3.data(), (size_t)o.Cout * 4, hipMemcpyHostToDevice) != hipSuccess) { delete h; return EC_ERR_LAUNCH; }
                    continue;
                }
                if (o.kind == OP_BNECK)
                    rc = o.wc1_off >= 0 ? ec_bneck3_pack_weights(h->w + o.wc1_off, h->w + o.w_off, h->w + o.w1_off, h->wbneck + o.w2_off, o.Cin, nullptr)
                                        : ec_bneck_pack_weights(h->w + o.w_off, h->w + o.w1_off, h->wbneck + o.w2_off, o.Cin, nullptr);
                else if (o.wimg_off >= 0) rc = ec_conv3x3_img_pack(h->w + o.w_off, h->wbneck + o.wimg_off, o.Cin, nullptr);
                if (rc != EC_OK) { delete h; return EC_ERR_LAUNCH; }
            }
            (void)hipStreamSynchronize(nullptr);
        }
    }
    *out = h;
    return EC_OK;
}
}  // namespace

extern "C" void ec_rn50_destroy(ec_rn50_t* h) { delete h; }
extern "C" int ec_rn50_out_channels(const ec_rn50_t* h) { return h ? h->out_c : 0; }
extern "C" int ec_rn50_out_spatial(const ec_rn50_t* h) { return h ? h->out_sp : 0; }
extern "C" int ec_rn50_num_ops(const ec_rn50_t* h) { return h ? (int)h->ops.size() : 0; }

// FNV-1a over the launch plan (op kinds, shapes, buffer routing) and the library version: what a PMC traffic
// summary under profiles/ was measured on.  bench.py refuses a summary whose hash differs (stale evidence).
extern "C" uint64_t ec_rn50_plan_hash(const ec_rn50_t* h) {
    if (!h) return 0;
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](long v) {
        for (int i = 0; i < 8; ++i) { x ^= (uint64_t)((v >> (8 * i)) & 0xff); x *= 1099511628211ull; }
    };
    mix(ec_version()); mix((long)ec_config_hash());
    mix(h->width); mix(h->res); mix(h->conv8_min_tiles); mix(h->tv);
    for (const Op& o : h->ops) {
        mix(o.kind); mix(o.src); mix(o.dst); mix(o.res); mix(o.H); mix(o.W); mix(o.Cin); mix(o.Cout); mix(o.ks);
        mix(o.pool); mix(o.act); mix(o.stride); mix(o.src1); mix(o.dst2); mix(o.N2); mix(o.dst3); mix(o.wc1_off >= 0); mix(o.ldo); mix(o.ocol); mix(o.wcat_off >= 0); mix(o.pdst); mix(o.pld); mix(o.pcol); mix(o.pool_of);
    }
    return x;
}

extern "C" int ec_rn50_set_conv8_min_tiles(ec_rn50_t* h, int n) {
    if (!h) return EC_ERR_ARG;
    h->conv8_min_tiles = n > 0 ? n : 0;
    return EC_OK;
}

extern "C" size_t ec_rn50_workspace_bytes(const ec_rn50_t* h, int batch) {
    if (!h || batch <= 0) return 0;
    return NBUF * align_up(h->max_elems_per_frame * 2 * (size_t)batch, 256);
}

int ec_conv1x1_regw_pool(const void* a, const void* w, const float* bias, const void* res, void* y, void* y_pooled, int B, int H,
                         int W, int K, int N, int act, int ld_pooled, hipStream_t s);   // conv_pair.hip
extern "C" int ec_stem_conv1_u8(const uint8_t* rgb_u8, const float* mean3, const float* std3, const float* w,
                                const float* bias, void* out, int B, int H, int W, int Cout, ec_stream_t stream);

namespace {
int rn50_run(const ec_rn50_t* h, const void* rgb, bool u8, const float* mean3, const float* std3, int batch,
             void* workspace, size_t ws_bytes, void* feat, int chunk, ec_stream_t stream);
}

extern "C" int ec_rn50_forward(const ec_rn50_t* h, const float* rgb, int batch, void* workspace, size_t ws_bytes,
                               void* feat, int chunk, ec_stream_t stream) {
    return rn50_run(h, rgb, false, nullptr, nullptr, batch, workspace, ws_bytes, feat, chunk, stream);
}

extern "C" int ec_rn50_forward_u8(const ec_rn50_t* h, const uint8_t* rgb_u8, const float* h_mean3, const float* h_std3,
                                  int batch, void* workspace, size_t ws_bytes, void* feat, int chunk,
                                  ec_stream_t stream) {
    if (!h_mean3 || !h_std3) return EC_ERR_ARG;
    return rn50_run(h, rgb_u8, true, h_mean3, h_std3, batch, workspace, ws_bytes, feat, chunk, stream);
}

namespace {
int rn50_run(const ec_rn50_t* h, const void* rgb, bool u8, const float* mean3, const float* std3, int batch,
             void* workspace, size_t ws_bytes, void* feat, int chunk, ec_stream_t stream_main) {
    if (!h || !rgb || !workspace || !feat) return EC_ERR_ARG;
    if (batch <= 0) return EC_ERR_SHAPE;
    if (chunk <= 0 || chunk > batch) chunk = batch;
    {   // the conv kernels address activations through 32-bit buffer descriptors (< 2 GiB per tensor)
        const long maxc = ((1L << 31) - 1) / (long)(h->max_elems_per_frame * 2);
        if (chunk > maxc) chunk = (int)(maxc > 0 ? maxc : 1);
    }
    if (ws_bytes < ec_rn50_workspace_bytes(h, chunk)) return EC_ERR_WORKSPACE;
    const ec_min_tiles_scope mint_scope(h->conv8_min_tiles);   // this handle's dispatch threshold, for this call only
    const size_t bufsz = align_up(h->max_elems_per_frame * 2 * (size_t)chunk, 256);
    unsigned char* base = (unsigned char*)workspace;
    const size_t rgb_stride = (size_t)h->res * h->res * 3;
    const size_t out_stride = (size_t)h->out_sp * h->out_sp * h->out_c;
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = std::min(chunk, batch - b0);
        auto buf = [&](int id) -> void* {
            if (id == -3) return (uint16_t*)feat + (size_t)b0 * out_stride;
            return base + (size_t)id * bufsz;
        };
        bool pooled_emitted = false;   // the last conv3 wrote the pooled copy of its output: the OP_POOL marked pool_of is skipped
        for (const Op& o : h->ops) {
            int rc;
            const hipStream_t stream = (hipStream_t)stream_main;
            switch (o.kind) {
                case OP_STEM1:
                    if (u8)
                        rc = ec_stem_conv1_u8((const uint8_t*)rgb + (size_t)b0 * rgb_stride, mean3, std3, h->stem_w,
                                              h->bias + o.b_off, buf(o.dst), nb, o.H, o.W, o.Cout, stream);
                    else
                        rc = ec_stem_conv1((const float*)rgb + (size_t)b0 * rgb_stride, h->stem_w, h->bias + o.b_off,
                                           buf(o.dst), nb, o.H, o.W, o.Cout, stream);
                    break;
                case OP_STEM7:
                    rc = ec_stem7_pool(u8 ? (const void*)((const uint8_t*)rgb + (size_t)b0 * rgb_stride)
                                          : (const void*)((const float*)rgb + (size_t)b0 * rgb_stride),
                                       u8 ? 1 : 0, mean3, std3, h->stem_w, h->bias + o.b_off, buf(o.dst), nb, o.H, o.W, stream);
                    break;
                case OP_POOL:
                    if (o.pool_of && pooled_emitted) { pooled_emitted = false; rc = EC_OK; break; }
                    if (o.ldo) rc = ec_avgpool2_bf16_ld(buf(o.src), (uint16_t*)buf(o.dst) + o.ocol, nb, o.H, o.W, o.Cin, o.ldo, stream);
                    else
                    rc = ec_avgpool2_bf16(buf(o.src), buf(o.dst), nb, o.H, o.W, o.Cin, stream);
                    break;
                case OP_PAIR:
                    if (o.dst3 >= 0) {
                        // (the full-resolution block output is dead here: the next block reads dst2 and dst3 only)
                        rc = ec_conv1x1_pair_pool_bf16(buf(o.src), h->w + o.w_off, h->bias + o.b_off, buf(o.res), nullptr,
                                                       buf(o.dst3), h->w + o.w2_off, h->bias + o.b2_off, buf(o.dst2), nb, o.H,
                                                       o.W, o.Cin, o.Cout, o.N2, stream);
                        break;
                    }
                    rc = ec_conv1x1_pair_bf16(buf(o.src), h->w + o.w_off, h->bias + o.b_off,
                                              o.src1 >= 0 ? buf(o.src1) : nullptr, o.src1 >= 0 ? h->w + o.w1_off : nullptr,
                                              o.src1 >= 0 ? h->bias + o.b1_off : nullptr, o.res >= 0 ? buf(o.res) : nullptr,
                                              buf(o.dst), h->w + o.w2_off, h->bias + o.b2_off, buf(o.dst2),
                                              (long)nb * o.H * o.W, o.Cin, o.Cout, o.N2, stream);
                    if (rc == EC_ERR_SHAPE && o.src1 < 0) {   // e.g. an odd number of 28x28 frames: the two convs separately
                        rc = ec_conv_bf16(buf(o.src), h->w + o.w_off, h->bias + o.b_off, o.res >= 0 ? buf(o.res) : nullptr,
                                          buf(o.dst), nb, o.H, o.W, o.Cin, o.Cout, 1, 0, EC_ACT_RELU, stream);
                        if (rc == EC_OK)
                            rc = ec_conv_bf16(buf(o.dst), h->w + o.w2_off, h->bias + o.b2_off, nullptr, buf(o.dst2), nb, o.H,
                                              o.W, o.Cout, o.N2, 1, 0, EC_ACT_RELU, stream);
                    }
                    break;
                case OP_BNECK:
                    if (nb >= ec_config().rn50_bneck && o.wc1_off >= 0)
                        rc = ec_bneck_conv123_bf16(buf(o.res), h->wbneck + o.w2_off, h->bias + o.bc1_off, h->bias + o.b_off, h->bias + o.b1_off,
                                                   buf(o.dst), nb, o.H, o.W, o.Cin, stream);
                    else if (nb >= ec_config().rn50_bneck)
                        rc = ec_bneck_conv23_bf16(buf(o.src), h->wbneck + o.w2_off, h->bias + o.b_off, h->bias + o.b1_off,
                                                  buf(o.res), buf(o.dst), nb, o.H, o.W, o.Cin, stream);
                    else {   // small launches: the convs separately (buffer 1 = conv1's, buffer 2 = conv2's output, as in the unfused plan)
                        if (o.wc1_off >= 0) {
                            rc = ec_conv_bf16(buf(o.res), h->w + o.wc1_off, h->bias + o.bc1_off, nullptr, buf(o.src), nb, o.H, o.W,
                                                 o.Cout, o.Cin, 1, 0, EC_ACT_RELU, stream);
                            if (rc != EC_OK) return rc;
                        }
                        // ... conv2 on the image-resident K-split kernel while its (image, slice) workgroups fit one round
                        if (ec_config().rn50_img3 && nb * 8 <= 256)
                            rc = ec_conv3x3_img_bf16(buf(o.src), h->wbneck + o.wimg_off, h->bias + o.b_off, buf(2), nb, o.H, o.W, o.Cin, 0, stream);
                        else
                        rc = ec_conv_bf16(buf(o.src), h->w + o.w_off, h->bias + o.b_off, nullptr, buf(2), nb, o.H, o.W,
                                             o.Cin, o.Cin, 3, 0, EC_ACT_RELU, stream);
                        if (rc == EC_OK)
                            rc = ec_conv_bf16(buf(2), h->w + o.w1_off, h->bias + o.b1_off, buf(o.res), buf(o.dst), nb, o.H,
                                                 o.W, o.Cin, o.Cout, 1, 0, EC_ACT_RELU, stream);
                    }
                    break;
                default:
                    if (o.stride == 2) {
                        rc = ec_conv_bf16_s2(buf(o.src), h->w + o.w_off, h->bias + o.b_off, o.res >= 0 ? buf(o.res) : nullptr, buf(o.dst),
                                             nb, o.H, o.W, o.Cin, o.Cout, o.ks, o.act, stream);
                        break;
                    }
                    if (o.wimg_off >= 0 && o.kind == OP_CONV && nb <= (o.pool ? 16 : 64)) {   // 7x7x512 3x3 convs of small launches (two rounds of
                        // workgroups at most); layer4.0's pooled 14x14x512 conv2 (two channel chunks, 16 slices per image: one round of workgroups) up to 16 frames -- at 32 it ties with conv_igemm (47.6 vs 46.6 us)
                        rc = ec_conv3x3_img_bf16_ld(buf(o.src), h->wbneck + o.wimg_off, h->bias + o.b_off, (uint16_t*)buf(o.dst) + o.ocol, nb, o.H, o.W,
                                                    o.Cin, o.pool, o.ldo ? o.ldo : o.Cin, stream);
                        break;
                    }
                    if (o.wcat_off >= 0) {   // conv3 | downsample conv over the concatenated K axis
                        rc = ec_conv_bf16(buf(o.src), h->wbneck + o.wcat_off, h->bias_cat + o.bcat_off, nullptr, buf(o.dst), nb, o.H, o.W,
                                          o.Cin, o.Cout, 1, 0, o.act, stream);
                        break;
                    }
                    if (o.pdst >= 0) {
                        rc = ec_conv1x1_regw_pool(buf(o.src), h->w + o.w_off, h->bias + o.b_off, buf(o.res), buf(o.dst),
                                                  (uint16_t*)buf(o.pdst) + o.pcol, nb, o.H, o.W, o.Cin, o.Cout, o.act, o.pld, stream);
                        if (rc == EC_OK) { pooled_emitted = true; break; }
                        if (rc != EC_ERR_SHAPE) return rc;
                    }
                    rc = ec_conv_bf16_ld(buf(o.src), h->w + o.w_off, h->bias + o.b_off, o.res >= 0 ? buf(o.res) : nullptr,
                                         (uint16_t*)buf(o.dst) + o.ocol, nb, o.H, o.W, o.Cin, o.Cout, o.ks, o.pool, o.act,
                                         o.ldo ? o.ldo : o.Cout, stream);
            }
            if (rc != EC_OK) return rc;
        }
    }
    return EC_OK;
}
}  // namespace
