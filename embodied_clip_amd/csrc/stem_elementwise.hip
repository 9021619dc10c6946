// Stem conv1 (fp32 NHWC frame -> bf16 NHWC) and small bandwidth-bound helpers.
//
// Replaces: the first conv-bn-relu of [U] openai/CLIP ModifiedResNet.forward
// (called at primitive_probing/generate_data/thor_image_features.py:109) plus
// the `permute(0,3,1,2)` + dtype cast of [U] ClipResNetPreprocessor.process --
// the kernel reads the sensor's NHWC fp32 frame directly, so neither the
// permute nor a cast pass over the image ever touches HBM.
#include "common.h"

namespace {

typedef unsigned int u32x4_stem __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// stem conv1: 3x3, stride 2, pad 1, Cin = 3 -> COUT, folded BN + ReLU.
// K = 27 is too thin for MFMA and the op is bandwidth-bound (602 KB in,
// 803 KB out per 224^2 frame, 10.8 MMAC), so it is an fp32 VALU kernel:
// a workgroup stages the (2*16+1)^2 x 3 fp32 input patch of a 16x16 output
// tile in LDS with coalesced row loads, each lane then owns one output pixel
// and all COUT channels; weights are wave-uniform (scalar loads).
// ---------------------------------------------------------------------------
constexpr int ST = 16;                 // output tile edge
constexpr int SP = 2 * ST + 1;         // input patch edge (33)
constexpr int SROW = SP * 3;           // floats per patch row (99)
constexpr int SROW4 = 104;             // vectorised staging: rows start one float early (16-B aligned) and span 26 float4

// U8: the frame is the raw uint8 HWC image; ToTensor (/255) and Normalize(mean, std) of the reference's
// `clip_preprocess` (thor_image_features.py:108; constants CLIP_RGB_MEANS/STDS of the plugin) are applied while the
// patch is staged into LDS, so padding stays exactly zero in the NORMALISED domain as in the reference.
template <int COUT, bool U8>
__global__ __launch_bounds__(256) void stem_conv1_kernel(const void* __restrict__ rgb_, const float* __restrict__ w,
                                                         const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                         int H, int W, int Ho, int Wo, int tiles_x, int tiles_y,
                                                         float3 nscale, float3 nshift) {
    // Staging: the patch rows are fetched as 26 aligned 4-element vectors per row (the row origin 6*ox0 - 4 is a
    // multiple of 4 and W*3 is a multiple of 4 when W % 4 == 0, so a vector is entirely inside or outside the frame)
    // -- 4 wide loads per thread instead of 13 scalar ones with a div/mod each (the kernel is VALU-issue bound).
    __shared__ __attribute__((aligned(16))) float patch[SP * SROW4];
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * ST, ox0 = tx * ST;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const long img_off = (long)b * H * W * 3;
    const int c0 = ix0 * 3 - 1;            // first staged element of a row (multiple of 4)
    const bool vec = (W & 3) == 0;
    if (vec && ((long)gridDim.x / (tiles_x * tiles_y)) * H * W * 3 * (U8 ? 1 : 4) < (1L << 32) - 16) {
        // all of a thread's patch loads in flight before its first LDS store (the `for e` loop issued load -> wait ->
        // store four times in turn); they go through a buffer descriptor, so a vector outside the frame is an
        // out-of-range offset that reads as zeros instead of a branch around the load
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned nbytes = (unsigned)(((long)gridDim.x / (tiles_x * tiles_y)) * H * W * 3 * (U8 ? 1 : 4));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rgb_, 0, nbytes, 0x00020000);
        constexpr int NV = SP * 26, IT = (NV + 255) / 256;
        u32x4_stem raw[IT];
        int col_[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = min((int)threadIdx.x + 256 * i, NV - 1);
            const int r = e / 26, q = e - r * 26;
            const int iy = iy0 + r, col = c0 + 4 * q;
            col_[i] = col;
            const bool ok = iy >= 0 && iy < H && col >= 0 && col < W * 3;
            const unsigned off = ok ? (unsigned)((img_off + (long)iy * W * 3 + col) * (U8 ? 1 : 4)) : 0xFFFFFFF0u;
            if (U8) raw[i] = u32x4_stem{__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0), 0u, 0u, 0u};
            else raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = threadIdx.x + 256 * i;
            if (e < NV) {
                const int r = e / 26, q = e - r * 26;
                float4 v;
                if (U8) {
                    const int col = col_[i];
                    const bool ok = (iy0 + r) >= 0 && (iy0 + r) < H && col >= 0 && col < W * 3;
                    const unsigned u = raw[i][0];
                    const int ch0 = ((col % 3) + 3) % 3;     // channel of element 0
                    const float sc[3] = {nscale.x, nscale.y, nscale.z}, sh[3] = {nshift.x, nshift.y, nshift.z};
                    const int c1 = ch0 == 2 ? 0 : ch0 + 1, c2 = c1 == 2 ? 0 : c1 + 1;
                    // (padding stays exactly zero in the NORMALISED domain, as in the reference)
                    v.x = ok ? (float)(u & 0xffu) * sc[ch0] + sh[ch0] : 0.f;
                    v.y = ok ? (float)((u >> 8) & 0xffu) * sc[c1] + sh[c1] : 0.f;
                    v.z = ok ? (float)((u >> 16) & 0xffu) * sc[c2] + sh[c2] : 0.f;
                    v.w = ok ? (float)(u >> 24) * sc[ch0] + sh[ch0] : 0.f;
                } else {
                    v = __builtin_bit_cast(float4, raw[i]);
                }
                *reinterpret_cast<float4*>(patch + r * SROW4 + 4 * q) = v;
            }
        }
#endif
    } else if (vec) {
        for (int e = threadIdx.x; e < SP * 26; e += 256) {
            const int r = e / 26, q = e - r * 26;
            const int iy = iy0 + r, col = c0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && col >= 0 && col < W * 3) {
                const long idx = img_off + (long)iy * W * 3 + col;
                if (U8) {
                    const unsigned u = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(rgb_) + idx);
                    const int ch0 = col % 3;     // channel of element 0 (col >= 0 here)
                    const float sc[3] = {nscale.x, nscale.y, nscale.z}, sh[3] = {nshift.x, nshift.y, nshift.z};
                    const int c1 = ch0 == 2 ? 0 : ch0 + 1, c2 = c1 == 2 ? 0 : c1 + 1;
                    v.x = (float)(u & 0xffu) * sc[ch0] + sh[ch0];
                    v.y = (float)((u >> 8) & 0xffu) * sc[c1] + sh[c1];
                    v.z = (float)((u >> 16) & 0xffu) * sc[c2] + sh[c2];
                    v.w = (float)(u >> 24) * sc[ch0] + sh[ch0];
                } else {
                    v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rgb_) + idx);
                }
            }
            *reinterpret_cast<float4*>(patch + r * SROW4 + 4 * q) = v;
        }
    } else {
        for (int e = threadIdx.x; e < SP * SROW; e += 256) {
            const int r = e / SROW, c = e - r * SROW;
            const int iy = iy0 + r;
            const int ixc = ix0 * 3 + c;   // element index within the image row
            float v = 0.f;
            if (iy >= 0 && iy < H && ixc >= 0 && ixc < W * 3) {
                const long idx = img_off + (long)iy * W * 3 + ixc;
                if (U8) {
                    const int ch = ixc % 3;
                    const float sc = ch == 0 ? nscale.x : (ch == 1 ? nscale.y : nscale.z);
                    const float sh = ch == 0 ? nshift.x : (ch == 1 ? nshift.y : nshift.z);
                    v = (float)reinterpret_cast<const unsigned char*>(rgb_)[idx] * sc + sh;
                } else {
                    v = reinterpret_cast<const float*>(rgb_)[idx];
                }
            }
            patch[r * SROW4 + c + 1] = v;  // same image as the vector path: element c sits at column c + 1
        }
    }
    __syncthreads();
    if constexpr (COUT % 32 == 0) {
        // ---- MFMA path (round 3): the 3x3x3 window is a K = 27 (-> 32) contraction on the bf16 MFMA --------------------
        // The fp32 VALU version issued 432 packed FMAs per lane for a kernel whose floor is its 359 MB of traffic
        // (137 us at 256 frames, 2.6 TB/s, "VALU 8.9 % active, waiting 60 %").  Here a wave owns two 32-pixel blocks of the
        // 16 x 16 tile; lane (px, h) gathers its 8 + 8 window values (k = 16 s + 8 h + e  <->  tap k / 3, channel k % 3)
        // from the fp32 LDS patch, rounds them to bf16 (the precision every later layer runs at; the reference feeds fp16
        // frames to CLIP) and two MFMAs per 32-channel block do the rest.  Swapped operands (D[channel][pixel]) as in the
        // other conv kernels: a lane ends up with one pixel and 4 consecutive channels per 4 accumulator registers.
        constexpr int FN = COUT / 32;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, px = lane & 31, h = lane >> 5;
        // patch offset (floats) of window element k relative to the pixel's window origin; k >= 27 -> a zero word
        int koff[2][8];
        bool kval[2][8];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * sidx + 8 * h + e;
                kval[sidx][e] = k < 27;
                koff[sidx][e] = (k / 9) * SROW4 + (k % 9);
            }
        // weight fragments: W[n][k] = w[k * COUT + n] as bf16, K padded with zeros
        s16x8_t wf[FN][2];
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * sidx + 8 * h + e;
                    v[e] = (k < 27) ? w[k * COUT + 32 * j + px] : 0.f;
                }
                const u32x4_stem pk = {ec_pack2(v[0], v[1]), ec_pack2(v[2], v[3]), ec_pack2(v[4], v[5]), ec_pack2(v[6], v[7])};
                wf[j][sidx] = __builtin_bit_cast(s16x8_t, pk);
            }
        s16x8_t af[2][2];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int p_ = (wave * 2 + b2) * 32 + px;              // pixel of the 16 x 16 tile this lane gathers for
            const int ly = p_ >> 4, lx = p_ & 15;
            const float* win = patch + (2 * ly) * SROW4 + (2 * lx) * 3 + 1;
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = kval[sidx][e] ? win[koff[sidx][e]] : 0.f;
                const u32x4_stem pk = {ec_pack2(v[0], v[1]), ec_pack2(v[2], v[3]), ec_pack2(v[4], v[5]), ec_pack2(v[6], v[7])};
                af[b2][sidx] = __builtin_bit_cast(s16x8_t, pk);
            }
        }
        __syncthreads();                                           // every wave has gathered: the patch becomes 4 staging images
        // output through a wave-private LDS image (32 pixels x 64 B, 80-B pitch): the tile's pixel rows are contiguous in
        // memory, so a wave then stores 16 pixels x 64 B = 1 KiB per instruction instead of 32 scattered 16-B pieces
        unsigned char* stg = reinterpret_cast<unsigned char*>(patch) + wave * (32 * 80);
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[j][0]), __builtin_bit_cast(bf16x8_t, af[b2][0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[j][1]), __builtin_bit_cast(bf16x8_t, af[b2][1]), acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * j + 8 * g + 4 * h);
                    uint2 o;
                    o.x = ec_pack2(fmaxf(acc[4 * g + 0] + bv.x, 0.f), fmaxf(acc[4 * g + 1] + bv.y, 0.f));
                    o.y = ec_pack2(fmaxf(acc[4 * g + 2] + bv.z, 0.f), fmaxf(acc[4 * g + 3] + bv.w, 0.f));
                    *reinterpret_cast<uint2*>(stg + px * 80 + (8 * g + 4 * h) * 2) = o;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {                      // (same wave wrote it: LDS is in order per wave)
                    const int c = lane + 64 * i, q = c >> 2, part = c & 3;
                    const int p2 = (wave * 2 + b2) * 32 + q;
                    const int oy = oy0 + (p2 >> 4), ox = ox0 + (p2 & 15);
                    const u32x4_stem v = *reinterpret_cast<const u32x4_stem*>(stg + q * 80 + part * 16);
                    if (oy < Ho && ox < Wo)
                        *reinterpret_cast<u32x4_stem*>(out + ((long)(b * Ho + oy) * Wo + ox) * COUT + 32 * j + part * 8) = v;
                }
            }
        }
        return;
    }
    const int ly = threadIdx.x / ST, lx = threadIdx.x % ST;
    const int oy = oy0 + ly, ox = ox0 + lx;
    // packed fp32 FMAs (v_pk_fma_f32: two channels per instruction); weights are wave-uniform (scalar loads)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[COUT / 2];
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) acc2[c] = *reinterpret_cast<const f32x2*>(bias + 2 * c);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = patch[(2 * ly + ky) * SROW4 + (2 * lx + kx) * 3 + ci + 1];
                const f32x2 v2 = {v, v};
                const f32x2* wr = reinterpret_cast<const f32x2*>(w + ((ky * 3 + kx) * 3 + ci) * COUT);
#pragma unroll
                for (int c = 0; c < COUT / 2; ++c) acc2[c] = __builtin_elementwise_fma(v2, wr[c], acc2[c]);
            }
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) { acc[2 * c] = acc2[c][0]; acc[2 * c + 1] = acc2[c][1]; }
    if (oy < Ho && ox < Wo) {
        uint4* dst = reinterpret_cast<uint4*>(out + ((long)(b * Ho + oy) * Wo + ox) * COUT);
#pragma unroll
        for (int c = 0; c < COUT; c += 8) {
            uint4 v;
            v.x = ec_pack2(fmaxf(acc[c + 0], 0.f), fmaxf(acc[c + 1], 0.f));
            v.y = ec_pack2(fmaxf(acc[c + 2], 0.f), fmaxf(acc[c + 3], 0.f));
            v.z = ec_pack2(fmaxf(acc[c + 4], 0.f), fmaxf(acc[c + 5], 0.f));
            v.w = ec_pack2(fmaxf(acc[c + 6], 0.f), fmaxf(acc[c + 7], 0.f));
            dst[c / 8] = v;
        }
    }
}

// AvgPool2d(2) on bf16 NHWC, 8 channels (16 B) per lane.
__global__ __launch_bounds__(256) void avgpool2_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                      int H, int W, int C8, long total, int ld8) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int c = (int)(t % C8); t /= C8;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho);
        const long b = t / Ho;
        const uint4* p = reinterpret_cast<const uint4*>(in) + ((b * H + 2 * yo) * W + 2 * xo) * C8 + c;
        const uint4 a = p[0], bq = p[C8], cq = p[(long)W * C8], d = p[(long)W * C8 + C8];
        uint4 o;
        o.x = ec_pack2(0.25f * (ec_lo(a.x) + ec_lo(bq.x) + ec_lo(cq.x) + ec_lo(d.x)),
                       0.25f * (ec_hi(a.x) + ec_hi(bq.x) + ec_hi(cq.x) + ec_hi(d.x)));
        o.y = ec_pack2(0.25f * (ec_lo(a.y) + ec_lo(bq.y) + ec_lo(cq.y) + ec_lo(d.y)),
                       0.25f * (ec_hi(a.y) + ec_hi(bq.y) + ec_hi(cq.y) + ec_hi(d.y)));
        o.z = ec_pack2(0.25f * (ec_lo(a.z) + ec_lo(bq.z) + ec_lo(cq.z) + ec_lo(d.z)),
                       0.25f * (ec_hi(a.z) + ec_hi(bq.z) + ec_hi(cq.z) + ec_hi(d.z)));
        o.w = ec_pack2(0.25f * (ec_lo(a.w) + ec_lo(bq.w) + ec_lo(cq.w) + ec_lo(d.w)),
                       0.25f * (ec_hi(a.w) + ec_hi(bq.w) + ec_hi(cq.w) + ec_hi(d.w)));
        reinterpret_cast<uint4*>(out)[(i / C8) * ld8 + c] = o;      // (ld8 == C8: the dense tensor, index i)
    }
}

// bf16 [B, HW, C] -> fp32 [B, C, HW]; one workgroup per (frame, 64-channel slab):
// coalesced bf16 reads along C, LDS transpose, the slab's [64][HW] floats are
// one contiguous run of the NCHW output.
constexpr int TC = 64;
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const uint16_t* __restrict__ in, float* __restrict__ out,
                                                          int HW, int C) {
    extern __shared__ float tile[];   // [HW][TC+1]
    const int slabs = C / TC;
    const int b = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * TC;
    const uint16_t* src = in + (long)b * HW * C + c0;
    for (int e = threadIdx.x; e < HW * TC; e += 256) {
        const int hw = e / TC, c = e % TC;
        tile[hw * (TC + 1) + c] = ec_bf2f(src[(long)hw * C + c]);
    }
    __syncthreads();
    float* dst = out + ((long)b * C + c0) * HW;
    for (int e = threadIdx.x; e < HW * TC; e += 256) {
        const int c = e / HW, hw = e % HW;
        dst[e] = tile[hw * (TC + 1) + c];
    }
}

// The inverse, for the drop-in policy module's learn pass: fp32 [B, C, HW] (the reference's tensor contract) -> bf16 [B, HW, C]
// rows, the operand format of the fast compressor kernels -- used ONLY when every value is exactly a bf16 (what
// ClipResNetPreprocessor.process returns: the trunk's bf16 output widened), which this kernel checks while it converts:
// *inexact is OR-ed with 1 as soon as one element does not survive the round trip (the caller then keeps the fp32 path).
__global__ __launch_bounds__(256) void nchw_to_nhwc_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out,
                                                               int HW, int C, int* __restrict__ inexact) {
    extern __shared__ float tile[];   // [HW][TC+1]
    const int slabs = C / TC;
    const int b = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * TC;
    const float* src = in + ((long)b * C + c0) * HW;          // the slab's [TC][HW] floats are contiguous
    int bad = 0;
    for (int e = threadIdx.x; e < HW * TC; e += 256) {
        const int c = e / HW, hw = e % HW;
        const float v = src[e];
        bad |= (ec_bf2f(ec_f2bf(v)) != v) && (v == v);         // (NaNs are left to the consumer)
        tile[hw * (TC + 1) + c] = v;
    }
    __syncthreads();
    uint16_t* dst = out + (long)b * HW * C + c0;
    for (int e = threadIdx.x; e < HW * TC; e += 256) {
        const int hw = e / TC, c = e % TC;
        dst[(long)hw * C + c] = ec_f2bf(tile[hw * (TC + 1) + c]);
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(inexact, 1);
}

extern "C" int ec_nchw_f32_to_nhwc_bf16(const float* in, void* out, int B, int HW, int C, int* inexact, ec_stream_t stream) {
    if (!in || !out || !inexact) return EC_ERR_ARG;
    if (B <= 0 || HW <= 0 || C % TC != 0) return EC_ERR_SHAPE;
    const size_t lds = (size_t)HW * (TC + 1) * sizeof(float);
    if (lds > 64 * 1024) return EC_ERR_SHAPE;
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3((unsigned)(B * (C / TC))), dim3(256), lds, (hipStream_t)stream, in,
                       (uint16_t*)out, HW, C, inexact);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

// mean over HW of bf16 [B, HW, C] -> fp32 [B, C]
__global__ __launch_bounds__(256) void spatial_mean_kernel(const uint16_t* __restrict__ in, float* __restrict__ out,
                                                          int HW, int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / C;
    const int c = (int)(i % C);
    const uint16_t* p = in + b * HW * C + c;
    float s = 0.f;
    for (int hw = 0; hw < HW; ++hw) s += ec_bf2f(p[(long)hw * C]);
    out[i] = s / (float)HW;
}

}  // namespace

extern "C" int ec_stem_conv1(const float* rgb, const float* w, const float* bias, void* out, int B, int H, int W,
                             int Cout, ec_stream_t stream) {
    if (!rgb || !w || !bias || !out) return EC_ERR_ARG;
    if (B <= 0 || H < 2 || W < 2) return EC_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int tx = (Wo + ST - 1) / ST, ty = (Ho + ST - 1) / ST;
    dim3 grid((unsigned)(B * tx * ty));
    hipStream_t s = (hipStream_t)stream;
    const float3 z = make_float3(0.f, 0.f, 0.f);
    if (Cout == 32)
        hipLaunchKernelGGL((stem_conv1_kernel<32, false>), grid, dim3(256), 0, s, (const void*)rgb, w, bias, (uint16_t*)out, H, W,
                           Ho, Wo, tx, ty, z, z);
    else if (Cout == 48)
        hipLaunchKernelGGL((stem_conv1_kernel<48, false>), grid, dim3(256), 0, s, (const void*)rgb, w, bias, (uint16_t*)out, H, W,
                           Ho, Wo, tx, ty, z, z);
    else if (Cout == 64)   // RN50x16: 48 real channels zero-padded to the 32-channel granule of the conv kernels
        hipLaunchKernelGGL((stem_conv1_kernel<64, false>), grid, dim3(256), 0, s, (const void*)rgb, w, bias, (uint16_t*)out, H, W,
                           Ho, Wo, tx, ty, z, z);
    else
        return EC_ERR_SHAPE;
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_stem_conv1_u8(const uint8_t* rgb_u8, const float* mean3, const float* std3, const float* w,
                                const float* bias, void* out, int B, int H, int W, int Cout, ec_stream_t stream) {
    if (!rgb_u8 || !mean3 || !std3 || !w || !bias || !out) return EC_ERR_ARG;   // mean3/std3 are HOST pointers
    if (B <= 0 || H < 2 || W < 2) return EC_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int tx = (Wo + ST - 1) / ST, ty = (Ho + ST - 1) / ST;
    dim3 grid((unsigned)(B * tx * ty));
    hipStream_t s = (hipStream_t)stream;
    // (u8/255 - mean)/std == u8 * (1/(255 std)) - mean/std
    const float3 sc = make_float3(1.f / (255.f * std3[0]), 1.f / (255.f * std3[1]), 1.f / (255.f * std3[2]));
    const float3 sh = make_float3(-mean3[0] / std3[0], -mean3[1] / std3[1], -mean3[2] / std3[2]);
    if (Cout == 32)
        hipLaunchKernelGGL((stem_conv1_kernel<32, true>), grid, dim3(256), 0, s, (const void*)rgb_u8, w, bias, (uint16_t*)out, H,
                           W, Ho, Wo, tx, ty, sc, sh);
    else if (Cout == 48)
        hipLaunchKernelGGL((stem_conv1_kernel<48, true>), grid, dim3(256), 0, s, (const void*)rgb_u8, w, bias, (uint16_t*)out, H,
                           W, Ho, Wo, tx, ty, sc, sh);
    else if (Cout == 64)
        hipLaunchKernelGGL((stem_conv1_kernel<64, true>), grid, dim3(256), 0, s, (const void*)rgb_u8, w, bias, (uint16_t*)out, H,
                           W, Ho, Wo, tx, ty, sc, sh);
    else
        return EC_ERR_SHAPE;
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_avgpool2_bf16(const void* in, void* out, int B, int H, int W, int C, ec_stream_t stream) {
    return ec_avgpool2_bf16_ld(in, out, B, H, W, C, C, stream);
}

// ... writing a column block of a wider tensor: pooled pixel q goes to out + q * out_row_stride (elements); see ec_conv_bf16_ld
extern "C" int ec_avgpool2_bf16_ld(const void* in, void* out, int B, int H, int W, int C, int out_row_stride, ec_stream_t stream) {
    if (!in || !out) return EC_ERR_ARG;
    if (B <= 0 || (H & 1) || (W & 1) || C % 8 != 0) return EC_ERR_SHAPE;
    if (out_row_stride < C || (out_row_stride & 7) || ((size_t)out & 15)) return EC_ERR_SHAPE;
    const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)in, (uint16_t*)out, H, W, C / 8, total, out_row_stride / 8);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int HW, int C, ec_stream_t stream) {
    if (!in || !out) return EC_ERR_ARG;
    if (B <= 0 || HW <= 0 || C % TC != 0) return EC_ERR_SHAPE;
    const size_t lds = (size_t)HW * (TC + 1) * sizeof(float);
    if (lds > 64 * 1024) return EC_ERR_SHAPE;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)(B * (C / TC))), dim3(256), lds, (hipStream_t)stream,
                       (const uint16_t*)in, out, HW, C);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

extern "C" int ec_spatial_mean_bf16(const void* in, float* out, int B, int HW, int C, ec_stream_t stream) {
    if (!in || !out) return EC_ERR_ARG;
    if (B <= 0 || HW <= 0 || C <= 0) return EC_ERR_SHAPE;
    const long total = (long)B * C;
    hipLaunchKernelGGL(spatial_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)in, out, HW, C, total);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
