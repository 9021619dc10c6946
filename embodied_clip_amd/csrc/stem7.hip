// Stem of torchvision's ResNet-50: conv1 (7x7, stride 2, pad 3, 3 -> 64) + bn1 (folded) + ReLU + MaxPool2d(3, 2, 1)
// in ONE launch -- the first four children of `Sequential(*list(resnet50.children())[:-2])`
// (primitive_probing/generate_data/thor_image_features.py:46-49, reachable_image_features.py:48-51), the network behind
// the `imagenet_conv` / `imagenet_avgpool` features (:102-106, :130-131).
//
// The 112 x 112 x 64 conv output (1.6 MB per frame in bf16) never touches HBM: a workgroup (4 waves, two workgroups per CU so
// that one tile's staging / MFMA / pooling phases run beside the other's) owns a 4 x 14 tile of the POOLED 56 x 56 map, computes
// the 9 x 29 conv pixels its 3 x 3 / stride-2 windows cover (1.17 x recompute at the tile seams) on the bf16 MFMA, keeps them
// in LDS and pools from there.  Per frame: 602 KB of fp32 image in (150 KB as uint8), 401 KB out.
//
//   * K = 7 x 7 x 3 = 147 is laid out as 7 rows of 24 (ky; kx * 3 + ci < 21 real, 3 zero weights) + 8 zeros = 176 =
//     11 k-steps of v_mfma_f32_32x32x16_bf16: with the NHWC frame a window row (7 pixels x 3 channels) is 21 CONTIGUOUS
//     elements of the staged patch row, so a lane's 8-element operand slice is 16 contiguous bytes of LDS (4-byte
//     aligned: four ds_read_b32).  The slots past 21 read the neighbouring pixels' values against zero weights.
//   * the patch (24 rows x 63 pixels, frame -> bf16 while staging: aligned 16-byte loads through a buffer descriptor, zero
//     outside the frame = zero padding in the NORMALISED domain, as `Normalize` precedes the conv in the reference) is 9 KB,
//     the conv tile 41 KB.
//   * swapped MFMA operands (D[channel][pixel]) as in conv_igemm.hip: a lane owns one pixel and 4 consecutive channels
//     per 4 accumulator registers -> bias + ReLU + one rounding to bf16 + 8-byte LDS stores.
//   * the weights (64 x 176 bf16 = 22 KB) live in registers as ready-made MFMA fragments for the whole persistent launch.
//   * max-pool on the bf16 bit patterns: every value is >= 0 after the ReLU, where bf16 order == unsigned 16-bit order
//     (v_pk_max_u16); out-of-frame conv pixels are stored as 0, which never wins against the window's real values >= 0
//     (PyTorch pads the pool with -inf: same result).
#include "common.h"

namespace {

constexpr int TPH = 4, TPW = 14;                 // pooled tile
constexpr int CTH = 2 * TPH + 1, CTW = 2 * TPW + 1;   // conv tile 9 x 29
constexpr int NPX = CTH * CTW;                   // 261
constexpr int NBLK = (NPX + 31) / 32;            // 9 MFMA pixel blocks
constexpr int PROWS = 2 * CTH + 5 + 1;           // 23 patch rows + 1 (the zero-weight K row 7 reads it)
constexpr int PPITCH = 192;                      // elements per patch row (63 pixels x 3 = 189)
constexpr int NCHUNK = 49;                       // aligned 4-element chunks per patch row (elements -1 .. 194 of the row)
constexpr int KROW = 24, KP = 176, NKS = KP / 16;
constexpr int CT_PITCH = 64 * 2 + 16;            // bytes per conv pixel in LDS (+16: staggers the banks)
constexpr int PATCH_BYTES = PROWS * PPITCH * 2;  // 9,216
constexpr int CT_BYTES = NBLK * 32 * CT_PITCH;  // 41,472
constexpr int LDS_BYTES = PATCH_BYTES + CT_BYTES + 64 * 4;      // + the bias vector = 50,944 B per workgroup
constexpr int NT = 256;                          // 4 waves: with two workgroups per CU the phases of one tile (stage -> MFMA ->
                                                 // pool) run beside the other workgroups' phases

typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_s7 __attribute__((ext_vector_type(4)));

template <bool U8>
__global__ __launch_bounds__(NT, 2) void stem7_pool_kernel(const void* __restrict__ rgb_, const uint16_t* __restrict__ w,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                           int B, int H, int W, int tiles_x, int tiles_y, float3 nscale,
                                                           float3 nshift, unsigned in_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* patch = reinterpret_cast<uint16_t*>(smem);
    unsigned char* ct = smem + PATCH_BYTES;
    float* bias_s = reinterpret_cast<float*>(smem + PATCH_BYTES + CT_BYTES);   // (in LDS: a global load in the epilogue would wait, in
    if (threadIdx.x < 64) bias_s[threadIdx.x] = bias[threadIdx.x];               //  vmcnt order, for the NEXT tile's prefetched chunks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int Hc = H >> 1, Wc = W >> 1, Hp = Hc >> 1, Wp = Wc >> 1;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rgb_, 0, in_bytes, 0x00020000);
#endif

    // weight fragments, once per workgroup: row n = j * 32 + frow, k = ks * 16 + fhalf * 8 .. + 7
    s16x8_t bfr[2][NKS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            bfr[j][ks] = *reinterpret_cast<const s16x8_t*>(w + (size_t)(j * 32 + frow) * KP + ks * 16 + fhalf * 8);

    const int ntiles = B * tiles_x * tiles_y;
    constexpr int NCH = PROWS * NCHUNK, IT = (NCH + NT - 1) / NT;
    // Patch staging: aligned 4-element chunks (16 B of fp32 / 4 B of uint8) through a buffer descriptor -- a chunk outside the
    // frame is an out-of-range offset that reads as zeros (W * 3 is a multiple of 4, so a chunk is entirely inside or outside).
    // The chunks of tile i + 1 are FETCHED (into registers) right after tile i's patch is complete, so their HBM latency runs
    // under tile i's MFMA and pooling phases.
    u32x4_s7 raw[IT];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int iy0 = 4 * ty * TPH - 5, al0 = (4 * tx * TPW - 5) * 3 - 1;       // (al0: row element of chunk 0, a multiple of 4)
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int e = tid + NT * i;
            e = e < NCH ? e : NCH - 1;
            const int r = e / NCHUNK, q = e - r * NCHUNK;
            const int iy = iy0 + r, col = al0 + 4 * q;
            const bool ok = tile < ntiles && iy >= 0 && iy < H && col >= 0 && col < W * 3;
            const long idx = ((long)b * H + iy) * (long)W * 3 + col;
#if defined(__HIP_DEVICE_COMPILE__)
            const unsigned off = ok ? (unsigned)(idx * (U8 ? 1 : 4)) : 0xFFFFFFF0u;
            if (U8) raw[i] = u32x4_s7{(unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0), 0u, 0u, 0u};
            else raw[i] = __builtin_bit_cast(u32x4_s7, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
#endif
        }
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int py0 = ty * TPH, px0 = tx * TPW;
        const int iy0 = 4 * py0 - 5;
        const int al0 = (4 * px0 - 5) * 3 - 1;
        // ---- registers -> patch (bf16): chunk q holds patch elements 4q - 1 .. 4q + 2 of its row: one 4-byte store for the
        // aligned middle pair, two 2-byte stores for the ends ----
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = tid + NT * i;
            if (e < NCH) {
                const int r = e / NCHUNK, q = e - r * NCHUNK;
                float v[4];
                if (U8) {
                    const int iy = iy0 + r, col = al0 + 4 * q;
                    const bool ok = iy >= 0 && iy < H && col >= 0 && col < W * 3;
                    const unsigned u = raw[i][0];
                    const int c0 = (2 + q) % 3;                   // channel of element 0: (12 px0 - 16 + 4 q) mod 3
                    const float sc[3] = {nscale.x, nscale.y, nscale.z}, sh[3] = {nshift.x, nshift.y, nshift.z};
                    const int c1 = c0 == 2 ? 0 : c0 + 1, c2 = c1 == 2 ? 0 : c1 + 1;
                    // (padding stays exactly zero in the NORMALISED domain, as in the reference)
                    v[0] = ok ? (float)(u & 0xffu) * sc[c0] + sh[c0] : 0.f;
                    v[1] = ok ? (float)((u >> 8) & 0xffu) * sc[c1] + sh[c1] : 0.f;
                    v[2] = ok ? (float)((u >> 16) & 0xffu) * sc[c2] + sh[c2] : 0.f;
                    v[3] = ok ? (float)(u >> 24) * sc[c0] + sh[c0] : 0.f;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(raw[i][k]);
                }
                uint16_t* dst = patch + r * PPITCH + 4 * q;      // element 4q of the row; the chunk starts one element earlier
                if (q > 0) dst[-1] = ec_f2bf(v[0]);
                // (the last chunk of a row, 4 q == PPITCH, holds only element PPITCH - 1: its middle pair would land on elements
                //  0..1 of patch row r + 1 -- real taps of conv column 0 -- in the same ds_write as that row's own chunk 0)
                if (4 * q + 1 < PPITCH) *reinterpret_cast<uint32_t*>(dst) = ec_pack2(v[1], v[2]);
                else if (4 * q < PPITCH) dst[0] = ec_f2bf(v[1]);
                if (4 * q + 2 < PPITCH) dst[2] = ec_f2bf(v[3]);
            }
        }
        __syncthreads();
        fetch(tile + (int)gridDim.x);                            // the next tile's chunks: in flight under this tile's phases (same-box A/B:
                                                                 // neutral on fp32 frames, 98-100 -> 93 us per 128 uint8 frames)
        // ---- conv: wave w owns pixel blocks w, w + 4 (, w + 8) of the 9 x 29 tile, all 64 channels, one block at a time ----
        for (int blk = wave; blk < NBLK; blk += NT / 64) {
            const int idx = blk * 32 + frow;
            const int idc = idx < NPX ? idx : NPX - 1;            // (padding rows of the last block: results discarded)
            const int cy = idc / CTW, cx = idc - cy * CTW;
            const int pbase = (2 * cy) * PPITCH + 6 * cx;         // element index of the window's (ky = 0, kx = 0, ci = 0)
            f32x16_t acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int kk = ks * 16 + fhalf * 8;
                const int ky = kk / KROW, off = kk - ky * KROW;
                const uint32_t* src = reinterpret_cast<const uint32_t*>(patch + pbase + ky * PPITCH + off);
                const uint4 v4 = make_uint4(src[0], src[1], src[2], src[3]);
                const s16x8_t af = __builtin_bit_cast(s16x8_t, v4);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bfr[j][ks]), __builtin_bit_cast(bf16x8_t, af),
                                                                     acc[j], 0, 0, 0);
            }
            // bias + ReLU + rounding -> conv tile in LDS (out-of-frame conv pixels = 0)
            const int gy = 2 * py0 - 1 + cy, gx = 2 * px0 - 1 + cx;
            const bool ok = idx < NPX && gy >= 0 && gy < Hc && gx >= 0 && gx < Wc;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias_s + j * 32 + 8 * g + 4 * fhalf);
                    const float v0 = fmaxf(acc[j][4 * g + 0] + bv.x, 0.f), v1 = fmaxf(acc[j][4 * g + 1] + bv.y, 0.f);
                    const float v2 = fmaxf(acc[j][4 * g + 2] + bv.z, 0.f), v3 = fmaxf(acc[j][4 * g + 3] + bv.w, 0.f);
                    uint2 o;
                    o.x = ok ? ec_pack2(v0, v1) : 0u;
                    o.y = ok ? ec_pack2(v2, v3) : 0u;
                    *reinterpret_cast<uint2*>(ct + idx * CT_PITCH + (j * 32 + 8 * g + 4 * fhalf) * 2) = o;
                }
        }
        __syncthreads();
        // ---- 3 x 3 / stride-2 max-pool out of LDS: one (pooled pixel, 8-channel chunk) per thread and pass ----
        for (int e = tid; e < TPH * TPW * 8; e += NT) {
            const int pp = e >> 3, c8 = e & 7;
            const int ppy = pp / TPW, ppx = pp - ppy * TPW;
            const int py = py0 + ppy, px = px0 + ppx;
            if (py < Hp && px < Wp) {
                u16x8_t m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const u16x8_t v = *reinterpret_cast<const u16x8_t*>(ct + ((2 * ppy + dy) * CTW + 2 * ppx + dx) * CT_PITCH + c8 * 16);
                        m = __builtin_elementwise_max(m, v);
                    }
                *reinterpret_cast<u16x8_t*>(out + (((long)b * Hp + py) * Wp + px) * 64 + c8 * 8) = m;
            }
        }
        __syncthreads();
    }
}

}  // namespace

// rgb: fp32 NHWC [B,H,W,3] ImageNet-normalised (u8 == 0) or raw uint8 NHWC with /255 + mean/std fused (u8 == 1; h_mean3 /
// h_std3 host pointers); w bf16 [64][176] (K layout above; encoder.pack_tv_resnet), bias f32 [64]; out bf16 [B,H/4,W/4,64].
extern "C" int ec_stem7_pool(const void* rgb, int u8, const float* h_mean3, const float* h_std3, const void* w,
                             const float* bias, void* out, int B, int H, int W, ec_stream_t stream) {
    if (!rgb || !w || !bias || !out) return EC_ERR_ARG;
    if (B <= 0 || H < 8 || W < 8 || (H & 3) || (W & 3)) return EC_ERR_SHAPE;
    if (u8 && (!h_mean3 || !h_std3)) return EC_ERR_ARG;
    const int Hp = H / 4, Wp = W / 4;
    const int tiles_y = (Hp + TPH - 1) / TPH, tiles_x = (Wp + TPW - 1) / TPW;
    const long ntiles = (long)B * tiles_x * tiles_y;
    if (ntiles >= (1L << 31)) return EC_ERR_SHAPE;
    float3 sc = make_float3(1.f, 1.f, 1.f), sh = make_float3(0.f, 0.f, 0.f);
    if (u8) {
        sc = make_float3(1.f / (255.f * h_std3[0]), 1.f / (255.f * h_std3[1]), 1.f / (255.f * h_std3[2]));
        sh = make_float3(-h_mean3[0] / h_std3[0], -h_mean3[1] / h_std3[1], -h_mean3[2] / h_std3[2]);
    }
    if ((long)B * H * W * 3 * (u8 ? 1 : 4) >= (1L << 32) - 16) return EC_ERR_SHAPE;   // (the frame tensor goes through a 32-bit buffer descriptor)
    const unsigned in_bytes = (unsigned)((long)B * H * W * 3 * (u8 ? 1 : 4));
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);      // two workgroups per CU (the weight fragments take 88 of a lane's registers)
    static std::atomic<uint64_t> attr_done{0};
    if (auto g = ec_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem7_pool_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem7_pool_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    if (u8)
        hipLaunchKernelGGL(stem7_pool_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, rgb, (const uint16_t*)w,
                           bias, (uint16_t*)out, B, H, W, tiles_x, tiles_y, sc, sh, in_bytes);
    else
        hipLaunchKernelGGL(stem7_pool_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, rgb, (const uint16_t*)w,
                           bias, (uint16_t*)out, B, H, W, tiles_x, tiles_y, sc, sh, in_bytes);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
