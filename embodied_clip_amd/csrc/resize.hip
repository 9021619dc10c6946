// CLIP image preprocessing on raw uint8 frames: Resize(n_px, BICUBIC) + CenterCrop(n_px).
//
// Replaces the PIL / torchvision half of `clip_preprocess(frame)`
// (primitive_probing/generate_data/thor_image_features.py:108; ImageNet analogue spelled out at :36-44) for the
// 300x300 frames of thor_frames.py:33-34: [U] openai/CLIP clip/clip.py `_transform`, torchvision 0.8.2
// `F.resize` / `F.center_crop` on PIL images, Pillow `Image.resize(size, BICUBIC)` == libImaging/Resample.c.
// ToTensor (/255) + Normalize(CLIP mean/std) stay fused into the stem kernel (`ec_rn50_forward_u8`).
//
// Pillow's resize is integer arithmetic once its coefficient tables exist (22-bit fixed point, int32 accumulators,
// uint8 intermediate between the horizontal and the vertical pass), so this path is BIT-EXACT with Pillow:
//   * host: `ec_clip_resize_table` builds the two coefficient tables in double precision in Resample.c's operation
//     order (precompute_coeffs + normalize_coeffs_8bpc) plus the torchvision resize / crop geometry;
//   * device: one workgroup per (frame, 8 output rows): horizontal pass of the input rows that tile needs, restricted
//     to the cropped columns, into an LDS image (uint8), barrier, vertical pass from LDS -> coalesced uint8 stores.
// HBM-bound and tiny (270 KB in, 150 KB out per frame).
#include <math.h>

#include <new>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int ROWS = 8;            // output rows per workgroup
constexpr int HDR = 16;            // table header ints

double bicubic_filter(double x) {  // Resample.c bicubic_filter, a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

int coeff_ksize(int in_size, int out_size) {
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

// precompute_coeffs + normalize_coeffs_8bpc for the full-image box; bounds[2*xx] = xmin, bounds[2*xx+1] = count
void precompute(int in_size, int out_size, int ksize, int* bounds, int* kk) {
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    double k[64];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < ksize; ++x) {
            const double v = x < xmax ? k[x] : 0.0;
            kk[xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

struct Geo { int oh, ow, top, left, ksx, ksy; };

int geometry(int H, int W, int n_px, Geo* g) {
    if (H <= 0 || W <= 0 || n_px <= 0) return EC_ERR_SHAPE;
    // torchvision 0.8.2 F.resize(img, int): smaller edge -> n_px, other edge int(n_px * long / short)
    if ((W <= H && W == n_px) || (H <= W && H == n_px)) { g->oh = H; g->ow = W; }
    else if (W < H) { g->ow = n_px; g->oh = (int)((double)n_px * H / W); }
    else { g->oh = n_px; g->ow = (int)((double)n_px * W / H); }
    if (g->oh < n_px || g->ow < n_px) return EC_ERR_SHAPE;
    // F.center_crop: int(round((H - th) / 2.)) -- Python 3 round() is round-half-even
    auto pyround = [](double v) { return (int)nearbyint(v); };
    g->top = pyround((g->oh - n_px) / 2.0);
    g->left = pyround((g->ow - n_px) / 2.0);
    g->ksx = coeff_ksize(W, g->ow);
    g->ksy = coeff_ksize(H, g->oh);
    if (g->ksx > 64 || g->ksy > 64) return EC_ERR_SHAPE;   // > 15x down-scaling: not a CLIP use case
    return EC_OK;
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// table layout (ints): [0]=oh [1]=ow [2]=top [3]=left [4]=ksx [5]=ksy [6]=max_rows | bx[2*n_px] kx[n_px*ksx] by[2*n_px] ky[n_px*ksy]
// (only the cropped n_px columns / rows are stored)
__global__ __launch_bounds__(256) void resize_crop_kernel(const unsigned char* __restrict__ in, const int* __restrict__ tab,
                                                         unsigned char* __restrict__ out, int H, int W, int n_px) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tmp[];   // [rows][n_px][3]
    const int ksx = tab[4], ksy = tab[5];
    const int* bx = tab + HDR;
    const int* kx = bx + 2 * n_px;
    const int* by = kx + n_px * ksx;
    const int* ky = by + 2 * n_px;
    const int b = blockIdx.y, r0 = blockIdx.x * ROWS;
    const int r1 = min(r0 + ROWS, n_px);
    const int y0 = by[2 * r0];
    const int y1 = by[2 * (r1 - 1)] + by[2 * (r1 - 1) + 1];          // input rows [y0, y1) feed this tile
    const unsigned char* src = in + (long)b * H * W * 3;
    const int row_elems = n_px * 3;
    // horizontal pass: (row, col, c) items, col fastest so a wave reads neighbouring input pixels
    for (int it = threadIdx.x; it < (y1 - y0) * row_elems; it += 256) {
        const int ry = it / row_elems, e = it - ry * row_elems;
        const int col = e / 3, c = e - col * 3;
        const int xmin = bx[2 * col], n = bx[2 * col + 1];
        const unsigned char* p = src + ((long)(y0 + ry) * W + xmin) * 3 + c;
        const int* k = kx + col * ksx;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < n; ++x) acc += (int)p[3 * x] * k[x];
        tmp[it] = clip8(acc);
    }
    __syncthreads();
    // vertical pass from the LDS image
    unsigned char* dst = out + ((long)b * n_px + r0) * row_elems;
    for (int it = threadIdx.x; it < (r1 - r0) * row_elems; it += 256) {
        const int r = it / row_elems, e = it - r * row_elems;
        const int ymin = by[2 * (r0 + r)], n = by[2 * (r0 + r) + 1];
        const int* k = ky + (r0 + r) * ksy;
        const unsigned char* p = tmp + (ymin - y0) * row_elems + e;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < n; ++y) acc += (int)p[y * row_elems] * k[y];
        dst[it] = clip8(acc);
    }
}

}  // namespace

extern "C" size_t ec_clip_resize_table_ints(int H, int W, int n_px) {
    Geo g;
    if (geometry(H, W, n_px, &g) != EC_OK) return 0;
    return (size_t)HDR + 2 * (size_t)n_px + (size_t)n_px * g.ksx + 2 * (size_t)n_px + (size_t)n_px * g.ksy;
}

extern "C" int ec_clip_resize_table(int H, int W, int n_px, int* host_table, size_t n_ints) {
    Geo g;
    if (!host_table) return EC_ERR_ARG;
    int rc = geometry(H, W, n_px, &g);
    if (rc != EC_OK) return rc;
    if (n_ints < ec_clip_resize_table_ints(H, W, n_px)) return EC_ERR_WORKSPACE;
    int* bx_full = new (std::nothrow) int[2 * (size_t)g.ow + (size_t)g.ow * g.ksx + 2 * (size_t)g.oh + (size_t)g.oh * g.ksy];
    if (!bx_full) return EC_ERR_ALLOC;
    int* kx_full = bx_full + 2 * g.ow;
    int* by_full = kx_full + (size_t)g.ow * g.ksx;
    int* ky_full = by_full + 2 * g.oh;
    precompute(W, g.ow, g.ksx, bx_full, kx_full);
    precompute(H, g.oh, g.ksy, by_full, ky_full);
    for (int i = 0; i < HDR; ++i) host_table[i] = 0;
    host_table[0] = g.oh; host_table[1] = g.ow; host_table[2] = g.top; host_table[3] = g.left;
    host_table[4] = g.ksx; host_table[5] = g.ksy;
    int* bx = host_table + HDR;
    int* kx = bx + 2 * n_px;
    int* by = kx + (size_t)n_px * g.ksx;
    int* ky = by + 2 * n_px;
    for (int c = 0; c < n_px; ++c) {           // the crop keeps resized columns [left, left + n_px) and rows [top, top + n_px)
        bx[2 * c] = bx_full[2 * (g.left + c)]; bx[2 * c + 1] = bx_full[2 * (g.left + c) + 1];
        for (int x = 0; x < g.ksx; ++x) kx[c * g.ksx + x] = kx_full[(g.left + c) * g.ksx + x];
        by[2 * c] = by_full[2 * (g.top + c)]; by[2 * c + 1] = by_full[2 * (g.top + c) + 1];
        for (int y = 0; y < g.ksy; ++y) ky[c * g.ksy + y] = ky_full[(g.top + c) * g.ksy + y];
    }
    int max_rows = 0;
    for (int r0 = 0; r0 < n_px; r0 += ROWS) {
        const int r1 = (r0 + ROWS < n_px ? r0 + ROWS : n_px) - 1;
        const int rows = by[2 * r1] + by[2 * r1 + 1] - by[2 * r0];
        if (rows > max_rows) max_rows = rows;
    }
    host_table[6] = max_rows;
    delete[] bx_full;
    return EC_OK;
}

extern "C" int ec_clip_resize_crop_u8(const uint8_t* frames_u8, const int* table_dev, int table_max_rows, uint8_t* out_u8,
                                      int B, int H, int W, int n_px, ec_stream_t stream) {
    if (!frames_u8 || !table_dev || !out_u8) return EC_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || n_px <= 0 || table_max_rows <= 0) return EC_ERR_SHAPE;
    const size_t lds = (size_t)table_max_rows * n_px * 3;
    if (lds > 160 * 1024) return EC_ERR_SHAPE;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resize_crop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    dim3 grid((unsigned)((n_px + ROWS - 1) / ROWS), (unsigned)B);
    hipLaunchKernelGGL(resize_crop_kernel, grid, dim3(256), lds, (hipStream_t)stream, frames_u8, table_dev, out_u8, H, W, n_px);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
