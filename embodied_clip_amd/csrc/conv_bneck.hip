// Bottleneck-level fusion for the 14x14 stage of the CLIP-RN50 trunk (round 4): conv2 (3x3, C -> C) + bn2 + ReLU and
// conv3 (1x1, C -> 4C) + bn3 + identity + ReLU of one Bottleneck in ONE launch, ONE WORKGROUP PER IMAGE.
//
// Replaces, inside `clip_model(clip_input)` (primitive_probing/generate_data/thor_image_features.py:109; [U] openai/CLIP
// clip/model.py Bottleneck.forward: `out = relu2(bn2(conv2(out))); out = avgpool(out) [identity here]; out = bn3(conv3(out));
// out += identity; out = relu3(out)`), the two launches conv_igemm runs for layer3.1 .. layer3.5.
//
// Why (VERDICT r3 item 2 / SURVEY.md section 7 "hard parts"): tiled over output pixels, every 128..256-pixel tile of conv2
// streams its im2col operand AND the weights through the CU's L2 -> LDS path (~30 B/clk/CU as those kernels issue them; a pure L2-resident stream reaches 64-74 B/clk/CU,
// tools/ubench/l2_feed.hip, round 5), and conv3 re-reads
// conv2's output from HBM/L2 once per N tile; both launches sit at the sum of their MFMA and operand-feed times.  A 14x14x256
// map is 98 KB: it FITS the LDS.  So a workgroup keeps one image's conv1 output resident (T, 196 rows x 512 B + a zero
// row), reads the nine taps of the im2col operand straight out of it (a tap is a row shift; out-of-frame taps read the
// zero row -- no address arithmetic in the K loop beyond nine v_cndmask per M block and tap), overwrites T with conv2's
// output and runs conv3 out of the same LDS image.  Only the weights (1.7 MB per image, L2-resident) and the residual /
// output rows move: 2.2 MB per image through L2 -> LDS instead of 3.4 MB, x read once (as the residual), c2 never written.
//
// Structure (wave64, v_mfma_f32_32x32x16_bf16, swapped operands D[n][m] as in conv_igemm.hip):
//   * 8 waves = 1 (M) x 8 (N): wave w owns output channels [32 w, 32 w + 32) of conv2 and, in each of the four conv3
//     passes, channels [256 q + 32 w, +32); ALL 7 pixel blocks (196 of 224 rows) -> 7 accumulator tiles (112 VGPRs),
//     8 fragment reads per 7 MFMAs.
//   * the weight operand is PRIVATE to a wave (its 32 rows of W), so each wave streams its own K-tiles (32 x 32 bf16 =
//     2 KB, two 1-KB LDS-DMA pieces) through its own 3-stage ring: NO workgroup barrier in either K loop -- a wave waits
//     for its own pieces with a counted s_waitcnt vmcnt (MI355X_MICROARCH.md: the issuing wave's covering vmcnt orders its
//     own ds_read behind its LDS-DMA).  T is read-only inside a K loop.  Barriers: after T is loaded, and around the
//     overwrite of T with conv2's output -- three per image.
//   * fragment reads are inline asm one k-step ahead of the MFMAs that use them (the compiler would put s_waitcnt
//     vmcnt(0) in front of every LDS read that follows an LDS-DMA); waits carry the fragment registers.
//   * conv3's epilogue runs per 32-pixel block through the wave's free ring stage: residual rows in as 16-byte chunks
//     (64 B contiguous per pixel), out as 16-byte chunks -- bias + identity + ReLU + ONE rounding to bf16 in between.
//
// Rounding points are those of the unfused path (bf16 c2, fp32 accumulate, one rounding of y) and each output element sums
// its K walk in the same order (tap-major, channel-minor; k-steps in order), so results are bit-identical to conv_igemm's.
#include <stdlib.h>

#include <algorithm>
#include <utility>

#include "common.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct BneckArgs {
    const uint16_t* c1;    // [B][PIX][C]   conv1 output (bf16, post-ReLU)
    const uint16_t* w2;    // conv2 weights ([C][9 C], K = (ky, kx, ci)) in STREAMING ORDER (ec_bneck_pack_weights)
    const float* b2;       // [C]
    const uint16_t* w3;    // conv3 weights ([4C][C]) in streaming order
    const float* b3;       // [4C]
    const uint16_t* xres;  // [B][PIX][4C]  block input (identity; with F1 also conv1's input)
    uint16_t* y;           // [B][PIX][4C]
    const uint16_t* w1;    // F1: conv1 weights ([C][4C]) in streaming order
    const float* b1;       // F1: [C]
    int B;
    unsigned w2_bytes, w3_bytes, w1_bytes;
    unsigned long long* dbg;   // profiling only (ec_bneck_set_debug): workgroup 0 stores {s_memtime, s_memrealtime} at entry / phase ends
    int dbg_mode = 0;          // tools build only: 1 = return at entry (the launch's fixed cost), 2 = return after conv1
};

__device__ __forceinline__ float bn_relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }

template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4_t& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void tie(u32x4_t& d) { asm volatile("" : "+v"(d)); }
// Epilogue LDS traffic in inline asm too: behind an LDS-DMA load (`buffer_load ... lds`) the compiler guards every LDS read it
// generates itself with s_waitcnt vmcnt(0) (the DMA may alias it) -- which would wait for the next pass's weight tiles and every
// prefetched identity row at the top of each epilogue.  The waits (lgkmcnt, in-order LDS) are written out by hand.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_read8m(u32x2_t& d, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void lds_read16m(u32x4_t& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void lds_write8m(unsigned addr, u32x2_t v) { asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write16m(unsigned addr, u32x4_t v) { asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void tie2(u32x2_t& d) { asm volatile("" : "+v"(d)); }

// F1: conv1 (1x1, 4C -> C) + bn1 + ReLU runs in the same launch, in front: the whole Bottleneck is one launch, c1 never
// exists in HBM.  Its im2col operand is the block input x (196 x 1024: does not fit), so x streams through a SHARED 3-stage
// ring of 16-KB K-tiles (256 rows x 32 k, inside T's still unused area; every wave issues two of a K-tile's sixteen 1-KB pieces
// next to its two private weight pieces: four per wave and K-tile, so the counted vmcnt stays uniform) with ONE raw barrier per
// K-tile (ring mode of conv_igemm: own pieces landed -> barrier -> issue K-tile kt + 2 -> fragments + MFMAs).
// WEMU (tools build only, tools/bench_bneck.py --wino-emu): FEED EMULATION of a Winograd F(2x2, 3x3) conv2 -- NOT a conv (the
// results are garbage): conv2's K loop is run with the operand traffic and the MFMA count the Winograd form would have in this
// kernel's structure -- 16 positions x (C / 32) K-tiles of PRIVATE transformed weights per wave (16 C^2 instead of 9 C^2
// elements: 2.1 MB per image through the L2 -> LDS path) against 2 pixel blocks (49 tiles of 2 x 2 outputs -> 64 rows) instead
// of 7 -- i.e. an UPPER bound of what the transform could buy (input / output transforms, their barriers and the LDS for the
// transformed map are not charged).  docs/experiments.md section C has the stamps.
template <int C, int HW, bool F1 = false, bool WEMU = false>
__global__ __launch_bounds__(512, 2) void bneck23_kernel(BneckArgs p) {
    constexpr int PIX = HW * HW, MB = (PIX + 31) / 32;          // 196 pixels, 7 blocks of 32
    constexpr int PITCH = C * 2 + 16;                            // T row pitch (bytes): +16 staggers the banks between rows
    constexpr int T_BYTES = (PIX + 1) * PITCH;                   // + the zero row (index PIX)
    constexpr int BK = 32, NS = 3, STAGE = 32 * BK * 2, RING = NS * STAGE;   // per-wave weight ring: 3 x 2 KB
    constexpr int NTAP = WEMU ? 16 : 9, MBC2 = WEMU ? 2 : (PIX + 31) / 32;   // (taps | Winograd positions; pixel blocks of conv2's K loop)
    constexpr int K2 = NTAP * C, NK2 = K2 / BK, KT_PER_TAP = C / BK;
    constexpr int K3 = C, NK3 = K3 / BK, NPASS = 4 * C / 256;
    static_assert(C == 256, "8 waves x 32 channels");
    static_assert(T_BYTES + 8 * RING <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* T = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, hh = lane >> 5;
    unsigned char* ring = smem + T_BYTES + wave * RING;
    const unsigned ring_lds = (unsigned)(unsigned long)(lds_void_t*)ring;
    const unsigned t_lds = (unsigned)(unsigned long)(lds_void_t*)T;
    const int img = blockIdx.x;
    auto stamp = [&](int slot) {   // profiling only: shader-clock and 100-MHz real-time counters of workgroup 0 / wave 0
        if (p.dbg && blockIdx.x == 0 && tid == 0) {
            p.dbg[2 * slot] = __builtin_amdgcn_s_memtime();
            p.dbg[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        }
    };
#ifdef EC_TOOLS
    if (p.dbg_mode == 1) return;
#endif
    stamp(0);
#ifdef EC_TOOLS   // tools build: EVERY workgroup's start / end on the 100-MHz real-time counter + its XCC id (slots 64 + 4 img ..)
    if (p.dbg && tid == 0) {
        p.dbg[64 + 4 * img] = __builtin_amdgcn_s_memrealtime();
        p.dbg[64 + 4 * img + 2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
    }
#endif

    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w3_bytes, 0x00020000);

    // ---- weight ring: K-tile kt of the wave's 32 rows -> stage st (two 1-KB pieces: 16 rows x 64 B each) ----
    // LDS image of a stage: row r (0..31) x 64 B, 16-byte chunk c stored at c ^ ((r >> 2) & 3) (conflict-free ds_read_b128
    // over the 16-lane groups); LDS-DMA writes lane-linear, so the swizzle is applied to the SOURCE chunk.
    // The weights arrive PACKED in this order (ec_bneck_pack_weights): the 2-KB stage image of (32-row slice, K-tile) is one
    // contiguous block, so a piece is a 1-KB contiguous read (8 full cache lines, consecutive channels of the L2).  Fetched
    // out of the plain [n][K] layout a piece touched 16 rows x 64 B, 4.6 KB apart: half-used lines on 4 of the 16 L2
    // channels, the same ones for every wave of every workgroup at the same time (per-image time 64 us alone -> 105 us
    // with 256 workgroups in flight).
    const unsigned wsrc2 = (unsigned)(wave * NK2 * STAGE + lane * 16);   // slice `wave`, K-tile 0, piece 0
    const unsigned wsrc3 = (unsigned)(wave * NK3 * STAGE + lane * 16);   // (+ pass * 8 slices: added per pass)
    auto issue_w2 = [&](int kt, int st) {                        // (kt >= NK2: out-of-range offsets, zeros into a free stage)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = kt < NK2 ? wsrc2 + (unsigned)kt * STAGE + j * 1024 : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_void_t*)(ring + st * STAGE + j * 1024), 16, off, 0, 0, 0);
        }
    };
    // B fragment of k-step ks (0, 1) of a stage: row frow, chunk 2 ks + hh
    unsigned boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[ks] = (unsigned)(frow * 64 + (((2 * ks + hh) ^ ((frow >> 2) & 3)) << 4));

    if constexpr (F1) {
        if (tid < PITCH / 16) *reinterpret_cast<uint4*>(T + PIX * PITCH + tid * 16) = make_uint4(0, 0, 0, 0);   // T's zero row
    } else {
    issue_w2(0, 0);
    issue_w2(1, 1);

    // ---- T <- this image's conv1 output (PIX rows x C bf16, 16-byte chunks through registers: the rows are padded) ----
        constexpr int CHUNKS = PIX * C / 8, IT = (CHUNKS + 511) / 512;
        const uint4* src = reinterpret_cast<const uint4*>(p.c1 + (size_t)img * PIX * C);
        uint4 v[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int q = tid + i * 512;
            q = q < CHUNKS ? q : CHUNKS - 1;
            v[i] = src[q];
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * 512;
            if (q < CHUNKS) *reinterpret_cast<uint4*>(T + (q / (C / 8)) * PITCH + (q % (C / 8)) * 16) = v[i];
        }
        if (tid < PITCH / 16) *reinterpret_cast<uint4*>(T + PIX * PITCH + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    stamp(1);

    // ---- per-lane geometry of the 7 pixel blocks: base address of the lane's pixel row and its 9-bit tap mask ----
    unsigned pbase[MB], pmask[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int px = i * 32 + frow;
        const int y = px / HW, x = px - y * HW;
        unsigned m = 0;
        if (px < PIX) {
            const unsigned xm = (x > 0 ? 1u : 0u) | 2u | (x < HW - 1 ? 4u : 0u);
            m = (y > 0 ? xm : 0u) | (xm << 3) | (y < HW - 1 ? (xm << 6) : 0u);
        }
        pmask[i] = m;
        pbase[i] = t_lds + (unsigned)(px < PIX ? px : PIX) * PITCH + hh * 16;
    }
    const unsigned zbase = t_lds + PIX * PITCH + hh * 16;

    f32x16_t acc[MB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    };
    u32x4_t fa[2][MB], fb[2];                                    // fragment double buffer: [k-step parity]
    auto mma = [&](int par) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[par]), __builtin_bit_cast(bf16x8_t, fa[par][i]),
                                                             acc[i], 0, 0, 0);
    };

    [[maybe_unused]] auto mma2 = [&](int par) {                 // (WEMU: two pixel blocks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[par]), __builtin_bit_cast(bf16x8_t, fa[par][i]),
                                                             acc[i], 0, 0, 0);
    };
    if constexpr (F1) {
        // =================================================================================================================
        // conv1: c1 = relu(x W1^T + b1), K = 4C; x through the shared ring in T's area, W1 through the private rings
        // =================================================================================================================
        constexpr int K1 = 4 * C, NK1 = K1 / BK, XSTAGE = 256 * BK * 2;      // 16 KB: 256 rows x 64 B (rows >= PIX read as zeros)
        static_assert(NS * XSTAGE <= PIX * PITCH, "the x ring lives below T's zero row");
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xres + (size_t)img * PIX * K1), 0,
                                                                              (unsigned)(PIX * K1 * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w1_bytes, 0x00020000);
        unsigned xsrc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {                            // piece (2 wave + j): rows 16 (2 wave + j) + lane / 4, swizzled source chunk
            const int r = (2 * wave + j) * 16 + (lane >> 2), sc = (lane & 3) ^ ((r >> 2) & 3);
            xsrc[j] = r < PIX ? (unsigned)(r * (K1 * 2) + sc * 16) : 0xFFFFFFF0u;
        }
        const unsigned wsrc1 = (unsigned)(wave * NK1 * STAGE + lane * 16);
        auto issue1 = [&](int kt, int stg) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned xo = (kt < NK1 && xsrc[j] != 0xFFFFFFF0u) ? xsrc[j] + (unsigned)kt * (BK * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(T + stg * XSTAGE + (2 * wave + j) * 1024), 16, xo, 0, 0, 0);
                const unsigned wo = kt < NK1 ? wsrc1 + (unsigned)kt * STAGE + j * 1024 : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_void_t*)(ring + stg * STAGE + j * 1024), 16, wo, 0, 0, 0);
            }
        };
        // Schedule per K-tile kt (stage kt % 3): the fragments of its first k-step were read right after the barrier that published
        // it; read the second k-step's, run the first k-step's MFMAs under those reads, wait for the own pieces of K-tile kt + 1,
        // barrier (K-tile kt + 1 landed everywhere, nobody reads K-tile kt any more), issue K-tile kt + 3 into K-tile kt's stage,
        // read the first k-step of K-tile kt + 1 and run the second k-step's MFMAs under those reads.  Every LDS read of the eight
        // waves (128 KB per K-tile) runs under MFMAs; the barrier sits between the two halves instead of in front of both.
        auto rd_x = [&]<int KS>(std::integral_constant<int, KS>, int stg) {
            const unsigned xa = t_lds + stg * XSTAGE + boff[KS];                  // (row swizzle depends on frow only)
            [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read16<I * 2048>(fa[KS][I], xa), ...); }(std::make_integer_sequence<int, MB>{});
            lds_read16<0>(fb[KS], ring_lds + stg * STAGE + boff[KS]);
        };
        issue1(0, 0);
        issue1(1, 1);
        issue1(2, 2);
        zero_acc();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // this wave's four pieces of K-tile 0
        __builtin_amdgcn_s_barrier();
        rd_x(std::integral_constant<int, 0>{}, 0);
        int s1 = 0;
        for (int kt = 0; kt < NK1; ++kt) {
            const int s1n = s1 + 1 == NS ? 0 : s1 + 1;
            rd_x(std::integral_constant<int, 1>{}, s1);
            asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(MB + 1));
            [&]<int... I>(std::integer_sequence<int, I...>) { (tie(fa[0][I]), ...); }(std::make_integer_sequence<int, MB>{});
            tie(fb[0]);
            mma(0);
            asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // own pieces of K-tile kt + 1 (kt + 2's in flight); second k-step in registers
            __builtin_amdgcn_s_barrier();                        // K-tile kt + 1 landed everywhere; everyone is done reading K-tile kt
            issue1(kt + 3, s1);                                  // into the stage K-tile kt has left
            rd_x(std::integral_constant<int, 0>{}, s1n);
            [&]<int... I>(std::integer_sequence<int, I...>) { (tie(fa[1][I]), ...); }(std::make_integer_sequence<int, MB>{});
            tie(fb[1]);
            mma(1);
            s1 = s1n;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the reads of the zero-filled K-tile past the end)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the zero-filled tail tiles)
        issue_w2(0, 0);                                          // conv2's first K-tiles under the c1 write
        issue_w2(1, 1);
        __syncthreads();                                         // every wave is done reading the x ring: T becomes c1
        {
            const int n0 = wave * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(p.b1 + n0 + 8 * g + 4 * hh);
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int px = i * 32 + frow;
                    uint2 o;
                    o.x = ec_pack2(bn_relu(acc[i][4 * g + 0] + bv.x), bn_relu(acc[i][4 * g + 1] + bv.y));
                    o.y = ec_pack2(bn_relu(acc[i][4 * g + 2] + bv.z), bn_relu(acc[i][4 * g + 3] + bv.w));
                    if (px < PIX) *reinterpret_cast<uint2*>(T + px * PITCH + (n0 + 8 * g + 4 * hh) * 2) = o;
                }
            }
        }
        __syncthreads();                                         // T holds c1
        stamp(1);
    }

    // =====================================================================================================================
    // conv2: K = 9 taps x C channels, K-tiles of 32 channels; tap loop at run time, the 8 K-tiles x 2 k-steps of a tap unrolled
    // =====================================================================================================================
    zero_acc();
    int st = 0;                                                  // ring stage of the K-tile being computed
    for (int tap = 0; tap < NTAP; ++tap) {
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;       // tap / 3, tap % 3
        const int shift = WEMU ? 0 : ((ky - 1) * HW + (kx - 1)) * PITCH;
        unsigned aaddr[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) aaddr[i] = (WEMU || ((pmask[i] >> tap) & 1u)) ? pbase[i] + (unsigned)shift : zbase;
        // first k-step of the tap: its K-tile's pieces must have landed (the two newest pieces in flight belong to the next tile)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        {
            const unsigned bb = ring_lds + st * STAGE + boff[0];
            [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read16<0>(fa[0][I], aaddr[I]), ...); }(std::make_integer_sequence<int, MBC2>{});
            lds_read16<0>(fb[0], bb);
        }
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                constexpr int s = S, j = s >> 1, ks = s & 1, par = s & 1;            // step s of the tap: K-tile j, k-step ks
                const int kt = tap * KT_PER_TAP + j;
                if constexpr (ks == 0) {
                    int st2 = st + 2; st2 = st2 >= NS ? st2 - NS : st2;
                    issue_w2(kt + 2, st2);                                            // into the stage K-tile kt - 1 has left
                }
                if constexpr (s + 1 < 2 * KT_PER_TAP) {                               // prefetch the next step's fragments
                    constexpr int s1 = s + 1, j1 = s1 >> 1, ks1 = s1 & 1;
                    int stn = st;
                    if constexpr (ks1 == 0) {                                         // ... of the next K-tile: landed?
                        stn = st + 1; stn = stn >= NS ? stn - NS : stn;
                        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");              // (all but K-tile kt + 2's two pieces)
                    }
                    const unsigned bb = ring_lds + stn * STAGE + boff[ks1];
                    [&]<int... I>(std::integer_sequence<int, I...>) {
                        (lds_read16<j1 * 64 + ks1 * 32>(fa[par ^ 1][I], aaddr[I]), ...);
                    }(std::make_integer_sequence<int, MBC2>{});
                    lds_read16<0>(fb[par ^ 1], bb);
                    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(MBC2 + 1));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)");
                }
                [&]<int... I>(std::integer_sequence<int, I...>) { (tie(fa[par][I]), ...); }(std::make_integer_sequence<int, MBC2>{});
                tie(fb[par]);
                if constexpr (WEMU) mma2(par); else mma(par);
                if constexpr (ks == 1) { st = st + 1; st = st >= NS ? st - NS : st; }
            }(), ...);
        }(std::make_integer_sequence<int, 2 * KT_PER_TAP>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the zero-filled tail tiles
    stamp(2);
    __syncthreads();                                             // every wave is done reading T as conv1's output

    // ---- T <- c2 = relu(conv2 + b2) in bf16: a lane owns one pixel and, per 4 registers, 4 consecutive channels ----
    {
        const int n0 = wave * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n0 + 8 * g + 4 * hh);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int px = i * 32 + frow;
                uint2 o;
                o.x = ec_pack2(bn_relu(acc[i][4 * g + 0] + bv.x), bn_relu(acc[i][4 * g + 1] + bv.y));
                o.y = ec_pack2(bn_relu(acc[i][4 * g + 2] + bv.z), bn_relu(acc[i][4 * g + 3] + bv.w));
                if (px < PIX) *reinterpret_cast<uint2*>(T + px * PITCH + (n0 + 8 * g + 4 * hh) * 2) = o;
            }
        }
    }
    // conv3's first two K-tiles (pass 0) while the other waves finish their rows
    auto issue_w3 = [&](int pass, int kt, int stg) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = (pass < NPASS && kt < NK3) ? wsrc3 + (unsigned)pass * (8u * NK3 * STAGE) + (unsigned)kt * STAGE + j * 1024 : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_void_t*)(ring + stg * STAGE + j * 1024), 16, off, 0, 0, 0);
        }
    };
    issue_w3(0, 0, 0);
    issue_w3(0, 1, 1);
    __syncthreads();                                             // T now holds c2
    stamp(3);
    // =====================================================================================================================
    // conv3: four passes of 256 output channels (32 per wave), K = C; epilogue per 32-pixel block through the free ring stage
    // =====================================================================================================================
    const size_t img_off = (size_t)img * PIX * (4 * C);
    auto wstamp = [&](int slot) {   // profiling only: shader clock of waves 0 and 4 (the two waves of SIMD 0) inside the conv3 passes
        if (p.dbg && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0) p.dbg[16 + (wave >> 2) * 16 + slot] = __builtin_amdgcn_s_memtime();
    };
    // Identity rows run RD blocks ahead of the epilogue (RD x 2 x 16 B per lane in registers): the first RD blocks' loads are
    // issued inside the K loop (after its last weight K-tile is on its way), block i + RD's as soon as block i's registers have
    // been consumed.  (One block ahead -- what a block's ~400 clocks of math cover -- left the ~2 k clocks of an HBM load
    // exposed SEVEN times per pass: 10 k of a pass's 14.5 k clocks.)
    constexpr int RD = 4;
    static_assert(NK3 >= 4 && RD <= MB, "identity prefetch schedule");
    const int ec = lane & 3, er0 = lane >> 2, er1 = (lane >> 2) + 16;   // chunk-order view of a block: lane -> (pixel row, 16-B chunk)
    u32x4_t res[RD][2];
    const uint16_t* xr = p.xres + img_off + wave * 32 + ec * 8;
    auto load_res = [&](int pass, int i) {
        int px0 = i * 32 + er0, px1 = i * 32 + er1;
        px0 = px0 < PIX ? px0 : PIX - 1; px1 = px1 < PIX ? px1 : PIX - 1;
        res[i % RD][0] = *reinterpret_cast<const u32x4_t*>(xr + pass * 256 + (size_t)px0 * (4 * C));
        res[i % RD][1] = *reinterpret_cast<const u32x4_t*>(xr + pass * 256 + (size_t)px1 * (4 * C));
    };
    for (int pass = 0; pass < NPASS; ++pass) {
        zero_acc();
        wstamp(4 * pass);
        // stages at pass start: K-tile 0 in stage 0, K-tile 1 in stage 1 (issued before the previous epilogue).  vmcnt(0), not a
        // counted wait: the epilogue's global stores share the counter and loads / stores may retire out of order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const unsigned bb = ring_lds + boff[0];
            [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read16<0>(fa[0][I], pbase[I]), ...); }(std::make_integer_sequence<int, MB>{});
            lds_read16<0>(fb[0], bb);
        }
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                constexpr int s = S, kt = s >> 1, ks = s & 1, par = s & 1;
                constexpr int stc = kt % NS;
                if constexpr (ks == 0 && kt + 2 < NK3) issue_w3(pass, kt + 2, (kt + 2) % NS);
                if constexpr (ks == 0 && kt == NK3 - 3) {            // the last weight K-tile is on its way: the first RD identity blocks
#pragma unroll
                    for (int i = 0; i < RD; ++i) load_res(pass, i);
                }
                if constexpr (s + 1 < 2 * NK3) {
                    constexpr int s1 = s + 1, kt1 = s1 >> 1, ks1 = s1 & 1;
                    // K-tile kt1 has landed when at most the younger loads are outstanding (loads return in order): the next K-tile's
                    // two pieces and, from K-tile NK3 - 2 on, the 2 RD identity loads
                    if constexpr (ks1 == 0)
                        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((kt1 + 1 < NK3 ? 2 : 0) + (kt1 >= NK3 - 2 ? 2 * RD : 0)) : "memory");
                    const unsigned bb = ring_lds + (kt1 % NS) * STAGE + boff[ks1];
                    [&]<int... I>(std::integer_sequence<int, I...>) {
                        (lds_read16<kt1 * 64 + ks1 * 32>(fa[par ^ 1][I], pbase[I]), ...);
                    }(std::make_integer_sequence<int, MB>{});
                    lds_read16<0>(fb[par ^ 1], bb);
                    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(MB + 1));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)");
                }
                (void)stc;
                [&]<int... I>(std::integer_sequence<int, I...>) { (tie(fa[par][I]), ...); }(std::make_integer_sequence<int, MB>{});
                tie(fb[par]);
                mma(par);
            }(), ...);
        }(std::make_integer_sequence<int, 2 * NK3>{});
        wstamp(4 * pass + 1);
        // next pass's first two K-tiles stream in under this pass's epilogue; the epilogue stages through ring stage 2
        issue_w3(pass + 1, 0, 0);
        issue_w3(pass + 1, 1, 1);
        // E = ring stage 2: [32 px][64 B], chunk c of pixel r at c ^ ((r >> 2) & 3)
        const int n0 = pass * 256 + wave * 32;
        float4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(p.b3 + n0 + 8 * g + 4 * hh);
        uint16_t* yo = p.y + img_off + n0;
        const unsigned e_lds = ring_lds + 2 * STAGE;
        const unsigned e0a = e_lds + er0 * 64 + ((ec ^ ((er0 >> 2) & 3)) << 4), e1a = e_lds + er1 * 64 + ((ec ^ ((er1 >> 2) & 3)) << 4);
        unsigned sa[4];                                          // accumulator-layout slots: pixel frow, channels 8 g + 4 hh .. + 3
#pragma unroll
        for (int g = 0; g < 4; ++g) sa[g] = e_lds + frow * 64 + ((g ^ ((frow >> 2) & 3)) << 4) + hh * 8;
        // The epilogue of a block is two dependent LDS round trips through the wave's 2-KB tile E (identity rows in -> read in
        // accumulator layout -> bias + identity + ReLU -> bf16 back -> read as 16-B row chunks -> global store).  LDS operations
        // of a wave execute in order, so block i + 1's first round trip (A) is ISSUED right behind block i's second one (B) and
        // runs under block i's stores: one exposed LDS latency per block.
        u32x2_t rr[4];
        auto stage_a = [&](int i) {                              // identity rows of block i -> E -> accumulator layout (6 LDS operations)
            lds_write16m(e0a, res[i % RD][0]);
            lds_write16m(e1a, res[i % RD][1]);
#pragma unroll
            for (int g = 0; g < 4; ++g) lds_read8m(rr[g], sa[g]);
        };
        stage_a(0);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < 4; ++g) tie2(rr[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2_t o;
                o.x = ec_pack2(bn_relu(acc[i][4 * g + 0] + bv[g].x + ec_lo(rr[g].x)), bn_relu(acc[i][4 * g + 1] + bv[g].y + ec_hi(rr[g].x)));
                o.y = ec_pack2(bn_relu(acc[i][4 * g + 2] + bv[g].z + ec_lo(rr[g].y)), bn_relu(acc[i][4 * g + 3] + bv[g].w + ec_hi(rr[g].y)));
                lds_write8m(sa[g], o);
            }
            u32x4_t v0, v1;
            lds_read16m(v0, e0a);
            lds_read16m(v1, e1a);
            if (i + 1 < MB) {
                stage_a(i + 1);                                  // (in-order LDS: behind the two row reads above)
                if (i + RD < MB) load_res(pass, i + RD);         // (block i's registers were consumed by stage_a(i))
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            tie(v0); tie(v1);
            const int px0 = i * 32 + er0, px1 = i * 32 + er1;
            // (write-through `sc1` and non-temporal stores here were A/B-ed in round 5: 113.1 / 115.2 against 113.8 us per 256 frames --
            //  the 6-9 us between two launches are not the end-of-kernel write-back of dirty lines; docs/experiments.md section E)
            if (px0 < PIX) *reinterpret_cast<u32x4_t*>(yo + (size_t)px0 * (4 * C) + ec * 8) = v0;
            if (px1 < PIX) *reinterpret_cast<u32x4_t*>(yo + (size_t)px1 * (4 * C) + ec * 8) = v1;
        }
        wstamp(4 * pass + 2);
        stamp(4 + pass);
    }
#ifdef EC_TOOLS
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[64 + 4 * img + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv3x3_img_kernel: the 3x3 conv of a 14x14x256 / 7x7x512 map for SMALL launches (32-64 frames: the strong-scaling
// operating points, readme_files/baselines_habitat.md:63-73).  Tiled over output pixels these launches are 200-400
// workgroups that each walk 36-72 K-tiles in series at ~0.5 us per K-tile (20-30 us per conv at 32 frames for 3 us of MFMA
// work at peak).  Here a workgroup is (image, slice of 32 FN output channels): the image's input map is resident in LDS as
// in bneck23_kernel (taps = row shifts, zero row for padding), and the EIGHT WAVES SPLIT K -- wave w walks K-tiles
// [w NK/8, (w+1) NK/8) out of its private weight ring, no barrier in the loop -- so the serial chain is 9-18 K-tiles of 32.
// The eight partial tiles are folded through LDS in a FIXED order (((w0+w4)+(w1+w5))+((w2+w6)+(w3+w7))): deterministic,
// no atomics; the summation order differs from the pixel-tiled kernels', so results agree with them to fp32-accumulation
// rounding (one bf16 ulp on rare elements), not bit for bit.
struct ImgArgs {
    const uint16_t* in;    // [B][PIX][C]
    const uint16_t* w;     // [C][9 C] in streaming order (32-row slices, ec_conv3x3_img_pack)
    const float* bias;     // [C]
    uint16_t* out;         // [B][PIX][C]
    int B;
    unsigned w_bytes;
    int ldo;               // POOL variant: row stride of `out` in elements (C = dense)
};

// NCH > 1: the input map does not fit the LDS (14 x 14 x 512 = 200 KB): it is made resident in NCH channel chunks, one after
// the other, every chunk's K-tiles again split over the eight waves into the same accumulators.  POOL: CLIP's anti-aliased
// stride -- ReLU then AvgPool2d(2) of the full-resolution result, through an LDS image of the folded tiles.
template <int C, int HW, int FN, int NCH = 1, bool POOL = false>
__global__ __launch_bounds__(512, 2) void conv3x3_img_kernel(ImgArgs p) {
    constexpr int PIX = HW * HW, MB = (PIX + 31) / 32, NT = MB * FN;
    constexpr int CT = C / NCH;                                  // input channels resident at a time
    constexpr int PITCH = CT * 2 + 16, T_BYTES = (PIX + 1) * PITCH;
    constexpr int BK = 32, NS = 3, SUB = 32 * BK * 2, STAGE = FN * SUB, RING = NS * STAGE;
    constexpr int K2 = 9 * C, NK2 = K2 / BK, KT_PER_TAP = CT / BK, KPW = 9 * KT_PER_TAP / 8, NSLICE = C / (32 * FN);
    static_assert((9 * KT_PER_TAP) % 8 == 0, "a chunk's K-tiles divide over the 8 waves");
    constexpr int SLOT = NT * 4096;                              // one wave's partial tiles (fp32)
    constexpr int E_OFF = (4 * SLOT + 255) / 256 * 256;          // per-wave 2-KB output staging, behind the four reduction slots
    static_assert(NK2 % 8 == 0 && NT <= 8, "K-tiles divide over the 8 waves; one output tile per wave in the last fold");
    static_assert(T_BYTES + 8 * RING <= 160 * 1024 && E_OFF + 8 * 2048 <= 160 * 1024, "LDS budget");
    static_assert(!POOL || (FN == 1 && (HW % 2) == 0 && 4 * SLOT + PIX * 32 * 4 <= 160 * 1024), "pooled variant: 32-channel slices");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* T = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, hh = lane >> 5;
    unsigned char* ring = smem + T_BYTES + wave * RING;
    const unsigned ring_lds = (unsigned)(unsigned long)(lds_void_t*)ring;
    const unsigned t_lds = (unsigned)(unsigned long)(lds_void_t*)T;
    const int img = blockIdx.x / NSLICE, slice = blockIdx.x - img * NSLICE;      // the slices of an image are neighbours: its map is fetched from L2

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const int kt0 = wave * KPW;
    // K-tile kt of 32-row sub-slice (slice FN + f): one contiguous 2-KB block of the packed weights
    const unsigned wsrc = (unsigned)((slice * FN) * NK2 * SUB + lane * 16);
    int chunk = 0;                                               // input-channel chunk resident in T
    auto issue_w = [&](int kk, int st) {                         // K-tile kt0 + kk of the chunk (kk >= KPW: zeros into a free stage)
        const int q = kt0 + kk, tap = q / KT_PER_TAP;
        const int kts = tap * (C / BK) + chunk * KT_PER_TAP + (q - tap * KT_PER_TAP);   // its index in the [N][9 C] weight matrix
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned off = kk < KPW ? wsrc + (unsigned)(f * NK2 + kts) * SUB + j * 1024 : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(ring + st * STAGE + f * SUB + j * 1024), 16, off, 0, 0, 0);
            }
    };
    unsigned boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[ks] = (unsigned)(frow * 64 + (((2 * ks + hh) ^ ((frow >> 2) & 3)) << 4));
    auto load_T = [&](int h) {   // T <- channels [h CT, (h + 1) CT) of the image's input map (padded rows: through registers)
        constexpr int CHUNKS = PIX * CT / 8, IT = (CHUNKS + 511) / 512;
        const uint4* src = reinterpret_cast<const uint4*>(p.in + (size_t)img * PIX * C + (size_t)h * CT);
        u32x4_t v[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int q = tid + i * 512;
            q = q < CHUNKS ? q : CHUNKS - 1;
            v[i] = *reinterpret_cast<const u32x4_t*>(src + (size_t)(q / (CT / 8)) * (C / 8) + (q % (CT / 8)));
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * 512;
            if (q < CHUNKS) *reinterpret_cast<u32x4_t*>(T + (q / (CT / 8)) * PITCH + (q % (CT / 8)) * 16) = v[i];
        }
        if (tid < PITCH / 16) *reinterpret_cast<uint4*>(T + PIX * PITCH + tid * 16) = make_uint4(0, 0, 0, 0);
    };
    issue_w(0, 0);
    issue_w(1, 1);
    load_T(0);
    __syncthreads();

    unsigned pbase[MB], pmask[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int px = i * 32 + frow;
        const int y = px / HW, x = px - y * HW;
        unsigned m = 0;
        if (px < PIX) {
            const unsigned xm = (x > 0 ? 1u : 0u) | 2u | (x < HW - 1 ? 4u : 0u);
            m = (y > 0 ? xm : 0u) | (xm << 3) | (y < HW - 1 ? (xm << 6) : 0u);
        }
        pmask[i] = m;
        pbase[i] = t_lds + (unsigned)(px < PIX ? px : PIX) * PITCH + hh * 16;
    }
    const unsigned zbase = t_lds + PIX * PITCH + hh * 16;
    auto addr_of = [&](int kt, unsigned (&a)[MB]) {              // fragment addresses of K-tile kt: tap shift + channel offset
        const int tap = kt / KT_PER_TAP, j = kt - tap * KT_PER_TAP;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        const unsigned sh = (unsigned)(((ky - 1) * HW + (kx - 1)) * PITCH + j * (BK * 2));
#pragma unroll
        for (int i = 0; i < MB; ++i) a[i] = ((pmask[i] >> tap) & 1u) ? pbase[i] + sh : zbase;
    };

    f32x16_t acc[MB][FN];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;
    u32x4_t fa[2][MB], fb[2][FN];
    auto mma = [&](int par) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int f = 0; f < FN; ++f)
                acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[par][f]),
                                                                    __builtin_bit_cast(bf16x8_t, fa[par][i]), acc[i][f], 0, 0, 0);
    };
    constexpr int G = MB + FN;                                   // fragment reads per k-step
    static_assert(G <= 15, "lgkmcnt is a 4-bit counter");
    unsigned a_cur[MB], a_nxt[MB];
    for (chunk = 0; chunk < NCH; ++chunk) {
    if (chunk > 0) {                                             // next input-channel chunk: same accumulators
        __syncthreads();                                         // every wave is done reading T
        issue_w(0, 0);
        issue_w(1, 1);
        load_T(chunk);
        __syncthreads();
    }
    addr_of(kt0, a_cur);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * FN) : "memory");   // K-tile 0 landed (K-tile 1's pieces may be in flight)
    {
        [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read16<0>(fa[0][I], a_cur[I]), ...); }(std::make_integer_sequence<int, MB>{});
        [&]<int... F>(std::integer_sequence<int, F...>) { (lds_read16<F * SUB>(fb[0][F], ring_lds + boff[0]), ...); }(std::make_integer_sequence<int, FN>{});
    }
    int st = 0;
    for (int kk = 0; kk < KPW; ++kk) {
        int st2 = st + 2; st2 = st2 >= NS ? st2 - NS : st2;
        int st1 = st + 1; st1 = st1 >= NS ? st1 - NS : st1;
        // ---- k-step 0 ----
        issue_w(kk + 2, st2);                                    // into the stage K-tile kk - 1 has left
        {
            const unsigned bb = ring_lds + st * STAGE + boff[1];
            [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read16<32>(fa[1][I], a_cur[I]), ...); }(std::make_integer_sequence<int, MB>{});
            [&]<int... F>(std::integer_sequence<int, F...>) { (lds_read16<F * SUB>(fb[1][F], bb), ...); }(std::make_integer_sequence<int, FN>{});
        }
        asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(G));
        [&]<int... I>(std::integer_sequence<int, I...>) { (tie(fa[0][I]), ...); }(std::make_integer_sequence<int, MB>{});
        [&]<int... F>(std::integer_sequence<int, F...>) { (tie(fb[0][F]), ...); }(std::make_integer_sequence<int, FN>{});
        mma(0);
        // ---- k-step 1 (+ the next K-tile's first fragments) ----
        if (kk + 1 < KPW) {
            addr_of(kt0 + kk + 1, a_nxt);
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * FN) : "memory");   // K-tile kk + 1 landed
            const unsigned bb = ring_lds + st1 * STAGE + boff[0];
            [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read16<0>(fa[0][I], a_nxt[I]), ...); }(std::make_integer_sequence<int, MB>{});
            [&]<int... F>(std::integer_sequence<int, F...>) { (lds_read16<F * SUB>(fb[0][F], bb), ...); }(std::make_integer_sequence<int, FN>{});
            asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(G));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        [&]<int... I>(std::integer_sequence<int, I...>) { (tie(fa[1][I]), ...); }(std::make_integer_sequence<int, MB>{});
        [&]<int... F>(std::integer_sequence<int, F...>) { (tie(fb[1][F]), ...); }(std::make_integer_sequence<int, FN>{});
        mma(1);
#pragma unroll
        for (int i = 0; i < MB; ++i) a_cur[i] = a_nxt[i];
        st = st1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the zero-filled tail tiles)
    }   // chunks
    __syncthreads();                                             // T and the rings are dead: the LDS becomes the fold's workspace

    // ---- fold the eight partial sets in a fixed order: 4..7 -> 0..3, then tile t by wave t over the four sets ----
    auto slot_ptr = [&](int slot, int ti, int r4) { return smem + slot * SLOT + ((ti * 4 + r4) * 64 + lane) * 16; };
    if (wave >= 4) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    *reinterpret_cast<f32x4_t*>(slot_ptr(wave - 4, i * FN + f, r4)) =
                        f32x4_t{acc[i][f][4 * r4], acc[i][f][4 * r4 + 1], acc[i][f][4 * r4 + 2], acc[i][f][4 * r4 + 3]};
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4_t* sp = reinterpret_cast<f32x4_t*>(slot_ptr(wave, i * FN + f, r4));
                    const f32x4_t o = *sp;
                    f32x4_t v = f32x4_t{acc[i][f][4 * r4], acc[i][f][4 * r4 + 1], acc[i][f][4 * r4 + 2], acc[i][f][4 * r4 + 3]} + o;
                    *sp = v;                                     // (w + (w + 4)) back into the same slot: this wave only
                }
    }
    __syncthreads();
    if constexpr (POOL) {
        // folded tiles -> relu(+ bias) as fp32 into an LDS image P[px][32] behind the slots -> 2 x 2 average -> bf16 rows out
        float* P = reinterpret_cast<float*>(smem + 4 * SLOT);
        const int n0 = slice * 32;
        if (wave < NT) {
            const int ti = wave, px = ti * 32 + frow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(slot_ptr(0, ti, g)), s1 = *reinterpret_cast<const f32x4_t*>(slot_ptr(1, ti, g));
                const f32x4_t s2 = *reinterpret_cast<const f32x4_t*>(slot_ptr(2, ti, g)), s3 = *reinterpret_cast<const f32x4_t*>(slot_ptr(3, ti, g));
                const f32x4_t v = (s0 + s1) + (s2 + s3);
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + 8 * g + 4 * hh);
                if (px < PIX)
                    *reinterpret_cast<f32x4_t*>(P + px * 32 + 8 * g + 4 * hh) =
                        f32x4_t{bn_relu(v[0] + bv.x), bn_relu(v[1] + bv.y), bn_relu(v[2] + bv.z), bn_relu(v[3] + bv.w)};
            }
        }
        __syncthreads();
        constexpr int HP = HW / 2, PP = HP * HP;
        if (tid < PP * 8) {
            const int q = tid >> 3, c4 = tid & 7, yp = q / HP, xp = q - yp * HP;
            const float* b00 = P + ((2 * yp) * HW + 2 * xp) * 32 + c4 * 4;
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(b00), b = *reinterpret_cast<const f32x4_t*>(b00 + 32);
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(b00 + HW * 32), d = *reinterpret_cast<const f32x4_t*>(b00 + HW * 32 + 32);
            const f32x4_t o = ((a + b) + (c + d)) * 0.25f;       // (the fused epilogue's order: (s0 + s1) + (s2 + s3))
            uint2 w;
            w.x = ec_pack2(o[0], o[1]); w.y = ec_pack2(o[2], o[3]);
            *reinterpret_cast<uint2*>(p.out + ((size_t)img * PP + q) * p.ldo + n0 + c4 * 4) = w;
        }
        return;
    }
    if (wave < NT) {
        const int ti = wave, i = ti / FN, f = ti - i * FN;
        const int n0 = slice * (32 * FN) + f * 32;
        unsigned char* E = smem + E_OFF + wave * 2048;           // [32 px][64 B], chunk c of pixel r at c ^ ((r >> 2) & 3)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(slot_ptr(0, ti, g)), s1 = *reinterpret_cast<const f32x4_t*>(slot_ptr(1, ti, g));
            const f32x4_t s2 = *reinterpret_cast<const f32x4_t*>(slot_ptr(2, ti, g)), s3 = *reinterpret_cast<const f32x4_t*>(slot_ptr(3, ti, g));
            const f32x4_t v = (s0 + s1) + (s2 + s3);
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + 8 * g + 4 * hh);
            uint2 o;
            o.x = ec_pack2(bn_relu(v[0] + bv.x), bn_relu(v[1] + bv.y));
            o.y = ec_pack2(bn_relu(v[2] + bv.z), bn_relu(v[3] + bv.w));
            *reinterpret_cast<uint2*>(E + frow * 64 + ((g ^ ((frow >> 2) & 3)) << 4) + hh * 8) = o;
        }
        const int ec = lane & 3;
        uint16_t* yo = p.out + (size_t)img * PIX * C + n0;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int er = (lane >> 2) + 16 * h2, px = i * 32 + er;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(E + er * 64 + ((ec ^ ((er >> 2) & 3)) << 4));
            if (px < PIX) *reinterpret_cast<u32x4_t*>(yo + (size_t)px * C + ec * 8) = v;
        }
    }
}

// Streaming order of a [N][K] bf16 weight matrix for bneck23_kernel: 16-byte unit ((s * K/32 + kt) * 32 + r) * 4 + pc holds
// row 32 s + r, k = 32 kt + 8 (pc ^ ((r >> 2) & 3)) .. + 7 -- the swizzled 2-KB LDS image of (slice s, K-tile kt), contiguous.
__global__ void bneck_pack_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, int N, int K) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)N * K / 8) return;
    const int pc = (int)(t & 3), r = (int)((t >> 2) & 31);
    const long blk = t >> 7;
    const int nk = K / 32, kt = (int)(blk % nk), sidx = (int)(blk / nk);
    const int sc = pc ^ ((r >> 2) & 3);
    out[t] = w[((long)(sidx * 32 + r) * K + kt * 32 + sc * 8) >> 3];
}

}  // namespace

// relu(conv3x3(in) + bias) for the two map geometries of the trunk's late 3x3 convs, one workgroup per (image, channel slice):
// (H = W = 14, C = 256), (H = W = 7, C = 512) and, with pool = 1 (ReLU then AvgPool2d(2): out is [B,7,7,C]), (H = W = 14,
// C = 512).  in / out bf16 [B,H,W,C]; packed = ec_conv3x3_img_pack(w bf16 [C][3*3*C]).
// Meant for launches of <= 64 frames; EC_ERR_SHAPE for any other geometry.
extern "C" int ec_conv3x3_img_pack(const void* w, void* packed, int C, ec_stream_t stream) {
    if (!w || !packed) return EC_ERR_ARG;
    if (C != 256 && C != 512) return EC_ERR_SHAPE;
    const long n = (long)C * 9 * C / 8;
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)w,
                       (uint4*)packed, C, 9 * C);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
extern "C" int ec_conv3x3_img_bf16(const void* in, const void* packed, const float* bias, void* out, int B, int H, int W, int C,
                                   int pool, ec_stream_t stream) {
    return ec_conv3x3_img_bf16_ld(in, packed, bias, out, B, H, W, C, pool, C, stream);
}

// ... the pooled geometry writing a column block of a wider tensor (out_row_stride elements per pooled pixel; see ec_conv_bf16_ld)
extern "C" int ec_conv3x3_img_bf16_ld(const void* in, const void* packed, const float* bias, void* out, int B, int H, int W, int C,
                                      int pool, int out_row_stride, ec_stream_t stream) {
    if (!in || !packed || !bias || !out) return EC_ERR_ARG;
    if (B <= 0 || H != W) return EC_ERR_SHAPE;
    if (out_row_stride != C && (!pool || out_row_stride < C || (out_row_stride & 7))) return EC_ERR_SHAPE;
    ImgArgs a;
    a.in = (const uint16_t*)in; a.w = (const uint16_t*)packed; a.bias = bias; a.out = (uint16_t*)out; a.B = B;
    a.ldo = out_row_stride;
    a.w_bytes = (unsigned)((size_t)C * 9 * C * 2);
    auto go = [&](auto kern, int nslice, size_t lds, std::atomic<uint64_t>& done) {
        if (auto attr_g_ = ec_attr_needed(done))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * nslice)), dim3(512), lds, (hipStream_t)stream, a);
    };
    if (H == 14 && C == 256 && !pool) {
        static std::atomic<uint64_t> done{0};
        go(conv3x3_img_kernel<256, 14, 1>, 8, std::max<size_t>((size_t)197 * 528 + 8 * 3 * 2048, (size_t)4 * 7 * 4096 + 8 * 2048 + 256), done);
    } else if (H == 7 && C == 512 && !pool) {
        static std::atomic<uint64_t> done{0};
        go(conv3x3_img_kernel<512, 7, 2>, 8, std::max<size_t>((size_t)50 * 1040 + 8 * 3 * 4096, (size_t)4 * 4 * 4096 + 8 * 2048 + 256), done);
    } else if (H == 14 && C == 512 && pool) {   // layer4.0 conv2 + AvgPool2d(2): the 200-KB map resident in two channel chunks
        static std::atomic<uint64_t> done{0};
        go(conv3x3_img_kernel<512, 14, 1, 2, true>, 16, std::max<size_t>((size_t)197 * 528 + 8 * 3 * 2048, (size_t)4 * 7 * 4096 + 196 * 32 * 4 + 8 * 2048 + 256), done);
    } else {
        return EC_ERR_SHAPE;
    }
    EC_CHECK_LAUNCH();
    return EC_OK;
}

// profiling only: device buffer of 16 x u64 that workgroup 0 of every following fused launch fills with {shader clock,
// 100-MHz real time} stamps (entry, T loaded, conv2 done, c2 written, after each conv3 pass); nullptr switches it off
namespace { unsigned long long* g_bneck_dbg = nullptr; int g_bneck_mode = 0; }
#ifdef EC_TOOLS
extern "C" void ec_bneck_set_mode(int m) { g_bneck_mode = m; }   // tools-only build (`make tools`): not part of the product library or of include/ec_amd.h
extern "C" void ec_bneck_set_debug(void* dev_u64x16) { g_bneck_dbg = (unsigned long long*)dev_u64x16; }
#endif

// Packs conv2's [C][9C] and conv3's [4C][C] bf16 weights into the fused kernel's streaming order: packed holds C * 9C
// elements of conv2 followed by 4C * C of conv3 (ec_bneck_packed_elems).  Once per set of weights.
extern "C" size_t ec_bneck_packed_elems(int C) { return C > 0 ? (size_t)C * 9 * C + (size_t)4 * C * C : 0; }
extern "C" int ec_bneck_pack_weights(const void* w2, const void* w3, void* packed, int C, ec_stream_t stream) {
    if (!w2 || !w3 || !packed) return EC_ERR_ARG;
    if (C != 256) return EC_ERR_SHAPE;
    const long n2 = (long)C * 9 * C / 8, n3 = (long)4 * C * C / 8;
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)w2,
                       (uint4*)packed, C, 9 * C);
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)w3,
                       (uint4*)packed + n2, 4 * C, C);
    EC_CHECK_LAUNCH();
    return EC_OK;
}

// y = relu(conv3(relu(conv2(c1) + b2)) + b3 + x): the second and third conv of a stride-1 Bottleneck whose planes are C = 256
// on a 14 x 14 map (CLIP-RN50 layer3.1 .. layer3.5), BatchNorm folded into (w, b).  c1 bf16 [B,14,14,C] (conv1's output),
// packed = ec_bneck_pack_weights(w2 [C][3*3*C], w3 [4C][C]), x / y bf16 [B,14,14,4C].  EC_ERR_SHAPE for any other geometry (the
// caller then runs the two convs separately).
namespace {
template <bool F1, bool WEMU = false>
int launch_bneck(const void* c1, const void* packed, const float* b1, const float* b2, const float* b3, const void* x, void* y,
                 int B, int H, int W, int C, ec_stream_t stream) {
    if (!packed || !b2 || !b3 || !x || !y || (F1 ? !b1 : !c1)) return EC_ERR_ARG;
    if (B <= 0) return EC_ERR_SHAPE;
    if (H != 14 || W != 14 || C != 256) return EC_ERR_SHAPE;
    BneckArgs a;
    a.c1 = (const uint16_t*)c1; a.w2 = (const uint16_t*)packed; a.b2 = b2;
    a.w3 = (const uint16_t*)packed + (size_t)C * 9 * C; a.b3 = b3;
    a.w1 = (const uint16_t*)packed + (size_t)C * 9 * C + (size_t)4 * C * C; a.b1 = b1;
    a.xres = (const uint16_t*)x; a.y = (uint16_t*)y; a.B = B;
    a.w2_bytes = WEMU ? (unsigned)(ec_bneck3_packed_elems(C) * 2) : (unsigned)((size_t)C * 9 * C * 2);   // (WEMU reads 2 MB of whatever follows conv2's weights)
    a.w3_bytes = (unsigned)((size_t)4 * C * C * 2);
    a.w1_bytes = (unsigned)((size_t)C * 4 * C * 2);
    a.dbg = g_bneck_dbg;
    a.dbg_mode = g_bneck_mode;
    constexpr int PITCH = 256 * 2 + 16;
    const size_t lds = (size_t)(196 + 1) * PITCH + 8 * 3 * 2048;
    auto kern = bneck23_kernel<256, 14, F1, WEMU>;
    static std::atomic<uint64_t> attr_done{0};
    if (auto attr_g_ = ec_attr_needed(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(512), lds, (hipStream_t)stream, a);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
}  // namespace

extern "C" int ec_bneck_conv23_bf16(const void* c1, const void* packed, const float* b2, const float* b3,
                                    const void* x, void* y, int B, int H, int W, int C, ec_stream_t stream) {
    return launch_bneck<false>(c1, packed, nullptr, b2, b3, x, y, B, H, W, C, stream);
}

// The WHOLE Bottleneck in one launch: y = relu(conv3(relu(conv2(relu(conv1(x) + b1)) + b2)) + b3 + x).  packed =
// ec_bneck3_pack_weights(w1 [C][4C], w2 [C][3*3*C], w3 [4C][C]) (ec_bneck3_packed_elems(C) elements: conv2, conv3, conv1).
extern "C" size_t ec_bneck3_packed_elems(int C) { return C > 0 ? ec_bneck_packed_elems(C) + (size_t)C * 4 * C : 0; }
extern "C" int ec_bneck3_pack_weights(const void* w1, const void* w2, const void* w3, void* packed, int C, ec_stream_t stream) {
    if (!w1) return EC_ERR_ARG;
    const int rc = ec_bneck_pack_weights(w2, w3, packed, C, stream);
    if (rc != EC_OK) return rc;
    const long n1 = (long)C * 4 * C / 8;
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)w1,
                       (uint4*)((uint16_t*)packed + ec_bneck_packed_elems(C)), C, 4 * C);
    EC_CHECK_LAUNCH();
    return EC_OK;
}
extern "C" int ec_bneck_conv123_bf16(const void* x, const void* packed, const float* b1, const float* b2, const float* b3,
                                     void* y, int B, int H, int W, int C, ec_stream_t stream) {
    return launch_bneck<true>(nullptr, packed, b1, b2, b3, x, y, B, H, W, C, stream);
}
#ifdef EC_TOOLS   // tools-only build: n back-to-back launches from C (no Python between them: the GPU-side launch period)
extern "C" int ec_bneck_conv123_repeat(const void* x, const void* packed, const float* b1, const float* b2, const float* b3,
                                       void* y, int B, int H, int W, int C, int n, ec_stream_t stream) {
    unsigned long long* const base = g_bneck_dbg;       // (when set: launch i stamps into its own block of 64 + 4 B slots)
    for (int i = 0; i < n; ++i) {
        if (base) g_bneck_dbg = base + (size_t)i * (64 + 4 * (size_t)B);
        const int rc = launch_bneck<true>(nullptr, packed, b1, b2, b3, x, y, B, H, W, C, stream);
        if (rc != EC_OK) { g_bneck_dbg = base; return rc; }
    }
    g_bneck_dbg = base;
    return EC_OK;
}
#endif
#ifdef EC_TOOLS   // tools-only build: the Winograd feed emulation of bneck23_kernel (timing only, garbage results)
extern "C" int ec_bneck_wino_emu(const void* x, const void* packed, const float* b1, const float* b2, const float* b3,
                                 void* y, int B, int H, int W, int C, ec_stream_t stream) {
    return launch_bneck<true, true>(nullptr, packed, b1, b2, b3, x, y, B, H, W, C, stream);
}
#endif
