// Library-level entry points of the C-ABI (include/ec_amd.h).
#include <stdlib.h>

#include "common.h"

extern "C" int ec_version(void) { return 400; }   // 0.3.0 (bump whenever a kernel or a launch plan changes: keys profiles/*traffic*.json)

extern "C" const char* ec_strerror(int code) {
    switch (code) {
        case EC_OK: return "ok";
        case EC_ERR_ARG: return "invalid argument (null pointer or bad enum)";
        case EC_ERR_SHAPE: return "unsupported shape for this kernel";
        case EC_ERR_LAUNCH: return "HIP kernel launch failed";
        case EC_ERR_WORKSPACE: return "workspace too small";
        case EC_ERR_ALLOC: return "host allocation failed";
        case EC_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

namespace {
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
EcConfig read_config() {
    EcConfig c;
    c.conv_narrow = env_int("EC_CONV_NARROW", 3);
    c.conv_rowsn = env_int("EC_CONV_ROWSN", 1);
    c.rows_dbg = env_int("EC_ROWS_DBG", 0);
    c.conv_nbuf = env_int("EC_CONV_NBUF", 0);
    c.conv_ablate = env_int("EC_CONV_ABLATE", 0);
    c.conv_wgs = env_int("EC_CONV_WGS", 768);
    c.conv_waves = env_int("EC_CONV_WAVES", 0);
    c.conv_big = env_int("EC_CONV_BIG", 1);
    c.conv8_min_tiles = env_int("EC_CONV8_MIN_TILES", 0);
    c.conv8_bn128 = env_int("EC_CONV8_BN128", 0);
    c.conv_t224 = env_int("EC_CONV_T224", 2);
    c.conv_t64 = env_int("EC_CONV_T64", 150);
    c.conv_ring = env_int("EC_CONV_RING", 1);
    c.conv_regw = env_int("EC_CONV_REGW", 1);
    c.conv_regw_wide = env_int("EC_CONV_REGW_WIDE", 0);
    c.gemm_no_x3 = env_int("EC_GEMM_NO_X3", 0);
    c.gemm_bwd3 = env_int("EC_GEMM_BWD3", 0);
    c.act_split = env_int("EC_ACT_SPLIT", 1);
    c.tail_fused = env_int("EC_TAIL_FUSED", 1);
    c.gru_fused = env_int("EC_GRU_FUSED", 2);
    c.c1_pingpong = env_int("EC_C1_PINGPONG", 1);
    c.dw1_tr = env_int("EC_DW1_TR", 1);
    c.rn50_fuse = env_int("EC_RN50_FUSE", 1);
    c.wih_perm = env_int("EC_WIH_PERM", 1);
    c.dw_transposed = env_int("EC_DW_TRANSPOSED", 1);
    c.conv8_dirb = env_int("EC_CONV8_DIRB", 0);
    c.conv8_longseg = env_int("EC_CONV8_LONGSEG", 1);
    c.conv8_lowfill = env_int("EC_CONV8_LOWFILL", 100);
    c.conv8_lowfill_k = env_int("EC_CONV8_LOWFILL_K", 1024);
    c.conv8_res128 = env_int("EC_CONV8_RES128", 1);
    c.conv_ring_ilv = env_int("EC_CONV_RING_ILV", 1);
    c.conv_ring_w8 = env_int("EC_CONV_RING_W8", 1);
    c.rn50_side = env_int("EC_RN50_SIDE", 0);
    c.rn50_bneck = env_int("EC_RN50_BNECK", 128);
    c.rn50_img3 = env_int("EC_RN50_IMG3", 1);
    c.bneck_stagger = env_int("EC_BNECK_STAGGER", 0);
    c.rn50_bneck3 = env_int("EC_RN50_BNECK3", 1);
    c.rn50_band = env_int("EC_RN50_BAND", 0);
    c.rn50_band_max = env_int("EC_RN50_BAND_MAX", 1 << 30);
    c.conv_splitk = env_int("EC_CONV_SPLITK", 0);
    c.conv_splitk_tiles = env_int("EC_CONV_SPLITK_TILES", 200);
    c.conv_splitk_target = env_int("EC_CONV_SPLITK_TARGET", 400);
    c.conv_splitk_ns = env_int("EC_CONV_SPLITK_NS", 2);
    c.conv_splitk_tile = env_int("EC_CONV_SPLITK_TILE", 128);
    return c;
}
}  // namespace

// The one place the environment is read: a function-local static, initialised once (thread-safe) at first use.
const EcConfig& ec_config() {
    static const EcConfig c = read_config();
    return c;
}

uint64_t ec_config_hash() {
    const EcConfig& c = ec_config();
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](long v) {
        for (int i = 0; i < 8; ++i) { x ^= (uint64_t)((v >> (8 * i)) & 0xff); x *= 1099511628211ull; }
    };
    mix(c.conv_narrow); mix(c.conv_rowsn); mix(c.rows_dbg); mix(c.conv_nbuf); mix(c.conv_ablate); mix(c.conv_wgs);
    mix(c.conv_waves); mix(c.conv_big); mix(c.conv8_min_tiles); mix(c.conv8_bn128); mix(c.conv_t224); mix(c.conv_t64);
    mix(c.conv_ring); mix(c.conv_regw); mix(c.conv_regw_wide); mix(c.gemm_no_x3); mix(c.act_split); mix(c.tail_fused);
    mix(c.gru_fused); mix(c.c1_pingpong); mix(c.dw1_tr); mix(c.rn50_fuse); mix(c.wih_perm); mix(c.dw_transposed); mix(c.conv8_dirb); mix(c.conv8_longseg); mix(c.conv8_lowfill); mix(c.conv8_lowfill_k); mix(c.conv8_res128);
    mix(c.conv_ring_ilv); mix(c.conv_ring_w8); mix(c.rn50_side); mix(c.rn50_bneck); mix(c.rn50_img3); mix(c.bneck_stagger); if (c.gemm_bwd3) mix(1000 + c.gemm_bwd3); mix(c.rn50_bneck3); mix(c.rn50_band); mix(c.rn50_band_max); mix(c.conv_splitk); mix(c.conv_splitk_tiles); mix(c.conv_splitk_target); mix(c.conv_splitk_ns); mix(c.conv_splitk_tile);
    return x;
}
