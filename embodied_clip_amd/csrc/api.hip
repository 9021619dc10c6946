// Library-level entry points of the C-ABI (include/ec_amd.h).
#include <stdlib.h>

#include "common.h"

extern "C" int ec_version(void) { return 500; }   // 0.5.0 (bump whenever a kernel or a launch plan changes: keys profiles/*traffic*.json)

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
    c.conv_big = env_int("EC_CONV_BIG", 1);
    c.conv8_min_tiles = env_int("EC_CONV8_MIN_TILES", 0);
    c.conv8_bn128 = env_int("EC_CONV8_BN128", 0);
    c.conv8_longseg = env_int("EC_CONV8_LONGSEG", 1);
    c.conv_t224 = env_int("EC_CONV_T224", 2);
    c.conv_t64 = env_int("EC_CONV_T64", 150);
    c.conv_ring = env_int("EC_CONV_RING", 1);
    c.conv_regw = env_int("EC_CONV_REGW", 1);
    c.rn50_fuse = env_int("EC_RN50_FUSE", 1);
    c.rn50_bneck = env_int("EC_RN50_BNECK", 128);
    c.rn50_bneck3 = env_int("EC_RN50_BNECK3", 1);
    c.rn50_img3 = env_int("EC_RN50_IMG3", 1);
    c.rn50_dscat = env_int("EC_RN50_DSCAT", 1);
    c.gemm_no_x3 = env_int("EC_GEMM_NO_X3", 0);
    c.gemm_bwd3 = env_int("EC_GEMM_BWD3", 0);
    c.act_split = env_int("EC_ACT_SPLIT", 1);
    c.tail_fused = env_int("EC_TAIL_FUSED", 1);
    c.gru_fused = env_int("EC_GRU_FUSED", 2);
    c.c1_pingpong = env_int("EC_C1_PINGPONG", 1);
    c.dw1_tr = env_int("EC_DW1_TR", 1);
    c.wih_perm = env_int("EC_WIH_PERM", 1);
    c.dw_transposed = env_int("EC_DW_TRANSPOSED", 1);
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
    // EVERY field selects kernels: all of them key the profiles under profiles/ (ec_rn50_plan_hash / ec_vit_plan_hash)
    mix(c.conv_narrow); mix(c.conv_rowsn); mix(c.conv_big); mix(c.conv8_min_tiles); mix(c.conv8_bn128); mix(c.conv8_longseg);
    mix(c.conv_t224); mix(c.conv_t64); mix(c.conv_ring); mix(c.conv_regw); mix(c.rn50_fuse); mix(c.rn50_bneck); mix(c.rn50_bneck3);
    mix(c.rn50_img3); mix(c.rn50_dscat); mix(c.gemm_no_x3); mix(c.gemm_bwd3); mix(c.act_split); mix(c.tail_fused); mix(c.gru_fused); mix(c.c1_pingpong);
    mix(c.dw1_tr); mix(c.wih_perm); mix(c.dw_transposed);
    return x;
}
