// Library-level entry points of the C-ABI (include/ec_amd.h).
#include <stdlib.h>

#include "common.h"

extern "C" int ec_version(void) { return 600; }   // 0.6.0 (bump whenever a kernel or a launch plan changes: keys profiles/*traffic*.json)

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
    c.rn50_poolout = env_int("EC_RN50_POOLOUT", 1);
    c.gemm_no_x3 = env_int("EC_GEMM_NO_X3", 0);
    c.gemm_bwd3 = env_int("EC_GEMM_BWD3", 0);
    c.policy_fast = env_int("EC_POLICY_FAST", 1);
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
    mix(c.rn50_img3); mix(c.rn50_dscat); mix(c.rn50_poolout); mix(c.gemm_no_x3); mix(c.gemm_bwd3); mix(c.policy_fast); mix(c.act_split); mix(c.tail_fused); mix(c.gru_fused); mix(c.c1_pingpong);
    mix(c.dw1_tr); mix(c.wih_perm); mix(c.dw_transposed);
    return x;
}

// ---- stream plumbing ------------------------------------------------------------------------------------------------
// The HIP runtime binds a stream to one of its few hardware queues lazily, at the stream's first submission, and two
// streams that land on the SAME hardware queue run their launches one after the other.  Measured (tools/sync_probe.py): a
// worker whose two actor-slice streams saw their first work back to back on an idle device shared a queue in every second
// worker of a process -- 48.4 k instead of 63.1 k env-frames/s, the two encoder launches serialised -- while streams whose
// first submissions found the other streams BUSY got queues of their own.  ec_bind_streams gives every stream its first
// work while the others are busy; ec_stream_pair_overlap measures whether a pair really runs concurrently.
namespace {
__global__ void spin_kernel(unsigned long long ticks) {      // 100-MHz constant clock
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
}  // namespace

extern "C" int ec_bind_streams(ec_stream_t* streams, int n, int spin_us) {
    if (!streams || n <= 0 || spin_us <= 0 || spin_us > 100000) return EC_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return EC_ERR_LAUNCH;
    for (int round = 0; round < 2; ++round)                    // (second round: every stream busy while every other one submits)
        for (int i = 0; i < n; ++i)
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)streams[i], (unsigned long long)spin_us * 100ull);
    EC_CHECK_LAUNCH();
    return hipDeviceSynchronize() == hipSuccess ? EC_OK : EC_ERR_LAUNCH;
}

// *ratio = wall time of one spin kernel on each of the two streams, submitted together, over the time of one alone:
// ~1 = the streams run concurrently, ~2 = they are serialised (same hardware queue).  Blocking; set-up time only.
extern "C" int ec_stream_pair_overlap(ec_stream_t a, ec_stream_t b, int spin_us, float* ratio) {
    if (!ratio || spin_us <= 0 || spin_us > 100000 || a == b) return EC_ERR_ARG;
    hipStream_t sa = (hipStream_t)a, sb = (hipStream_t)b;
    hipEvent_t e[4];
    for (auto& ev : e)
        if (hipEventCreate(&ev) != hipSuccess) return EC_ERR_LAUNCH;
    const unsigned long long ticks = (unsigned long long)spin_us * 100ull;
    float best = 1e30f;
    int rc = EC_OK;
    for (int rep = 0; rep < 3 && rc == EC_OK; ++rep) {
        float alone = 0.f, both = 0.f;
        if (hipDeviceSynchronize() != hipSuccess) { rc = EC_ERR_LAUNCH; break; }
        (void)hipEventRecord(e[0], sa);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sa, ticks);
        (void)hipEventRecord(e[1], sa);
        if (hipDeviceSynchronize() != hipSuccess) { rc = EC_ERR_LAUNCH; break; }
        (void)hipEventRecord(e[2], sa);                        // common start: b waits for it, a records it
        (void)hipStreamWaitEvent(sb, e[2], 0);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sa, ticks);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sb, ticks);
        (void)hipEventRecord(e[3], sb);
        (void)hipStreamWaitEvent(sa, e[3], 0);                 // a's end event is behind both kernels
        (void)hipEventRecord(e[1], sa);
        if (hipDeviceSynchronize() != hipSuccess) { rc = EC_ERR_LAUNCH; break; }
        if (hipEventElapsedTime(&both, e[2], e[1]) != hipSuccess) { rc = EC_ERR_LAUNCH; break; }
        // (e[0] / e[1] were re-recorded: time the lone kernel again, behind everything, for the same clock state)
        (void)hipEventRecord(e[0], sa);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sa, ticks);
        (void)hipEventRecord(e[3], sa);
        if (hipDeviceSynchronize() != hipSuccess) { rc = EC_ERR_LAUNCH; break; }
        if (hipEventElapsedTime(&alone, e[0], e[3]) != hipSuccess || alone <= 0.f) { rc = EC_ERR_LAUNCH; break; }
        best = both / alone < best ? both / alone : best;
    }
    for (auto& ev : e) (void)hipEventDestroy(ev);
    if (rc == EC_OK) *ratio = best;
    return rc;
}

