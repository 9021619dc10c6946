"""Copies the evidence of `tools/profile_round.sh <round>` from gpurun_out/<round>/ (merged back by gpurun) into profiles/ and
assembles profiles/{trunk,vit}_hbm_traffic.json (the files bench.py reads):  python tools/collect_profiles.py r03"""
import json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
O = f"gpurun_out/{R}"
for f in ("bench_line.json", "bench_vit_line.json", "bench_zeroshot_line.json", "bench_rn50x16_line.json", "bench_128actors_line.json",
          "bench_64actors_line.json", "bench_32actors_line.json", "bench_kernel_stats.csv", "fetch_calibration.json",
          "strong_scaling_projection.json", "trunk_b128_per_kernel.txt", "trunk_b256_per_kernel.txt",
          "update_kernel_stats.txt", "update_pmc_by_kernel.txt", "update_ms.txt", "vit_b128_per_kernel.txt",
          "trunk_b32_per_kernel.txt", "bneck_stamps.txt", "img3x3_vs_conv_igemm.txt", "act_step_us.txt", "env_step_32actors.txt",
          "winograd_feed_emulation.txt", "bench_32actors_forcedist_line.json", "plugin_iteration_phases.txt", "tvresnet_b128.txt", "engine_step_union.txt"):
    if os.path.exists(f"{O}/{f}"):
        shutil.copy(f"{O}/{f}", f"profiles/{R}_{f}")
    else:
        print("missing", f)
t, t256 = json.load(open(f"{O}/trunk_b128_hbm_traffic.json")), json.load(open(f"{O}/trunk_b256_hbm_traffic.json"))
t["single_launch_256"] = {k: t256[k] for k in ("plan_hash", "kernel_time_us", "hbm_bytes_per_launch", "mfma_busy_frac_of_busy_cus")}
json.dump(t, open("profiles/trunk_hbm_traffic.json", "w"), indent=1)
json.dump(json.load(open(f"{O}/vit_b128_hbm_traffic.json")), open("profiles/vit_hbm_traffic.json", "w"), indent=1)
if os.path.exists(f"{O}/engine_step_union.json"):
    shutil.copy(f"{O}/engine_step_union.json", "profiles/engine_step_union.json")
print(json.dumps(t)[:600])
