#!/bin/bash
# Regenerates the evidence under gpurun_out/ for profiles/ (one gpurun call):  bash tools/profile_round.sh r02
R=${1:-r03}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
# the launch the ENGINE issues (bench.py: two slices of 128 frames, dispatch threshold 50): same plan hash as bench.py's handles
bash tools/pmc_collect.sh trunk_$R tools/bench_trunk.py --batch 128 --min-tiles 50 --iters 1 > $O/pmc_trunk.log 2>&1
H=$(grep -h plan_hash gpurun_out/pmc_trunk_$R.kt.log | tail -1 | cut -d" " -f2)
NL=$(grep -h num_ops gpurun_out/pmc_trunk_$R.kt.log | tail -1 | cut -d" " -f2)      # launches per forward (>= 128 frames: one per op)
python tools/pmc_summary.py gpurun_out/pmc_trunk_$R 0 128 $O/trunk_b128 $H 45.7 > $O/trunk_summary_tail.txt 2>&1
# ... and the single 256-frame launch (default threshold), the shape rounds 1-2 profiled: per-kernel table for continuity
bash tools/pmc_collect.sh trunk256_$R tools/bench_trunk.py --batch 256 --iters 1 > $O/pmc_trunk256.log 2>&1
H2=$(grep -h plan_hash gpurun_out/pmc_trunk256_$R.kt.log | tail -1 | cut -d" " -f2)
python tools/pmc_summary.py gpurun_out/pmc_trunk256_$R 0 256 $O/trunk_b256 $H2 45.7 > $O/trunk256_summary_tail.txt 2>&1
# the strong-scaling operating point: one 32-frame launch (the fused layer-3 ops fall back to three launches each: + 10)
bash tools/pmc_collect.sh trunk32_$R tools/bench_trunk.py --batch 32 --iters 1 > $O/pmc_trunk32.log 2>&1
H3=$(grep -h plan_hash gpurun_out/pmc_trunk32_$R.kt.log | tail -1 | cut -d" " -f2)
python tools/pmc_summary.py gpurun_out/pmc_trunk32_$R 0 32 $O/trunk_b32 $H3 45.7 > $O/trunk32_summary_tail.txt 2>&1
bash tools/pmc_collect.sh vit_$R tools/bench_vit.py --batch 128 --min-tiles 50 --iters 1 > $O/pmc_vit.log 2>&1
NV=$(python - <<PY
import csv,glob
f=glob.glob("gpurun_out/pmc_vit_$R/kt/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# launches of the last forward = everything from its patchify launch on (ec_vit_create's weight-folding launches precede the forwards)
last=max(i for i,r in enumerate(rows) if "patchify_kernel" in r["Kernel_Name"])
print(len(rows)-last)
PY
)
HV=$(grep -h plan_hash gpurun_out/pmc_vit_$R.kt.log | tail -1 | cut -d" " -f2)
python tools/pmc_summary.py gpurun_out/pmc_vit_$R $NV 128 $O/vit_b128 "$HV" 23.3 > $O/vit_summary_tail.txt 2>&1
bash tools/pmc_collect.sh upd_$R tools/bench_update.py --iters 1 > $O/pmc_upd.log 2>&1
python tools/pmc_by_name.py gpurun_out/pmc_upd_$R 2.0 > $O/update_pmc_by_kernel.txt 2>&1
rm -rf gpurun_out/pmc_trunk32_$R gpurun_out/pmc_trunk_$R gpurun_out/pmc_trunk256_$R gpurun_out/pmc_vit_$R gpurun_out/pmc_upd_$R    # raw counter CSVs: ~80 MB, summaries are kept
python - <<PY
import json
O = "$O"
t, t256 = json.load(open(f"{O}/trunk_b128_hbm_traffic.json")), json.load(open(f"{O}/trunk_b256_hbm_traffic.json"))
t["single_launch_256"] = {"plan_hash": t256["plan_hash"], "kernel_time_us": t256["kernel_time_us"], "hbm_bytes_per_launch": t256["hbm_bytes_per_launch"],
                          "mfma_busy_frac_of_busy_cus": t256["mfma_busy_frac_of_busy_cus"]}
json.dump(t, open("profiles/trunk_hbm_traffic.json", "w"), indent=1)
json.dump(json.load(open(f"{O}/vit_b128_hbm_traffic.json")), open("profiles/vit_hbm_traffic.json", "w"), indent=1)
PY
bash tools/calibrate_fetch.sh $O/fetch_calibration.json > $O/fetch_calibration.log 2>&1
python bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench.err
python bench.py --steps 2 --warmup 1 --encoder vit --no-cpu-baseline > $O/bench_vit_line.json 2> $O/bench_vit.err
python bench.py --steps 2 --warmup 1 --encoder zeroshot --no-cpu-baseline > $O/bench_zeroshot_line.json 2> $O/bench_zs.err
python bench.py --steps 1 --warmup 1 --encoder rn50x16 --no-cpu-baseline --no-h2d --no-plugin > $O/bench_rn50x16_line.json 2> $O/bench_x16.err
for A in 128 64 32; do
  python bench.py --steps 3 --warmup 1 --actors $A --no-cpu-baseline --no-h2d --no-plugin > $O/bench_${A}actors_line.json 2> $O/bench$A.err
done
python - <<PY
import json
O = "$O"
pts = {256: json.load(open(f"{O}/bench_line.json"))["value"]}
for a in (128, 64, 32):
    pts[a] = json.load(open(f"{O}/bench_{a}actors_line.json"))["value"]
proj = {"what": "measured 1-GPU env-frames/s at 256 / 128 / 64 / 32 actors per GPU (bench.py --actors N) and the strong-scaling "
                "totals they project for BASELINE's '256 actors over 1/2/4/8 GPUs' (N x per-GPU rate; communication = one 13.9 MB "
                "all-reduce per optimiser step, not included)",
        "per_gpu_rate": {str(k): v for k, v in pts.items()},
        "projected_total": {"1": pts[256], "2": 2 * pts[128], "4": 4 * pts[64], "8": 8 * pts[32]},
        "projected_speedup_over_1gpu": {"2": round(2 * pts[128] / pts[256], 2), "4": round(4 * pts[64] / pts[256], 2),
                                        "8": round(8 * pts[32] / pts[256], 2)},
        "config4_512_actors_8_gpus": 8 * pts[64]}
json.dump(proj, open(f"{O}/strong_scaling_projection.json", "w"), indent=1)
print(proj)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof_bench
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_upd -o u -- python $GRAFT_REPO_ROOT/tools/bench_update.py --iters 3 > $GRAFT_REPO_ROOT/$O/update_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
(echo "# rocprofv3 --kernel-trace --stats of tools/bench_update.py --iters 3 (4 updates of 4 epochs incl. warm-up + one 128-step rollout): non-encoder kernels, total ms / calls / avg us / min us"; python tools/stats_noconv.py $(find $O/prof_upd -name "*kernel_stats.csv" | head -1) 0.5; grep "conv_igemm8.*true>" $(find $O/prof_upd -name "*kernel_stats.csv" | head -1) | cut -c1-160; tail -1 $O/update_under_rocprof.log) > $O/update_kernel_stats.txt
rm -rf $O/prof_upd
python tools/bench_update.py --iters 3 | tail -1 > $O/update_ms.txt
python tools/bench_img3x3.py 2>&1 | grep -v amdgpu.ids > $O/img3x3_vs_conv_igemm.txt
# the exchange step at world size 1 through RCCL (communicator + kernel really run): per-call time of the 13.9 MB bucket all-reduce
python bench.py --gpus 1 --force-dist --actors 32 --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin --no-sync-actions --no-traffic > $O/bench_32actors_forcedist_line.json 2> $O/bench32fd.err
# the plugin route's phases (one full iteration, fp32 and uint8 sensor frames)
python tools/plugin_iter_phases.py 2>&1 | grep -v amdgpu.ids > $O/plugin_iteration_phases.txt
U8=1 python tools/plugin_iter_phases.py 2>&1 | grep -v amdgpu.ids >> $O/plugin_iteration_phases.txt
# the ImageNet (torchvision ResNet-50) tower: frames per second of one 128-frame launch, per-kernel time
python tools/bench_tvresnet.py 2>&1 | grep -v amdgpu.ids > $O/tvresnet_b128.txt
# band-fused layer-2 bottleneck prototype (not in the plan) against the launches it would replace
for N in 32 128; do python tools/bench_act.py --actors $((2 * N > 48 ? 2 * N : N)) 2>&1 | tail -1; done > $O/act_step_us.txt
# one env step of the engine at 32 actors per GPU, kernel by kernel
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr32 -o t -- python $GRAFT_REPO_ROOT/bench.py --actors 32 --steps 1 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin --no-sync-actions --no-traffic > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $(find $O/tr32 -name "*kernel_trace.csv" | head -1) 40 > $O/env_step_32actors.txt; rm -rf $O/tr32
# the regime the headline runs in: rocprofv3 kernel trace of bench.py itself (two slice streams) -> per-env-step UNION of the
# encoder kernels' intervals (profiles/engine_step_union.json: what bench.py's frac_profiles_concurrent is computed from)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/esu -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin --no-sync-actions --no-traffic > $GRAFT_REPO_ROOT/$O/esu_line.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/engine_step_union.py $(find /tmp/esu -name "*kernel_trace.csv" | head -1) $O/esu_line.json $O/engine_step_union.json > $O/engine_step_union.txt 2>&1
rm -rf /tmp/esu
tail -c 600 $O/bench_line.json; echo; tail -3 $O/trunk_summary_tail.txt; tail -2 $O/vit_summary_tail.txt
