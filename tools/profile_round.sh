#!/bin/bash
# Regenerates the evidence under gpurun_out/ for profiles/ (one gpurun call):  bash tools/profile_round.sh r02
R=${1:-r02}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
bash tools/pmc_collect.sh trunk_$R tools/bench_trunk.py --batch 256 --iters 1 > $O/pmc_trunk.log 2>&1
H=$(grep -h plan_hash gpurun_out/pmc_trunk_$R.kt.log | tail -1 | cut -d" " -f2)
python tools/pmc_summary.py gpurun_out/pmc_trunk_$R 50 256 $O/trunk_b256 $H 45.7 > $O/trunk_summary_tail.txt 2>&1
bash tools/pmc_collect.sh vit_$R tools/bench_vit.py --batch 256 --iters 1 > $O/pmc_vit.log 2>&1
NV=$(python - <<PY
import csv,glob
f=glob.glob("gpurun_out/pmc_vit_$R/kt/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
# launches of the last forward = total / number of forwards (warm-up + 1 timed)
print(len(rows)//2)
PY
)
python tools/pmc_summary.py gpurun_out/pmc_vit_$R $NV 256 $O/vit_b256 "" 0 > $O/vit_summary_tail.txt 2>&1
bash tools/pmc_collect.sh upd_$R tools/bench_update.py --iters 1 > $O/pmc_upd.log 2>&1
python tools/pmc_by_name.py gpurun_out/pmc_upd_$R 2.0 > $O/update_pmc_by_kernel.txt 2>&1
rm -rf gpurun_out/pmc_trunk_$R gpurun_out/pmc_vit_$R gpurun_out/pmc_upd_$R    # raw counter CSVs: ~80 MB, summaries are kept
cp $O/trunk_b256_hbm_traffic.json profiles/trunk_b256_hbm_traffic.json
cp $O/vit_b256_hbm_traffic.json profiles/vit_b256_hbm_traffic.json
python bench.py --steps 2 --warmup 1 > $O/bench_line.json 2> $O/bench.err
python bench.py --steps 2 --warmup 1 --encoder vit --no-cpu-baseline > $O/bench_vit_line.json 2> $O/bench_vit.err
python bench.py --steps 2 --warmup 1 --encoder zeroshot --no-cpu-baseline > $O/bench_zeroshot_line.json 2> $O/bench_zs.err
python bench.py --steps 2 --warmup 1 --actors 64 --no-cpu-baseline --no-h2d > $O/bench_64actors_line.json 2> $O/bench64.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof_bench
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_upd -o u -- python $GRAFT_REPO_ROOT/tools/bench_update.py --iters 3 > $GRAFT_REPO_ROOT/$O/update_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
(echo "# rocprofv3 --kernel-trace --stats of tools/bench_update.py --iters 3 (4 updates of 4 epochs incl. warm-up + one 128-step rollout): non-encoder kernels, total ms / calls / avg us / min us"; python tools/stats_noconv.py $(find $O/prof_upd -name "*kernel_stats.csv" | head -1) 0.5; grep "conv_igemm8.*true>" $(find $O/prof_upd -name "*kernel_stats.csv" | head -1) | cut -c1-160; tail -1 $O/update_under_rocprof.log) > $O/update_kernel_stats.txt
rm -rf $O/prof_upd
python tools/bench_update.py --iters 3 | tail -1 > $O/update_ms.txt
tail -c 600 $O/bench_line.json; echo; tail -3 $O/trunk_summary_tail.txt; tail -2 $O/vit_summary_tail.txt
