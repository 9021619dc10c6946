set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_trunk.sh r3 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r3 54 > gpurun_out/pmc_r3_summary.txt
tail -1 gpurun_out/pmc_r3_summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r3 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_r3.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/bench_r3.log | cut -c1-1500
ls gpurun_out/prof_bench_r3/* | head
