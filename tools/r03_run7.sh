#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_enc.txt 2>&1
tail -2 $O/pytest_enc.txt
for r in 0 1 2; do
  export EC_CONV_RING=$r
  for B in 32 64 128 256; do python tools/bench_shapes.py --B $B > $O/shapes_b${B}_ring$r.txt 2>&1; done
  for b in 256 128 64 32; do python tools/bench_trunk.py --batch $b --iters 10 2>&1 | grep -v "plan_hash\|amdgpu"; done > $O/trunk_ring$r.txt
done
for r in 0 1; do
  export EC_CONV_RING=$r
  for a in 32 64 256; do
  python bench.py --actors $a --no-weak --no-h2d --no-cpu-baseline --no-traffic --steps 3 2>/dev/null | tail -1 > $O/bench_a${a}_ring$r.json
  done
done
unset EC_CONV_RING
for B in 32 64 128 256; do echo "== B=$B ring 0 | 1 | 2"; paste <(grep -v amdgpu $O/shapes_b${B}_ring0.txt | cut -c1-75) <(grep -v amdgpu $O/shapes_b${B}_ring1.txt | cut -c60-75) <(grep -v amdgpu $O/shapes_b${B}_ring2.txt | cut -c60-75); done
cat $O/trunk_ring0.txt $O/trunk_ring1.txt $O/trunk_ring2.txt
for r in 0 1; do for a in 32 64 256; do python -c "import json; d=json.load(open('$O/bench_a${a}_ring$r.json')); print('ring$r actors $a', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"; done; done
