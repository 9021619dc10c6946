#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
python bench.py --no-cpu-baseline --no-traffic --steps 3 2> $O/bench.err | tail -1 > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('h2d_inclusive'), d.get('plugin_path'))"
tail -5 $O/bench.err
