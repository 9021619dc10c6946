#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/ls; mkdir -p $O
EC_CONV8_LONGSEG=0 python tools/bench_shapes.py 2>/dev/null | grep -E "L2|L4.x conv1|sum" > $O/s0.txt
EC_CONV8_LONGSEG=1 python tools/bench_shapes.py 2>/dev/null | grep -E "L2|L4.x conv1|sum" > $O/s1.txt
paste -d'|' $O/s0.txt $O/s1.txt | cut -c1-230
EC_CONV8_LONGSEG=0 python tools/bench_shapes.py --batch 128 2>/dev/null | grep -E "L2|L4.x conv1|sum" > $O/s0.txt
EC_CONV8_LONGSEG=1 python tools/bench_shapes.py --batch 128 2>/dev/null | grep -E "L2|L4.x conv1|sum" > $O/s1.txt
paste -d'|' $O/s0.txt $O/s1.txt | cut -c1-230
