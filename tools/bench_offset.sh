#!/bin/bash
# experiment: headline bench against a start offset of the second slice's act/encode chain (torch.cuda._sleep cycles)
for c in 0 50000 100000 200000 400000 800000; do
  EC_SLICE_OFFSET_CYCLES=$c python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > /tmp/bs.json
  python -c "import json; d=json.load(open('/tmp/bs.json')); print('offset_cycles', $c, d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
