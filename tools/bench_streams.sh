#!/bin/bash
# headline bench against the number of concurrent encoder streams (slices)
for s in 1 2 3 4; do
  python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic --encoder-streams $s 2>/dev/null | tail -1 > /tmp/bs.json
  python -c "import json; d=json.load(open('/tmp/bs.json')); print('streams', $s, d['value'], d['ms_per_step'])"
done
