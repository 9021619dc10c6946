#!/bin/bash
# headline bench against the per-GPU actor count (strong-scaling operating points), with phase times
for a in ${ACTORS:-32 64 128 256}; do
  python bench.py --actors $a --no-weak --no-h2d --no-cpu-baseline --no-traffic --phase-times 2>/dev/null | tail -1 > /tmp/ba.json
  python -c "import json; d=json.load(open('/tmp/ba.json')); r=d['roofline']; print('actors', $a, d['value'], 'ms/iter', d['ms_per_step'], d.get('phases'), 'enc union ms', r.get('avg_step_union_ms'), 'launch ms', r.get('avg_launch_ms'))"
done
