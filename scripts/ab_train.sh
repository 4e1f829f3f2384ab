#!/bin/bash
# A/B of two builds of the library on ONE box: the training leg of bench.py, alternating.  usage: bash scripts/ab_train.sh <variant.so> [rounds]
V=$1; N=${2:-3}
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-stages --no-traffic --no-configs --no-runner-loop --no-bf16 2>/dev/null | python -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; t=d['train']; r=t['roofline']
print('$1', 'headline', round(d['value']), d['roofline']['frac'], 'train ms', round(t['ms_per_step'],3), 'fwd', r['forward_record']['ms_per_step'], 'bwd', r['backward']['ms_per_step'])"; }
for i in $(seq $N); do
  NERF_AMD_LIB=$V one variant
  one tree
done
