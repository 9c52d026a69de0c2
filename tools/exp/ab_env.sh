#!/bin/bash
# A/B of one environment switch on the headline bench, alternating on one box:  bash tools/exp/ab_env.sh VFML_FUSE_HEAD 0 1
VAR=$1; shift
for i in 1 2; do
  for v in "$@"; do
    echo "$VAR=$v"
    env $VAR=$v timeout -k 10 200 python bench.py --no-jobs --no-roofline --no-cpu-baseline --steps 40 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],2), round(d['ms_per_step'],3), round(d.get('plans',{}).get('f16x3',{}).get('value',0),2))" || exit 1
  done
done
