#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
for rep in 1 2; do
for cfg in "" "--sel-priority 0" "--depth 2" "--depth 4" "--resets kernel"; do
  timeout 600 python bench.py --cpu-images 0 $cfg > $OUT/bench_sw.json 2>> $OUT/bench_err.log
  python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_sw.json') if l.startswith('{')][-1]); print('%-22s'%'$cfg', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
done
