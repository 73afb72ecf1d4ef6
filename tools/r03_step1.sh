#!/bin/bash
# round 3, first GPU call: new pool-side tests, fused tail A/B, bench with the three state-restore variants
set -x
OUT=gpurun_out/r03
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_pool.py -x -q > $OUT/pytest_pool.log 2>&1; echo "pool rc=$?" >> $OUT/pytest_pool.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or bench or golden" > $OUT/pytest_tail.log 2>&1; echo "tail rc=$?" >> $OUT/pytest_tail.log
for r in kernel fills undo kernel fills undo; do
  timeout 600 python bench.py --cpu-images 0 --resets $r >> $OUT/bench_resets_$r.json 2>> $OUT/bench_err.log
done
HALO_NO_FUSE_TAIL=1 timeout 600 python bench.py --cpu-images 0 --resets fills >> $OUT/bench_round2_tail.json 2>> $OUT/bench_err.log
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench_err.log
tail -3 $OUT/pytest_pool.log $OUT/pytest_tail.log
cat $OUT/bench_resets_*.json $OUT/bench_round2_tail.json $OUT/bench_default.json | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['state_resets'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step'] - d['roofline']['avg_launch_ms'], 3))
"
