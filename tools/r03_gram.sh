#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "gram or lowres or region_selection" > $OUT/pytest_gram.log 2>&1; echo "rc=$?" >> $OUT/pytest_gram.log
tail -n 4 $OUT/pytest_gram.log
for e in "" "HALO_GRAM_8B=1"; do
  env $e timeout 600 python bench.py --cpu-images 0 --source lowres > $OUT/bench_gram.json 2>> $OUT/bench_err.log
  python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_gram.json') if l.startswith('{')][-1]); print('$e', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['lowres_passes_ms'])"
done
