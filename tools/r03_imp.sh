#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "impurity or golden or helper or fused_tail or histogram or vs_oracle or range" > $OUT/pytest_imp.log 2>&1; echo "rc=$?" >> $OUT/pytest_imp.log
tail -n 3 $OUT/pytest_imp.log
for br in ripu hyper; do
timeout 600 python bench.py --cpu-images 0 --branch $br > $OUT/bench_$br.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_$br.json') if l.startswith('{')][-1]); print('$br', d['value'], d['ms_per_step'])"
done
