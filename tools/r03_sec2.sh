#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "head_kernel or bilinear or classifier or v2" > $OUT/pytest_head.log 2>&1; echo "rc=$?" >> $OUT/pytest_head.log
tail -n 3 $OUT/pytest_head.log
python tools/time_secondary.py 2>&1 | grep "bilinear" | grep -v flat
