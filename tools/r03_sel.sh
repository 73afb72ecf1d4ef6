#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace_sel16
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_sel16 -- python3 $R/tools/prof_select16.py 2>&1 | grep -v amdgpu.ids | tail -3
python3 $R/tools/prof_select16.py summarize $OUT/trace_sel16 | tee $OUT/select16_breakdown.txt
find $OUT/trace_sel16 -name "*.csv" -size +8M -delete
