#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "head or bilinear or expmap or hypermapper or grad or classifier or v2 or region_selection_deeplab" > $OUT/pytest_head.log 2>&1; echo "rc=$?" >> $OUT/pytest_head.log
tail -n 4 $OUT/pytest_head.log
python tools/time_secondary.py 2>&1 | grep -v amdgpu.ids | grep -v "two-pass\|flat" | tee $OUT/secondary_kernels.txt
echo "--- round-2 kernels (A/B)"
HALO_EXPMAP_ONESHOT=1 HALO_BILINEAR_LDS1=1 python tools/time_secondary.py 2>&1 | grep -v amdgpu.ids | grep -v "two-pass\|flat\|hypermlr" | tee $OUT/secondary_kernels_r2.txt
