#!/bin/bash
# the XCD-contiguous chunk map of k_feat_reduce: its test, the interleaved A/B, the bench three times
OUT=gpurun_out/r03_xcd; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "chunk_map or fused_tail or golden" 2>&1 | tail -2
python tools/ab_feat_map.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_feat_map.txt
for rep in 1 2 3 4; do
for g in -1 256; do
  HALO_FEAT_XCD_GRANULE=$g python bench.py --cpu-images 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench granule $g:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['flat_read']['GB/s'])" | tee -a $OUT/bench_ab.txt
done
done
