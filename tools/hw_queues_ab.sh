#!/bin/bash
# GPU_MAX_HW_QUEUES A/B of the bench lines and of RegionSelection's driver on one box (round 6: the value-binned selector changed
# the picture of profiles/archive/r03_hw_queues.txt / r04_hw_queues.txt).  usage: bash tools/hw_queues_ab.sh
for rep in 1 2; do
for q in 2 4 8; do
for args in "" "--feat-dtype f32" "--branch ripu" "--source lowres" "--branch hyper"; do
  a=$(HALO_BENCH_HW_QUEUES=$q python bench.py --cpu-images 0 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d['roofline'] or {}).get('avg_launch_ms'))")
  echo "rep $rep GPU_MAX_HW_QUEUES=$q bench.py $args: $a"
done
done
done
for q in 2 4 8; do
  echo "=== GPU_MAX_HW_QUEUES=$q time_region_selection.py (no backbone)"
  HALO_RS_HW_QUEUES=$q HALO_RS_REPEATS=5 timeout 300 python tools/time_region_selection.py 2>&1 | grep "stand-in none\|host floor" | cut -c1-170
done
