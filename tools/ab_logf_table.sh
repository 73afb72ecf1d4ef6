V=halo_amd/csrc/variants/libhalo_hip_logf_global.so
for rep in 1 2; do
for args in "" "--feat-dtype f32" "--branch ripu" "--source lowres"; do
  a=$(python bench.py --cpu-images 0 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d['roofline'] or {}).get('avg_launch_ms'))")
  b=$(HALO_LIB_PATH=$V HALO_ALLOW_STALE_LIB=1 python bench.py --cpu-images 0 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d['roofline'] or {}).get('avg_launch_ms'))")
  echo "rep $rep [$args]  LDS table: $a   | device-memory table: $b"
done
done
