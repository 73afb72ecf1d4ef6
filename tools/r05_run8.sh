mkdir -p gpurun_out/r05h
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "head or hypermlr or resize or bilinear or v2 or golden or fuzz or expmap or variants or gradient or autograd or region_selection" > gpurun_out/r05h/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05h/tests.log
python tools/time_head.py > gpurun_out/r05h/head_timing.txt 2>&1
python tools/time_secondary.py 2>&1 | grep -v "two-pass\|flat:" > gpurun_out/r05h/secondary.txt
for v in "f32_pm1:--feat-dtype f32" "f32_p0:--feat-dtype f32 --sel-priority 0" "hyper_auto:--branch hyper" "hyper_inline:--branch hyper --tail inline" "hyper_p0:--branch hyper --sel-priority 0" "default_p0:--sel-priority 0"; do
  name=${v%%:*}; args=${v#*:}
  python bench.py --cpu-images 0 $args > gpurun_out/r05h/bench_$name.json 2> /dev/null
done
tail -3 gpurun_out/r05h/tests.log
