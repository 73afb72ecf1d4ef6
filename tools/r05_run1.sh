mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "headline_launch_shape or bench_contract or bench_data_variants" > gpurun_out/r05a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05a/tests.log
python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err
for d in late_round saturated peaked late_round+saturated+peaked; do
  python bench.py --data $d --cpu-images 4 > gpurun_out/r05a/bench_$d.json 2> gpurun_out/r05a/bench_$d.err
done
python bench.py --resets fills --cpu-images 0 > gpurun_out/r05a/bench_fills.json 2>&1
tail -3 gpurun_out/r05a/tests.log
