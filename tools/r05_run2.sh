mkdir -p gpurun_out/r05b
for br in ripu hyper; do
  for d in gaussian late_round peaked late_round+saturated+peaked; do
    python bench.py --branch $br --data $d --cpu-images 2 > gpurun_out/r05b/bench_${br}_$d.json 2> gpurun_out/r05b/bench_${br}_$d.err
  done
done
python bench.py --feat-dtype f32 --cpu-images 2 > gpurun_out/r05b/bench_f32.json 2> gpurun_out/r05b/bench_f32.err
python bench.py --source lowres > gpurun_out/r05b/bench_lowres.json 2> gpurun_out/r05b/bench_lowres.err
