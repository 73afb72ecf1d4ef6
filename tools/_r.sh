python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_lowres_source or bench_data_variants or bench_contract" 2>&1 | tail -3
python bench.py --source lowres 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity_vs_cpu'], d['parity_images_checked'], d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:80])"
