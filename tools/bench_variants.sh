set -x
cd $GRAFT_REPO_ROOT
for args in "--cpu-images 0" "--cpu-images 0 --depth 2" "--cpu-images 0 --depth 4" "--cpu-images 0 --source lowres --depth 3" "--cpu-images 0 --source lowres --depth 6" "--cpu-images 0 --feat-dtype f32 --depth 3" "--cpu-images 0 --feat-dtype f32 --depth 6" "--cpu-images 0 --branch ripu" "--cpu-images 0 --branch hyper" "--cpu-images 0 --channels 512 --ring 16"; do
  python bench.py $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline') or {}
        print('$args', '=>', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'feat', r.get('avg_launch_ms'), 'frac', r.get('frac'))
"
done
