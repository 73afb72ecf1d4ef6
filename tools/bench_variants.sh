#!/bin/bash
# One line per bench variant (run through gpurun from the repository root): value, ms/step, k_feat_reduce ms, fraction.
cd "${GRAFT_REPO_ROOT:-.}"
for args in "" "--depth 2" "--depth 4" "--source lowres" "--feat-dtype f32" "--branch ripu" "--branch hyper" "--channels 512 --ring 16" "--pool-images 2975"; do
  python bench.py --cpu-images 0 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline') or {}
        print('%-28s => %9.1f img/s %8.3f ms/step  feat %s ms  frac %s' % (sys.argv[1] or '(default)', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('frac')))
" "$args"
done
