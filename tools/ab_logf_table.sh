#!/bin/bash
# A/B on one MI355X box: the logarithm's 2 KB table read from an LDS copy (the default build) against a -DHALO_LOGF_LDS=0 build that
# gathers it from device memory through the vector cache (halo_amd/csrc/variants/libhalo_hip_logf_global.so, loaded through
# HALO_LIB_PATH).  Same values either way (the GPU suite passes with both); prints images/s, ms per step and the feature kernel's
# average launch per bench line.  Run from the repository root: bash tools/ab_logf_table.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
V=halo_amd/csrc/variants/libhalo_hip_logf_global.so
# the variant must be a build of the CURRENT sources (same ABI): fail loudly instead of printing an empty column
HALO_LIB_PATH=$V HALO_ALLOW_STALE_LIB=1 python -c "from halo_amd import _lib; _lib.lib()" || { echo "the variant library does not load (rebuild it: see tools/collect_profiles.sh)"; exit 1; }
for rep in 1 2; do
for args in "" "--feat-dtype f32" "--branch ripu" "--source lowres"; do
  a=$(python bench.py --cpu-images 0 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d['roofline'] or {}).get('avg_launch_ms'))")
  b=$(HALO_LIB_PATH=$V HALO_ALLOW_STALE_LIB=1 python bench.py --cpu-images 0 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d['roofline'] or {}).get('avg_launch_ms'))")
  echo "rep $rep [$args]  LDS table: $a   | device-memory table: $b"
done
done
