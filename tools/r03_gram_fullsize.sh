#!/bin/bash
# the 'gram' low-res mode against the oracle on >= 32 full-size images (C = 256 embedding 160x320, logits 640x1280
# -> 1024x2048, smooth embeddings with a projected region); output -> profiles/r03_gram_fullsize.txt
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r03
cd $R
HALO_GRAM_IMAGES=${1:-32} timeout 3000 python -m pytest tests/test_gpu_parity.py -q -k "gram_mode_full_size" -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/gram_fullsize.txt
cat gpurun_out/r03/gram_fullsize.txt
