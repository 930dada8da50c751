#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_gpu_extra.py tests/test_gram_groups_gpu.py tests/test_random_structures_gpu.py tests/test_band_sub_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -8
timeout 600 python scripts/prof_gram_k.py 2>&1 | grep cams
bash scripts/gpu_ab_env.sh SSFM_GRAM_BACKSUB=0
