#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_random_structures_gpu.py tests/test_band_sub_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -5
bash scripts/gpu_ab_env.sh SSFM_GRAM=0 SSFM_GRAM_PTS=128 SSFM_GRAM_PTS=192
