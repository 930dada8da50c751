#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_gpu_extra.py tests/test_random_structures_gpu.py tests/test_band_sub_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -8
for w in big small; do for pts in 64 128; do
echo "== $w pts=$pts"; SSFM_GRAM_STAMPS=1 SSFM_GRAM_PTS=$pts timeout 300 python scripts/prof_gram_stamps.py $w 2>&1 | grep "\[gram\]"
done; done
bash scripts/gpu_ab_env.sh SSFM_GRAM=0 SSFM_GRAM_PTS=128
