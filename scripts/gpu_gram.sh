#!/bin/bash
# GPU box: the signature-group path -- its tests, the assembly by observations per point (Gram kernel against pair lists + camera sums), the grouped back substitution by K,
# and the A/B of the whole solve against SSFM_GRAM=0 at config 2 and at the configs[4] size.  Usage: bash scripts/gpu_gram.sh
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gram_groups_gpu.py tests/test_ba_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -4
timeout 600 python scripts/prof_gram_k.py 2>&1 | grep cams
timeout 600 python scripts/prof_gram_backsub.py 2>&1 | grep cams
bash scripts/gpu_ab_env.sh SSFM_GRAM=0
