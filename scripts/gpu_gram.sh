#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_band_solver_gpu.py tests/test_band_sub_gpu.py tests/test_ba_gpu.py -x -q 2>&1 | tail -4
bash scripts/gpu_ab_env.sh BASE2=1
