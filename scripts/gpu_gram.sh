#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gram_groups_gpu.py -x -q 2>&1 | tail -3
timeout 500 python scripts/prof_gram_backsub.py 2>&1 | grep cams
