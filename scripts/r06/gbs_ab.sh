# usage (GPU box): bash scripts/r06/gbs_ab.sh "ENV=.." ...  -- k_gram_backsub2 under environments (e.g. SSFM_GBS_SPLIT=n) at the configs[4] size (scripts/dev/scale.py, hipEvent averages)
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== [$e]"
  env CHECK=${CHECK:-0} $e python3 scripts/dev/scale.py 2>&1 | grep "obs/s\|gram_backsub\|rel cam"
done
