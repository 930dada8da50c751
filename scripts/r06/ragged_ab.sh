# usage (GPU box): bash scripts/r06/ragged_ab.sh "ENV=.." ...  -- scripts/dev/ring.py ragged14 / ragged8 under each environment (timing only)
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== [$e]"
  env CHECK=0 $e python3 scripts/dev/ring.py ragged14 ragged8 2>&1 | grep -v "^$" | cut -c1-170
done
