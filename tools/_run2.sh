R=$GRAFT_REPO_ROOT; cd $R
for pp in 12 40 64; do
  ANNCHOR_JOIN_PP_MAX=$pp timeout 600 python tools/join_tau_grid.py 1000000 2 22,36,54 2>&1 | grep -E "passes|PP_MAX" | paste - - 
done
