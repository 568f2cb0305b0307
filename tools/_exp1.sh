cd $GRAFT_REPO_ROOT
for v in 8 13 14 15 10 11; do
  echo "cwbvh bistro variant $v"; timeout 200 python tools/perf_probe.py --scene bistro --width 4096 --height 4096 --layouts 9 --variant $v --passes 3 2>&1 | grep layout
done
