cd $GRAFT_REPO_ROOT
for l in 9 6; do for v in 7 20 21 22 23 24 25 26 27 28 29; do
  echo "tlas layout $l variant $v"; timeout 120 python tools/tlas_probe.py --layout $l --variant $v --frames 3 --random 4194304 2>&1 | grep "^frame\|incoherent" | sed -n '2,3p;7p' | sed 's/.*trace/trace/'
done; done
