cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for E in "X=1" "GLX_CONV3X3_TH=8" "GLX_CONV3X3_TH=6" "GLX_CONV3X3_BLOCKS_PER_CU=2"; do
    ms=$(env $E python bench.py --steps 60 --warmup 10 --no-extra --no-config1 --no-stages --no-cpu-baseline 2>/tmp/ab_err.txt | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])" 2>/dev/null || tail -3 /tmp/ab_err.txt)
    echo "[$E] $ms"
  done
done
