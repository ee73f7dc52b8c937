cd $GRAFT_REPO_ROOT
bash tools/profile_step.sh r05q > /dev/null 2>&1
head -40 gpurun_out/r05q/train_step_kernels.md | cut -c1-120
python bench.py --steps 60 --warmup 10 --no-extra --no-config1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value']); print(json.dumps(d['stages_ms'],indent=0)[:1500])"
