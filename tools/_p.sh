cd $GRAFT_REPO_ROOT
python bench.py --steps 60 --warmup 10 --no-extra --no-config1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value']); s=d['stages_ms']; print({k:v for k,v in s.items() if k!='note'})"
