#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dbg; mkdir -p $OUT; cd $R
GLX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 100 --warmup 10 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "2-rank rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/dbg/bench_2rank.json"):
    if l.startswith("{"):
        b=json.loads(l); print(b["n_gpus"], b["value"], b["ms_per_step"], b["config"]["parallelism"], b.get("cpu_baseline"), b["fwd_bwd"])
PY
tail -4 $OUT/bench_2rank.err | grep -v amdgpu.ids | cut -c1-200
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
b=json.load(open("gpurun_out/dbg/bench_default.json")); print(b["value"], b["ms_per_step"], b["cpu_baseline"], b["fwd_bwd"]["frames_per_s"])
PY
