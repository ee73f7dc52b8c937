#!/bin/bash
# everything the driver runs at round end, in one gpurun call: GPU tests, smoke, default bench (+ a 2-rank plumbing run of
# bench.py over gloo on the one GPU: the N > 1 code path end to end, not a scaling number)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
GLX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 10 --warmup 2 --no-extra --no-config1 --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank gloo bench rc=$?"
python - <<'PY'
import json
b = json.loads(open("gpurun_out/final/bench.json").read().strip().splitlines()[-1])
r = b["roofline"]
print("value %.1f %s  ms/step %.4f | %s frac %.4f traffic %s | all convs %.4f ms | cpu %s | inference %s / %s ms | cfg3 train %s ms"
      % (b["value"], b["unit"], b["ms_per_step"], r["kernel"], r["frac"], r["traffic"], r["all_sparse_conv"]["ms_per_step"],
         b["cpu_baseline"]["value"], b["inference"]["eager_ms_per_step"], b["inference"]["graph_ms_per_step"],
         b["config3"]["train_step"]["ms_per_step"]))
try:
    d = json.loads(open("gpurun_out/final/bench_2rank_gloo.json").read().strip().splitlines()[-1])
    print("2 ranks over gloo on one GPU: n_gpus %d value %.1f ms/step %.3f" % (d["n_gpus"], d["value"], d["ms_per_step"]))
except Exception as e:
    print("2-rank run:", e)
PY
