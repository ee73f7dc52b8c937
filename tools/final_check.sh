#!/bin/bash
# everything the driver runs at round end, in one gpurun call: GPU tests, smoke, default bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b = json.load(open("gpurun_out/final/bench.json"))
r = b["roofline"]
print("value %.1f %s  ms/step %.4f | %s frac %.4f traffic %s | all convs %.4f ms | cpu %s | inference %s / %s ms"
      % (b["value"], b["unit"], b["ms_per_step"], r["kernel"], r["frac"], r["traffic"], r["all_sparse_conv"]["ms_per_step"],
         b["cpu_baseline"]["value"], b["inference"]["eager_ms_per_step"], b["inference"]["graph_ms_per_step"]))
PY
