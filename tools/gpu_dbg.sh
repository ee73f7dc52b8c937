#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dbg; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?"
python - <<'PY'
import json
b=json.load(open("gpurun_out/dbg/bench_full.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["frac"], b["roofline"]["traffic"], b["fwd_bwd"])
PY
tail -3 $OUT/bench_full.err | grep -v amdgpu.ids
