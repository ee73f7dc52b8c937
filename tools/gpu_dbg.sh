#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dbg; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/sconv_sweep.py -1 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --no-cpu-baseline --no-train > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?"
python - <<'PY'
import json
b=json.load(open("gpurun_out/dbg/bench_full.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["frac"], b["roofline"]["all_sparse_conv"]["ms_per_step"], b["roofline"]["all_sparse_conv"]["frac"])
PY
