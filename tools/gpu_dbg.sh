#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dbg; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for m in dynamic static graph; do
  timeout 600 python bench.py --mode $m --steps 1000 --warmup 20 --no-cpu-baseline --no-roofline > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  echo "$m: rc=$? $(cut -c1-200 $OUT/bench_$m.json)"; tail -2 $OUT/bench_$m.err | grep -v amdgpu.ids
done
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?"
cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err | grep -v amdgpu.ids
