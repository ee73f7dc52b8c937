cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv2d_gpu.py -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_detector_flow.py tests/test_train_step_gpu.py tests/test_reference_step_gpu.py tests/test_dropin_gpu.py -x -q -m gpu 2>&1 | tail -3
for on in 0 1; do GLX_CONV3X3_BN_BWD=$on timeout 600 python bench.py --no-config1 --no-cpu-baseline --no-extra > gpurun_out/b_bnbwd$on.json 2> gpurun_out/b_bnbwd$on.err; python - <<PY
import json
d=json.loads(open("gpurun_out/b_bnbwd$on.json").read().strip().splitlines()[-1])
s=d["stages_ms"]
print("bn_bwd_in_dgrad=$on", d["value"], d["ms_per_step"], "head", s["BEV backbone + anchor head fwd"], "roi_bwd_end", s["backward: RoI head (RoI stream)"], "bev_bwd_end", s["backward: BEV backbone"], "bwd", s["backward"])
PY
done
