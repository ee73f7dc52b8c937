cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-config1 --no-cpu-baseline --no-extra > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_$tag.json").read().strip().splitlines()[-1])
    s=d["stages_ms"]
    print("$tag", d["value"], d["ms_per_step"], "head", s["BEV backbone + anchor head fwd"], "roi_bwd_end", s["backward: RoI head (RoI stream)"], "bev_bwd_end", s["backward: BEV backbone"], "bwd", s["backward"])
except Exception as e:
    print("$tag failed", e)
PY
}
run base A=1
run wgroi GLX_WGRAD_DURING_ROI=1
run wgroi_q3 GLX_WGRAD_DURING_ROI=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run scales GLX_ROI_SCALE_STREAMS=1
run base2 A=1
