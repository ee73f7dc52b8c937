cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04b_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04b_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04b_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04b_smoke.log
bash tools/profile_r04.sh r04b > gpurun_out/r04b_profile.log 2>&1; echo "profile rc=$?"
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
r = b["roofline"]
print("value %.1f ms/step %.4f | %s frac %.4f us %.1f traffic %s | cfg3 train %s | inference %s" % (b["value"], b["ms_per_step"], r["kernel"], r["frac"], r["avg_launch_us"], r["traffic"], b["config3"]["train_step"]["ms_per_step"], b["inference"]["graph_ms_per_step"]))
PY
