cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, sys, re
out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_sparse_gpu.py", "--collect-only", "-q", "-m", "gpu"], capture_output=True, text=True).stdout
ids = [l.strip() for l in out.splitlines() if "::" in l]
print(len(ids), "tests in test_sparse_gpu")
target = "tests/test_train_step_gpu.py::test_static_step_and_graph_reproduce_the_exact_shape_step"
# which single sparse test, run before the target, makes it fail?
bad = []
fns = sorted(set(i.split("[")[0] for i in ids))
for fn in fns:
    r = subprocess.run([sys.executable, "-m", "pytest", fn, target, "-x", "-q", "-m", "gpu"], capture_output=True, text=True)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "crash rc=%d" % r.returncode
    print(fn.split("::")[1][:60].ljust(62), last[:60], flush=True)
PY
