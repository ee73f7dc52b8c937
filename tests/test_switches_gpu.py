"""Every GLX_* switch the package still reads ("remove or table-test"): the settled experiment switches of
rounds 1-5 became constants in round 6; what is left are the documented ones below -- arithmetic (the library's: README
"Arithmetic"), the structure of the recorded step (streams, staged backward, deferred sums) and the graph hygiene knobs.  Each
setting runs a small recorded training step in its own process (the switches are read at import / library load) and must
reproduce the default's loss terms: structure switches to rounding, arithmetic switches to 2e-3."""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    # (environment, relative tolerance on the loss terms)
    ({}, 0.0),
    ({"GLX_SCONV_ARITH": "fp32"}, 2e-3), ({"GLX_CONV3X3_ARITH": "bf16x3"}, 2e-3), ({"GLX_WGRAD_FORM": "1"}, 2e-3),
    ({"GLX_SCONV_WGRAD_F16": "0"}, 2e-3), ({"GLX_SCONV_WGRAD_F16": "2"}, 2e-3),
    ({"GLX_SCONV_ARITH": "fp32", "GLX_CONV3X3_ARITH": "bf16x3", "GLX_WGRAD_FORM": "1"}, 2e-3),      # bench.py's strict_arithmetic
    ({"GLX_OVERLAP_ROI": "0"}, 2e-4), ({"GLX_STAGE_CUTS": "0"}, 2e-4), ({"GLX_OVERLAP_WGRAD": "0"}, 2e-4),
    ({"GLX_OVERLAP_PLAN": "0"}, 2e-4), ({"GLX_DEFER_WGRAD_REDUCES": "0"}, 2e-4),
    ({"GLX_REUSE_GRAPH_POOL": "0"}, 2e-4), ({"GLX_AUDIT_GRAPHS": "0"}, 2e-4),
]
# not run here: GLX_KEEP_GRAPH_EXECS=0 (destroying a hipGraphExec can crash a later launch on ROCm 7.2:
# profiles/r03_graph_exec_destroy_crash.txt -- the switch exists for runtimes without the defect), GLX_MAX_RETIRED_GRAPHS (a
# warning threshold), GLX_ALLOW_UNFIXED_MEMSETS (accepts a graph finish_graph could not repair), GLX_HIP_LIB / GLX_HIPCC_EXTRA
# (which library / how it is built), GLX_DIST_BACKEND / GLX_BENCH_FORCE_DP (bench.py and tests/test_dist_*).


def _run(env):
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_switch_worker.py")], capture_output=True, text=True, env=e,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, (env, r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_every_documented_switch_reproduces_the_default_step():
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_run, [s[0] for s in SWITCHES]))
    base = results[0]["steps"]
    keys = ("loss", "loss_rpn", "rcnn_loss_cls", "rcnn_loss_reg", "rcnn_loss_corner")
    assert all(abs(base[0][k] - base[1][k]) <= 2e-4 * abs(base[0][k]) + 1e-7 for k in keys)        # lr = 0: a replay repeats itself
    for (env, tol), res in zip(SWITCHES[1:], results[1:]):
        for k in keys:
            a, b = res["steps"][0][k], base[0][k]
            assert abs(a - b) <= tol * abs(b) + 1e-6, (env, k, a, b)
        assert abs(res["grad_norm"] - results[0]["grad_norm"]) <= 5 * tol * results[0]["grad_norm"] + 1e-6, (env, res["grad_norm"])
