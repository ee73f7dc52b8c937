"""The switch table of tests/test_switches_gpu.py is complete (CPU: a grep of the package and the kernels)."""
import os

from test_switches_gpu import ROOT, SWITCHES


def test_no_other_switch_is_read_by_the_package():
    """The table above is complete: a grep of the package and the kernels for environment reads."""
    import re
    names = set()
    for base, _, files in os.walk(os.path.join(ROOT, "glenet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                names |= set(re.findall(r"(?:getenv\(|environ\.get\(|environ\[)\s*[\"'](GLX_[A-Z0-9_]+)", open(os.path.join(base, f)).read()))
    documented = {k for env, _ in SWITCHES for k in env} | {"GLX_KEEP_GRAPH_EXECS", "GLX_MAX_RETIRED_GRAPHS", "GLX_ALLOW_UNFIXED_MEMSETS",
                                                            "GLX_HIP_LIB", "GLX_HIPCC_EXTRA"}
    assert names == documented, sorted(names ^ documented)
