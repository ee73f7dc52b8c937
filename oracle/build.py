"""Build the oracle (test infrastructure): oracle/_build/liboracle.so from glenet_oracle.c with
gcc, and -- only when /root/reference is present (this container) -- oracle/_ref/, the
reference's own pcdet/ops/iou3d/src/iou3d_cpu.cpp compiled where it lies (see build_ref)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "glenet_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
REF_ROOT = "/root/reference"
REF_DIR = os.path.join(HERE, "_ref")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fvisibility=hidden", "-shared", "-fPIC", "-fopenmp",
               "-std=c11", "-Wall", "-o", LIB, SRC, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stderr)
    return LIB


def build_ref(force=False):
    """Compile the reference's iou3d CPU twins (no CUDA includes, pcdet/ops/iou3d/src/
    iou3d_cpu.cpp:1-4) from the sources where they lie, plus our pybind TU ref_iou3d_api.cpp,
    into oracle/_ref/iou3d_ref*.so.  Returns the module or None when the reference is absent."""
    src = os.path.join(REF_ROOT, "pcdet/ops/iou3d/src/iou3d_cpu.cpp")
    if not os.path.exists(src):
        return load_ref()
    os.makedirs(REF_DIR, exist_ok=True)
    from torch.utils.cpp_extension import load
    return load(name="iou3d_ref", sources=[src, os.path.join(HERE, "ref_iou3d_api.cpp")],
                build_directory=REF_DIR, with_cuda=False, extra_cflags=["-O2"], verbose=False)


def load_ref():
    """Import a previously built oracle/_ref module (it travels with gpurun) or None."""
    import glob
    import importlib.util
    cands = glob.glob(os.path.join(REF_DIR, "iou3d_ref*.so"))
    if not cands:
        return None
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("iou3d_ref", cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--ref" in sys.argv:
        print(build_ref())
