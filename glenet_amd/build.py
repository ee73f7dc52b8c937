"""Build libglenet_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libglenet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
FLAGS += os.environ.get("GLX_HIPCC_EXTRA", "").split()      # experiments: e.g. -DBN_THREADS=512


HOST_SRC = os.path.join(CSRC, "host", "glx_host.cpp")
HOST_LIB = os.path.join(CSRC, "libglenet_host.so")
HOST_FLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall"]


def build_host(force=False):
    """libglenet_host.so: the host-memory entry points (include/glenet_host.h), g++ only -- it must not link
    the HIP runtime (forked DataLoader workers call it)."""
    hdr = os.path.join(HERE, "..", "include", "glenet_host.h")
    if force or _stale(HOST_LIB, [HOST_SRC, hdr]):
        r = subprocess.run([os.environ.get("CXX", "g++")] + HOST_FLAGS + ["-o", HOST_LIB, HOST_SRC],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed on glx_host.cpp:\n%s\n%s" % (r.stdout, r.stderr))
    return HOST_LIB


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "glenet_hip.h"))
    objs, jobs = [], []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    build_host(force)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
