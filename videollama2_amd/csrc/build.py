"""Build libvl2hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so the .so travels to
the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libvl2hip.so")
SRC = os.path.join(HERE, "vl2_abi.hip")


def _deps():
    return [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".hip", ".inc"))] + \
           [os.path.join(os.path.dirname(PKG), "include", "vl2hip.h")]


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in _deps()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffast-math",
           "-fno-finite-math-only", SRC, "-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libvl2hip.so")
    if verbose:
        print(r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))
