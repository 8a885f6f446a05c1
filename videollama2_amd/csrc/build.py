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


OUT_F16 = os.path.join(PKG, "libvl2hip_f16.so")      # the same sources on IEEE half (-DVL2_ELEM_F16): the reference's own dtype


def build(force=False, verbose=False):
    """Builds libvl2hip.so (bf16 elements) and libvl2hip_f16.so (fp16 elements) side by side (two hipcc processes)."""
    deps = _deps()
    todo = [(out, extra) for out, extra in ((OUT, []), (OUT_F16, ["-DVL2_ELEM_F16"]))
            if force or not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps)]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for out, extra in todo:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffast-math",
               "-fno-finite-math-only", *extra, SRC, "-o", out]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for out, pr in procs:
        so, se = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(so + se)
            raise RuntimeError(f"hipcc failed building {os.path.basename(out)}")
        if verbose:
            print(se)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))
