"""Build libvl2hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so the .so travels to
the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).

  libvl2hip.so      the product: bf16 elements, the default path + the documented options
  libvl2hip_f16.so  the same sources on IEEE half (-DVL2_ELEM_F16): the reference's own dtype
  libvl2hip_lab.so  (build_lab(), scripts/build_lab_lib.sh) the same sources with -DVL2_LAB: the product + the experiments that were measured
                    and lost (vl2_abi.hip `kLab`), for A/B scripts and tests/test_gpu_lab.py; never loaded by the product

A library is rebuilt when the SHA-256 of its sources + flags differs from the one recorded beside it (<lib>.srchash) -- not by mtime: a fresh
checkout, a copied tree or a touched file cannot leave a stale binary behind a "does it build" check."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libvl2hip.so")
OUT_F16 = os.path.join(PKG, "libvl2hip_f16.so")
OUT_LAB = os.path.join(PKG, "libvl2hip_lab.so")
SRC = os.path.join(HERE, "vl2_abi.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffast-math", "-fno-finite-math-only"]


def _deps():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".hip", ".inc"))) + \
           [os.path.join(os.path.dirname(PKG), "include", "vl2hip.h")]


def _src_hash(extra):
    h = hashlib.sha256(" ".join(FLAGS + list(extra)).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def _stale(out, extra):
    try:
        return not os.path.exists(out) or open(out + ".srchash").read().strip() != _src_hash(extra)
    except OSError:
        return True


def _run(targets, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for out, extra in targets:
        cmd = [hipcc, *FLAGS, *extra, SRC, "-o", out]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if os.path.exists(out + ".srchash"):
            os.remove(out + ".srchash")             # a build that dies leaves no hash behind
        h = _src_hash(extra)                        # of the sources as they are when the compiler starts: an edit during the build leaves a mismatch
        procs.append((out, h, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for out, h, pr in procs:
        so, se = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(so + se)
            raise RuntimeError(f"hipcc failed building {os.path.basename(out)}")
        open(out + ".srchash", "w").write(h + "\n")
        if verbose:
            print(se)


def build(force=False, verbose=False):
    """Builds libvl2hip.so (bf16 elements) and libvl2hip_f16.so (fp16 elements) side by side (two hipcc processes)."""
    _run([(out, extra) for out, extra in ((OUT, []), (OUT_F16, ["-DVL2_ELEM_F16"])) if force or _stale(out, extra)], verbose)
    return OUT


def build_lab(force=False, verbose=False):
    """Builds libvl2hip_lab.so (bf16 elements, -DVL2_LAB)."""
    if force or _stale(OUT_LAB, ["-DVL2_LAB"]):
        _run([(OUT_LAB, ["-DVL2_LAB"])], verbose)
    return OUT_LAB


if __name__ == "__main__":
    if "--lab" in sys.argv:
        print(build_lab(force=True, verbose="-v" in sys.argv))
    else:
        print(build(force=True, verbose="-v" in sys.argv))
