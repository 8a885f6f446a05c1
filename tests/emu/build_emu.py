"""TEST INFRASTRUCTURE ONLY: builds tests/emu/_emu_kernels.so (host CPU) from the product's kernel headers +
tests/emu/hip_emu.h.  The only source rewrite is `extern __shared__` -> `extern` (dynamic LDS array is
provided by the harness) and the one inline-asm drain `s_waitcnt vmcnt(0)` -> no-op."""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "videollama2_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OUT = os.path.join(HERE, "_emu_kernels.so")


def build(force=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    deps = srcs + [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "emu_kernels.cpp")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in deps):
        return OUT
    tmp = tempfile.mkdtemp(prefix="vl2emu_")
    try:
        for s in srcs:
            txt = open(s).read().replace("extern __shared__", "extern")
            txt = txt.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "((void)0)")
            open(os.path.join(tmp, os.path.basename(s)), "w").write(txt)
        cxx = CLANG if os.path.exists(CLANG) else "clang++"
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-everything", "-I", tmp, "-I", HERE,
               os.path.join(HERE, "emu_kernels.cpp"), "-o", OUT]
        subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
