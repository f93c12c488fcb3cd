"""In-tree build of the gfx950 C-ABI library (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libamps_recc.so")
# -ffp-contract=off: the float stage is specified operation by operation (include/amps_recc_numerics.h)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-fno-slp-vectorize",  # SLP packs the demod into v_pk_* + v_mov shuffles: -10 % (measured)
               "-Wall", "-Wno-unused-function"]


def _sources():
    out = []
    for d in (CSRC, os.path.join(_ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".hpp")):
                out.append(os.path.join(d, f))
    return out


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_lib(force=False, verbose=False):
    srcs = _sources()
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in srcs):
        return LIB
    cmd = [hipcc()] + HIPCC_FLAGS + ["-I" + os.path.join(_ROOT, "include"), "-I" + CSRC,
                                     os.path.join(CSRC, "amps_recc.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
