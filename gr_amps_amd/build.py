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


RESOURCES = os.path.join(_HERE, "kernel_resources.json")


def kernel_resources(force=False):
    """Per-kernel register / LDS / scratch figures as hipcc reports them for gfx950 (-Rpass-analysis=kernel-resource-usage on a
    device-only compile of the library's translation unit, ~40 s, cached beside the library): {demangled-ish name: {...}}.
    tests/test_cpu_kernel_resources.py holds the occupancy-critical kernels to their budgets with it."""
    import json
    import re
    srcs = _sources()
    if not force and os.path.exists(RESOURCES) and all(os.path.getmtime(s) <= os.path.getmtime(RESOURCES) for s in srcs):
        with open(RESOURCES) as f:
            return json.load(f)
    flags = [f for f in HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    cmd = [hipcc()] + flags + ["--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                               "-I" + os.path.join(_ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "amps_recc.hip"), "-o", os.devnull]
    err = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stderr
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "lds_bytes", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}
    out, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip() or m.group(1)
            cur = out.setdefault(name, {})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    with open(RESOURCES, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    return out

