"""TEST INFRASTRUCTURE: builds tests/loopccl/libloopccl.so, the loop-back stand-in for librccl.so that lets the library's own
collective code (gr_amps_amd/csrc/recc_rccl.hip.h) run with several ranks on the one GPU of a test box.  See loopccl.cpp."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "loopccl.cpp")
LIB = os.path.join(_HERE, "libloopccl.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I%s/include" % rocm, SRC, "-o", LIB,
                           "-L%s/lib" % rocm, "-lamdhip64", "-Wl,-rpath,%s/lib" % rocm, "-lpthread"])
    return LIB
