"""Build of the host-side C++ mirror of gr::amps::recc / recc_decode / recc_fused (g++, links the C ABI)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_ROOT = os.path.dirname(_PKG)
BLOCKS_LIB = os.path.join(_PKG, "libgnuradio-amps-mi355x.so")
RECCTEST = os.path.join(_PKG, "recctest")


def build_host(force=False):
    srcs = [os.path.join(_HERE, "lib", f) for f in ("recc_impl.cc", "recc_decode_impl.cc", "recc_fused_impl.cc", "recc_bank_impl.cc", "recc_wideband_impl.cc")]
    hdrs = [os.path.join(_HERE, "lib", "recc_impl.h"), os.path.join(_HERE, "lib", "recc_decode_impl.h"),
            os.path.join(_HERE, "gr_min", "gnuradio_min.h"), os.path.join(_ROOT, "include", "amps_recc.h")]
    app = os.path.join(_HERE, "apps", "recctest.cc")
    deps = srcs + hdrs + [app]
    if not force and all(os.path.exists(p) for p in (BLOCKS_LIB, RECCTEST)) and \
            all(os.path.getmtime(d) <= min(os.path.getmtime(BLOCKS_LIB), os.path.getmtime(RECCTEST)) for d in deps):
        return BLOCKS_LIB, RECCTEST
    inc = ["-I" + os.path.join(_HERE, "include"), "-I" + os.path.join(_HERE, "gr_min"), "-I" + os.path.join(_ROOT, "include")]
    common = ["g++", "-std=c++17", "-O2", "-fPIC", "-Wall"] + inc
    subprocess.check_call(common + ["-shared", "-o", BLOCKS_LIB] + srcs + ["-L" + _PKG, "-lamps_recc", "-Wl,-rpath,$ORIGIN"])
    subprocess.check_call(common + ["-o", RECCTEST, app, "-L" + _PKG, "-lgnuradio-amps-mi355x", "-lamps_recc", "-Wl,-rpath,$ORIGIN"])
    return BLOCKS_LIB, RECCTEST
