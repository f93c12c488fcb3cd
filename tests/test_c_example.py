"""The boundary from plain C: examples/recc_abi_example.c sees include/amps_recc.h and nothing else of the library, compiles as strict
C99 with gcc (no HIP header, no C++), links against the in-tree library and chains the two reference blocks of the path through the ABI
(gr::amps::recc::work -> amps_recc_push_symbols, recc_decode::bursts_message -> amps_recc_decode_bursts, lib/recc_impl.cc:93-145,
lib/recc_decode_impl.cc:81-169).  Without a GPU it must say so and leave with 77 -- there is no CPU path behind the ABI; on the MI355X
it must decode the MIN it sent."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "recc_abi_example.c")
LIBDIR = os.path.join(ROOT, "gr_amps_amd")


def _build(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    from gr_amps_amd import build
    build.build_lib()
    exe = str(tmp_path / "recc_abi_example")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L" + LIBDIR, "-lamps_recc", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    return exe


def test_headers_are_strict_c99(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    for h in ("amps_recc.h", "amps_recc_numerics.h"):
        src = tmp_path / ("use_" + h.replace(".h", ".c"))
        src.write_text('#include "%s"\nint main(void) { return 0; }\n' % h)
        p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout


def test_c_example_builds_and_refuses_to_run_without_a_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is here: the run is test_c_example_decodes_what_it_sent")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 77, (p.returncode, p.stdout, p.stderr)
    assert "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_c_example_decodes_what_it_sent(gpu, tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "MIN 2065551234" in p.stdout and "published a burst on channel 0" in p.stdout and p.stdout.strip().endswith("ok"), p.stdout


@pytest.mark.gpu
def test_python_example_of_the_wideband_seam(gpu):
    """examples/decode_wideband.py: twelve mobiles on random channels of the band, 50 ppm off the bit clock, the stream pushed in ragged
    blocks at the library's default decimation -- every burst back with the MIN that was sent (the script's own exit code)"""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "decode_wideband.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "12 bursts sent, 12 decoded" in p.stdout and "MISMATCH" not in p.stdout
