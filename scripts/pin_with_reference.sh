#!/bin/bash
# pin_with_reference.sh -- turns "parity: partial" into reference-pinned parity.  It CANNOT run in this repository's build image (no
# GNU Radio, no IT++, no Boost, no network); it is the recipe for whoever has them:
#
#   needs   GNU Radio 3.7 (>= 3.7.2, with python + swig), IT++ (libitpp-dev), Boost, CppUnit, cmake;
#           a checkout of unsynchronized/gr-amps ($AMPS_REFERENCE, default /root/reference);
#           this repository (for tests/golden/pin_inputs.npz and the runner).
#   does    1. builds the reference with ITS OWN CMake, out of tree, and installs it into a scratch prefix (nothing of it is copied here);
#           2. runs scripts/pin/run_reference.py under GNU Radio's python: the reference's amps.recc / amps.recc_decode and the four GNU
#              Radio blocks of grc/recctest.grc on the committed seeded inputs -> tests/golden/reference_pins.npz (DATA: inputs'
#              outputs, no reference source);
#           3. runs tests/test_cpu_reference_pins.py, which holds oracle/ref_chain.c to that file row by row (R2, R3-R8 + replies, G1-G4).
#   then    commit tests/golden/reference_pins.npz: from that commit on the CPU suite pins the oracle to reference-produced vectors and
#           DESIGN.md section 2's "parity partial" rows G3 / G1-rotator / MMSE table / IT++ version are settled by data.
set -euo pipefail
HERE=$(cd "$(dirname "$0")/.." && pwd)
REF=${AMPS_REFERENCE:-/root/reference}
PREFIX=${AMPS_PIN_PREFIX:-/tmp/gr-amps-ref}
BUILD=${AMPS_PIN_BUILD:-/tmp/gr-amps-ref-build}
PY=${AMPS_PIN_PYTHON:-python2}
command -v gnuradio-config-info >/dev/null || { echo "GNU Radio 3.7 is not installed here: this recipe is for a machine that has it (see the header)"; exit 2; }
echo "GNU Radio $(gnuradio-config-info --version), reference at $REF"
mkdir -p "$BUILD" && cd "$BUILD"
cmake -DCMAKE_INSTALL_PREFIX="$PREFIX" -DCMAKE_BUILD_TYPE=Release "$REF"
make -j"$(nproc)"
make install
PYVER=$($PY -c 'import sys; print("%d.%d" % sys.version_info[:2])')
export PYTHONPATH="$PREFIX/lib/python$PYVER/dist-packages:$PREFIX/lib/python$PYVER/site-packages:$PREFIX/lib64/python$PYVER/site-packages:${PYTHONPATH:-}"
export LD_LIBRARY_PATH="$PREFIX/lib:$PREFIX/lib64:${LD_LIBRARY_PATH:-}"
cd "$HERE"
[ -f tests/golden/pin_inputs.npz ] || python3 scripts/pin/make_pin_inputs.py
$PY scripts/pin/run_reference.py
# (tests/golden/pin_manifest.json says what a good file looks like: arrays, dtype kinds, shapes, GNU Radio 3.7 -- the test checks it first)
python3 -m pytest tests/test_cpu_reference_pins.py -q -rs
