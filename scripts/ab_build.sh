#!/bin/bash
# A/B build for kernel experiments: compiles the library of a git revision (default HEAD) into scripts/variants/base.so so that
# `scripts/ab_run.sh` can time it against the working tree's library inside ONE gpurun call (box-to-box spread is +-4 %).
REV=${1:-HEAD}
set -e
mkdir -p scripts/variants /tmp/ab_src
rm -rf /tmp/ab_src/*
git archive $REV include gr_amps_amd/csrc | tar -x -C /tmp/ab_src
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
    -I/tmp/ab_src/include -I/tmp/ab_src/gr_amps_amd/csrc /tmp/ab_src/gr_amps_amd/csrc/amps_recc.hip -o scripts/variants/base.so
ls -la scripts/variants/base.so
