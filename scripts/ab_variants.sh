#!/bin/bash
# Build the working tree's library under several -D sets into scripts/variants/<name>.so (one hipcc per variant, in parallel) so
# that ONE gpurun call can time them against each other on the same box.
# usage: scripts/ab_variants.sh name1:"-DX=1 -DY=2" name2:"-DX=0" ...
set -e
mkdir -p scripts/variants
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
      $defs -Iinclude -Igr_amps_amd/csrc gr_amps_amd/csrc/amps_recc.hip -o scripts/variants/$name.so 2>/dev/null && echo "built $name ($defs)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
