#!/bin/bash
# The differential campaigns at length on the code as it stands (GPU box); seeds start where the caller says so that two campaigns of a round
# do not repeat each other.  usage: scripts/fuzz_round.sh <first_seed> <scale>   (scale 1 = 3000 wideband streams at D = 768, ...)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
S=${1:-50000}; K=${2:-1}
echo "== wideband vs CPU model, D = 768"; python scripts/fuzz_wideband_model.py $S $((3000 * K)) 768 2>&1 | tail -1
echo "== wideband vs CPU model, D = 512"; python scripts/fuzz_wideband_model.py $S $((500 * K)) 512 2>&1 | tail -1
echo "== wideband schedules (fused / unfused / tolerant), library default"; python scripts/fuzz_wideband.py $S $((40 * K)) 2>&1 | tail -1
echo "== IQ seam, device blocks"; python scripts/fuzz_parity.py $((6000 * K)) $S device 2>&1 | tail -1
echo "== IQ seam, host blocks"; python scripts/fuzz_parity.py $((1000 * K)) $S 2>&1 | tail -1
echo "== symbol seam"; python scripts/fuzz_symbols_decode.py $((200 * K)) $S 2>&1 | tail -1
