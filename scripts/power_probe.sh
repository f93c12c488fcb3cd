#!/bin/bash
# GPU box: sample clocks and power while bench.py runs a long timed region (is the sustained stream clock- or power-limited?)
python bench.py --steps ${1:-15000} --workload ${2:-wideband832} --secondary none --no-cpu-baseline --no-other-specs > /tmp/pp.json 2>/dev/null &
BP=$!
sleep ${3:-6}
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo ---
  sleep 1
done
wait $BP
tail -1 /tmp/pp.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])'
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | head -4
