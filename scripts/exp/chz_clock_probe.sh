#!/bin/bash
# EXPERIMENT (round 6): shader clock and package power while the filter bank runs with its input from HBM / from the L2 / switched off
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/chz_clock_probe.txt; : > $OUT
for v in base cachedloads noloads base cachedloads noloads; do
  ( CHZ_DECIM=768 AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python scripts/bench_chz.py 30000 exact 1500 2>&1 | tail -1 > /tmp/r_$v.txt ) &
  P=$!
  : > /tmp/smi.txt
  while kill -0 $P 2>/dev/null; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import json,sys
try:
    c=next(iter(json.load(sys.stdin).values()))
    p=[float(v) for k,v in c.items() if 'Power' in k and '(W)' in k]
    s=[v for k,v in c.items() if k.startswith('sclk')]
    print(p[0] if p else 0, ''.join(ch for ch in str(s[0]) if ch.isdigit()) if s else 0)
except Exception: pass" >> /tmp/smi.txt
    sleep 0.2
  done
  echo "$v: $(cat /tmp/r_$v.txt | cut -c1-30) | $(python -c "
rows=[l.split() for l in open('/tmp/smi.txt') if l.strip()]
rows=[(float(a),float(b)) for a,b in rows if float(a)>700]
rows=rows[len(rows)//3:]
print('%d samples: %.0f W, %.0f MHz' % (len(rows), sum(r[0] for r in rows)/max(1,len(rows)), sum(r[1] for r in rows)/max(1,len(rows))))")" >> $OUT
done
cat $OUT
