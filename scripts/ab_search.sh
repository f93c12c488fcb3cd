# A/B inside one call: the wideband seam's trigger search as its own launch (AMPS_RECC_BITS_KERNEL=separate, rounds 2-5) against the search
# stage inside the resolve kernel (default since round 6)
for i in 1 2 3; do for v in separate default; do
  if [ $v = separate ]; then export AMPS_RECC_BITS_KERNEL=separate; else unset AMPS_RECC_BITS_KERNEL; fi
  echo -n "$v: "
  python bench.py --steps 3000 --no-cpu-baseline --no-other-specs --no-latency --no-other-decim --secondary none ${1:-} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print(d['value'], 'ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'e2e', d['roofline']['frac_end_to_end'], d['roofline']['other_kernels_ms_per_step'], d['config']['checked'])"
done; done
