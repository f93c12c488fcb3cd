# one rank, real librccl: the library's distributed push (header all-gather + host poll + ncclBroadcast / scatter + all-gather into one of
# two receive buffers) against the plain push of the same block -- VERDICT r05 item 4: what does the per-push exchange cost a step?
export AMPS_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
for mode in bands broadcast_abi scatter_allgather_abi; do
  python bench.py --gpus 1 --steps 3000 --warmup 5 --no-cpu-baseline --no-other-specs --no-other-decim --secondary none --dist $mode 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-22s ms_per_step %.4f  kernel_ms %.4f  value %.1f  collective %s' % (d['dist'], d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d.get('collective')))"
done; done
