#!/bin/bash
# the timed step of bench.py with the statistics pair vs the attn_edge pair, interleaved on one box
for rep in 1 2 3; do
  for st in 1 0; do
    DFGNN_STATS=$st python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-c4 "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('DFGNN_STATS=$st', l['ms_per_step'], l['config']['training_pair'][:14], {k:v['avg_us'] for k,v in l['roofline']['all_kernels'].items()})"
  done
done
echo "host time per step (bs = 8: the GPU is idle most of the time)"
for st in 1 0 1 0; do
  DFGNN_STATS=$st python bench.py --steps 200 --warmup 20 --batch-size 8 --no-cpu-baseline --no-c4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('DFGNN_STATS=$st bs=8', l['ms_per_step'], l['config']['training_pair'][:14])"
done
