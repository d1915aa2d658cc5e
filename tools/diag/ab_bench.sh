#!/bin/bash
# A/B of library builds on the headline step: usage ab_bench.sh <lib.so> ...   (bench.py per-kernel device-event averages)
for lib in "$@"; do
  DFGNN_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels']; print('$lib', 'step', d['ms_per_step'], 'fwd', k['gt_hyper_fwd']['avg_us'], 'bwd', k['gt_bwd']['avg_us'])"
done
