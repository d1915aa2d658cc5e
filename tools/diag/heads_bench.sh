#!/bin/bash
# headline step and the multi-head variants on the C3 batch: step ms, forward us, backward us
for h in ${@:-1 2 4 8}; do
  python bench.py --heads $h --steps 20 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels']; print('heads', $h, 'step', d['ms_per_step'], 'fwd', k['gt_hyper_fwd']['avg_us'], 'bwd', k['gt_bwd']['avg_us'])"
done
