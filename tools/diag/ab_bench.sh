#!/bin/bash
# A/B of library builds on the headline step: usage ab_bench.sh <lib file name under df-gnn_amd/> ...   (bench.py
# per-kernel device-event averages; DFGNN_LIB is taken relative to df-gnn_amd/)
for lib in "$@"; do
  DFGNN_LIB=$(basename $lib) python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels']; print('$lib', 'step', d['ms_per_step'], 'fwd', k['gt_hyper_fwd']['avg_us'], 'bwd', k['gt_bwd']['avg_us'], 'gat', d['secondary']['gat_train'])"
done
