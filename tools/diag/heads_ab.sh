#!/bin/bash
# per-(range, head) single-head bodies vs the all-heads workgroup for the multi-head backward of the statistics pair
for mf in 64 0; do
  echo "DFGNN_HEADS2_MAXF=$mf"
  DFGNN_HEADS2_MAXF=$mf python tools/diag/stats_check.py --time 2>&1 | grep "'h'"
done
