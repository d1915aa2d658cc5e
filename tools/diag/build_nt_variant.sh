#!/bin/bash
# libdfgnn_nt.so = the shipped objects with gt_dense.o and gt_dense_stats_w.o recompiled under extra flags, e.g. -DDFGNN_NT_STORES=0 -DDFGNN_NT_LOADS=0
set -e
root=$(cd "$(dirname "$0")/../.." && pwd); cs=$root/df-gnn_amd/csrc; mkdir -p $root/build/variants/nt
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics"
for t in gt_dense gt_dense_stats_w; do /opt/rocm/bin/hipcc $flags "$@" -c $cs/$t.hip -o $root/build/variants/nt/$t.o & done; wait
objs=""
for o in $cs/*.o; do b=$(basename $o .o); if [ "$b" = gt_dense ] || [ "$b" = gt_dense_stats_w ]; then objs="$objs $root/build/variants/nt/$b.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic $objs -o $root/df-gnn_amd/libdfgnn_nt.so
echo built
