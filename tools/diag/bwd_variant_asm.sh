#!/bin/bash
# ISA + register use of ONE geometry of the dense backward: bwd_variant_asm.sh <variant 0|1|2> [extra flags] -> build/scratch/bwd_v<variant>.s
v=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/build/scratch
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -munsafe-fp-atomics --cuda-device-only \
  -DDFGNN_BWD_VARIANT=$v "$@" -S -o $root/build/scratch/bwd_v$v.s $root/df-gnn_amd/csrc/gt_dense.hip 2>&1 | grep -E "error" 
awk '/^_ZN5dfgnn19gt_dense_bwd_kernelILi128/{f=1} f{print} f&&/^\s*\.end_amdhsa_kernel/{exit}' $root/build/scratch/bwd_v$v.s > $root/build/scratch/bwd_v${v}_f128.s
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|name):" $root/build/scratch/bwd_v$v.s | paste - - - - | grep "bwd_kernelILi128"
