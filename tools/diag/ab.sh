# A/B of library builds on the C3 backward: usage ab.sh <lib.so> [<lib.so> ...]  (rocprofv3 kernel-trace averages)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/ab_$lib
  DFGNN_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$lib -o r -- python3 /root/repo/tools/run_kernel.py bwd 20 > /tmp/ab_$lib.log 2>&1
  echo "== $lib"; grep "gt_dense" /tmp/ab_$lib/r_kernel_stats.csv | awk -F'","|",|,' '{print substr($1,1,40), $(NF-6), $(NF-4)}'
done
