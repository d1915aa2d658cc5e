# Round-3 evidence run (on the MI355X box): everything under profiles/r03_* comes from gpurun_out/ev_r03 of this script.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev_r03; mkdir -p $O
cd $R
BID=$(python3 -c "import sys; sys.path.insert(0,'df-gnn_amd'); import dfgnn_native as n; print(n.build_id())")
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for h in 2 4 8; do timeout -k 10 200 python bench.py --heads $h --no-cpu-baseline --no-c4 > $O/bench_heads$h.json 2>/dev/null; done
for h in 2 4 8; do DFGNN_STATS=0 timeout -k 10 200 python bench.py --heads $h --no-cpu-baseline --no-c4 > $O/bench_heads${h}_attn_pair.json 2>/dev/null; done
DFGNN_STATS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-c4 > $O/bench_stats_pair_h1.json 2>/dev/null
DFGNN_RANKED=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-c4 > $O/bench_csr_order_pair_h1.json 2>/dev/null
timeout -k 10 300 python tests/tools/bench_configs.py c2 c5 c3gat > $O/configs.jsonl 2>&1
timeout -k 10 300 python tools/train_stack.py > $O/train_stack.jsonl 2>&1
timeout -k 10 200 python tools/diag/class_bench_stats.py > $O/class_bench.txt 2>&1
timeout -k 10 200 python tools/diag/class_bench_stats.py --heads 8 >> $O/class_bench.txt 2>&1
timeout -k 10 300 python tools/shard_scaling.py > $O/shard_scaling.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c4 > $O/bench_profiled.json 2> $O/prof_bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 $R/tools/run_kernel.py c4 6 > $O/prof_c4.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/run_kernel.py pairs 4 > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/run_kernel.py pairs 4 > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds -o p -- python3 $R/tools/run_kernel.py pairs 4 > $O/pmc_lds.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python3 $R/tools/run_kernel.py pairs 4 > $O/pmc_mfma.log 2>&1
cd $R
python3 tools/pmc_summary.py "C3 PATTERN-like bs=1024 seed=1: m=120490 nnz=6288908 h=1 f=128 build=$BID" $O/pmc_fetch $O/pmc_write $O/pmc_lds $O/pmc_mfma > $O/pmc_dense_kernels.json
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/prof_c4 -name "*kernel_stats.csv" -exec cp {} $O/c4_kernel_stats.csv \;
rm -rf $O/prof_bench $O/prof_c4 $O/pmc_fetch $O/pmc_write $O/pmc_lds $O/pmc_mfma
ls -la $O
