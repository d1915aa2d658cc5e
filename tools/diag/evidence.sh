set -x
O=/root/repo/gpurun_out/ev3; mkdir -p $O
cd /root/repo
timeout -k 10 800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python tools/bench_prep.py > $O/prep.json 2> $O/prep.err
timeout -k 10 300 python tools/train_stack.py > $O/train_stack.log 2>&1
timeout -k 10 300 python tests/tools/bench_configs.py c2 c5 c3gat gattrain --no-reddit > $O/configs.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gat -o gat -- python3 /root/repo/tools/run_kernel.py gat_train 10 > $O/prof_gat.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 /root/repo/tools/run_kernel.py gat_train 4 > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 /root/repo/tools/run_kernel.py gat_train 4 > $O/pmc_write.log 2>&1
ls -R $O | head -40
bash /root/repo/tools/diag/lowdeg_prof.sh > $O/lowdeg_kernels.txt 2>&1
python3 /root/repo/tools/diag/gt_cora.py > $O/gt_cora.txt 2>&1
