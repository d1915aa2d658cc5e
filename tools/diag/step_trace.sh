# In-step kernel durations of the headline step (rocprofv3 kernel trace of 200 autograd steps and nothing else), for the
# rank-ordered pair, the CSR-ordered pair (DFGNN_RANKED=0) and the statistics pair (DFGNN_STATS=1).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/step_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${STEP_TRACE_VARIANTS:-ranked csr stats}; do
  unset DFGNN_BWD_REVERSE DFGNN_LIB
  case $v in ranked) export DFGNN_RANKED=1 DFGNN_STATS=auto;; csr) export DFGNN_RANKED=0 DFGNN_STATS=auto;; stats) export DFGNN_RANKED=1 DFGNN_STATS=1;; rev*) export DFGNN_RANKED=1 DFGNN_STATS=auto DFGNN_BWD_REVERSE=${v#rev};; lib_*) export DFGNN_RANKED=1 DFGNN_STATS=auto DFGNN_LIB=libdfgnn_${v#lib_}.so;; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o t -- python3 $R/tools/run_kernel.py step 200 > $O/$v.log 2>&1
  find $O/p_$v -name "*kernel_stats.csv" -exec cp {} $O/${v}_kernel_stats.csv \;
  find $O/p_$v -name "*kernel_trace.csv" -exec cp {} $O/${v}_kernel_trace.csv \;
  rm -rf $O/p_$v
done
python3 - <<'PY'
import csv, os, numpy as np
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/step_trace")
for v in os.environ.get("STEP_TRACE_VARIANTS", "ranked csr stats").split():
    rows = [r for r in csv.DictReader(open(f"{O}/{v}_kernel_trace.csv")) if "gt_dense" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[100:]                     # steady part
    st = np.array([int(r["Start_Timestamp"]) for r in rows], float); en = np.array([int(r["End_Timestamp"]) for r in rows], float)
    names = [r["Kernel_Name"].split("(")[0].split("::")[-1][:28] for r in rows]
    dur = en - st; gap = st[1:] - en[:-1]
    per = {}
    for n_, d in zip(names, dur): per.setdefault(n_, []).append(d)
    step = (en[-1] - st[0]) / (len(rows) / 2) / 1e3
    print(v, "step %.1f us" % step, {k: round(float(np.mean(x)) / 1e3, 1) for k, x in per.items()}, "mean gap %.1f us" % (gap.mean() / 1e3))
PY
