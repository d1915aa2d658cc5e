cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 /root/repo/tools/bench_prep.py > /tmp/pp.log 2>&1
tail -1 /tmp/pp.log
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pp/t_kernel_stats.csv')):
    n = r['Name']
    if ('400200' in n or 'dfgnn' in n) and float(r['AverageNs']) > 3000:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']:>4}  {n[:60]} ... {n[-150:-90]}")
PY
