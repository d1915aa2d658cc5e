cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ld
rocprofv3 --kernel-trace --output-format csv -d /tmp/ld -o t -- python3 /root/repo/tools/diag/lowdeg_kernels.py > /tmp/ld.log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/ld/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'dfgnn' not in n or 'plan' in n or 'gather' in n or 'pointers' in n: continue
    key = n.split('<')[0].replace('void dfgnn::','') + ('<' + n.split('FeatCfg<')[1].split('>')[0] + '>' if 'FeatCfg<' in n else '') + (' PASS' + n.split('>, ')[1].split('>')[0] if 'gat_train_' in n else '')
    acc.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in acc.items():
    v = v[2:] if len(v) > 4 else v
    print(f"{sum(v)/len(v):8.1f} us x{len(v):3d}  {k}")
PY
