#!/usr/bin/env python3
"""Run-length trace of a kernel's instruction classes from a .s file: where loads, waits, barriers, MFMAs and branches sit.
usage: asm_trace.py file.s kernel_substring [start_line end_line]"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith(pat) or (l.startswith("_Z") and pat in l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
if len(sys.argv) > 4:
    start, end = int(sys.argv[3]), int(sys.argv[4])
def cls(l):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."): return None
    op = l.split()[0]
    if op.endswith(":"): return "\n" + op
    if op.startswith("global_load") or op.startswith("buffer_load"): return "L"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "S"
    if op.startswith("scratch_"): return "X"
    if op.startswith("v_mfma"): return "M"
    if op == "s_barrier": return "B"
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", l)
        return f"W{m.group(1)}" if m else None
    if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
    if op.startswith("s_cbranch") or op == "s_branch": return "j"
    if op.startswith("v_exp"): return "e"
    if op.startswith("v_"): return "v"
    return None
out, prev, cnt = [], None, 0
for i in range(start, end):
    c = cls(lines[i])
    if c is None: continue
    if c == prev and not c.startswith("\n"):
        cnt += 1
    else:
        if prev is not None: out.append(prev + (str(cnt) if cnt > 1 else ""))
        prev, cnt = c, 1
out.append(prev + (str(cnt) if cnt > 1 else ""))
print(" ".join(out))
