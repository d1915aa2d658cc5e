#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into per-kernel means per launch.
usage: pmc_summary.py <workload string> <dir> [<dir> ...]   -> JSON on stdout
A workload string of the form "... m=<m> nnz=<nnz> h=<h> f=<f> build=<id>" also fills the m / nnz / h / f / build_id keys
bench.py matches against before it quotes `roofline.traffic` from the file.
HBM traffic = FETCH_SIZE x 2 (gfx950 counts 128-byte fetches as 64, MI355X_MICROARCH.md) + WRITE_SIZE, both in KB."""
import csv, glob, json, sys
from collections import defaultdict

work, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> per-dispatch sums
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(float)                 # (dispatch, kernel, counter) -> value summed over dims
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("dfgnn::", "")
            per[(r["Dispatch_Id"], name, r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, name, ctr), v in per.items():
            acc[name][ctr].append(v)
import re
tags = {k: re.search(rf"\b{k}=(\w+)", work) for k in ("m", "nnz", "h", "f", "build")}
out = {"workload": work, **{k: int(v.group(1)) for k, v in tags.items() if v and k != "build"},
       "build_id": tags["build"].group(1) if tags["build"] else None, "note": "rocprofv3 --pmc, one pass per counter group, means per launch (first launch of each "
       "kernel dropped as warm-up); FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled for gfx950",
       "traffic": {}, "counters_mean_per_launch": {}}
for name, ctrs in acc.items():
    if not any(k in name for k in ("dense", "block")):
        continue
    means = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in ctrs.items()}
    out["counters_mean_per_launch"][name] = means
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        rd, wr = means["FETCH_SIZE"] * 1024 * 2, means["WRITE_SIZE"] * 1024
        out["traffic"][name] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr,
                                "FETCH_SIZE_KB_raw": means["FETCH_SIZE"], "WRITE_SIZE_KB_raw": means["WRITE_SIZE"]}
json.dump(out, sys.stdout, indent=1)
