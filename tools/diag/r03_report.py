#!/usr/bin/env python3
"""Regenerate the round-3 table block of DESIGN.md section 3.5 (between the r03-table markers) from profiles/r03_*, and print
the figures the other documents quote.  usage: python3 tools/diag/r03_report.py [--write]"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles") + "/"
L = lambda f: json.loads(open(P + f).read().strip().splitlines()[-1])
b, st, cs, pr = L("r03_bench.json"), L("r03_bench_stats_pair_h1.json"), L("r03_bench_csr_order_pair_h1.json"), L("r03_bench_profiled.json")
rk, sk, ck = b["roofline"]["all_kernels"], st["roofline"]["all_kernels"], cs["roofline"]["all_kernels"]
w, v = b["secondary"]["weighted_edges"], b["arithmetic"]["f32_valu_step"]
pm = json.load(open(P + "r03_pmc_dense_kernels.json"))["traffic"]
T = lambda k: "%.0f" % (pm[k]["total_bytes"] / 1e6)
H = lambda h, sfx="": "%.3f" % L("r03_bench_heads%d%s.json" % (h, sfx))["ms_per_step"]
HG = lambda h, sfx="": "%.3f" % L("r03_bench_heads%d%s.json" % (h, sfx))["secondary"]["step_as_hipgraph_ms"]
ks = {}
for r in csv.DictReader(open(P + "r03_bench_kernel_stats.csv")):
    m = re.search(r"dfgnn::(gt_dense\w+)", r["Name"])
    if m:
        ks[m.group(1) + ("<true>" if ", true>" in r["Name"] else "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
c4 = {re.search(r"dfgnn::(gat_chunk\w+)", r["Name"]).group(1): float(r["AverageNs"]) / 1e6
      for r in csv.DictReader(open(P + "r03_c4_kernel_stats.csv")) if "gat_chunk" in r["Name"]}
block = f'''| C3 step, same box | fwd (µs) | bwd (µs) | step (ms) | edges/s | algorithmic MB (fwd / bwd) | fraction of 8 TB/s (fwd / bwd) |
|---|---|---|---|---|---|---|
| `attn_edge` pair in rank order (§3.3h; h = 1: what the step launches) | {rk['gt_hyper_fwd_ranked']['avg_us']:.0f} | {rk['gt_bwd_ranked']['avg_us']:.0f} | **{b['ms_per_step']:.4f}** ({b['secondary']['step_as_hipgraph_ms']:.4f} as one HIP graph) | **{b['value']:.2e}** | 276.3 (its own: no `rows` / `col_ind` / `val`) / 608.9 (§8d) | {rk['gt_hyper_fwd_ranked']['frac']:.3f} / **{rk['gt_bwd_ranked']['frac']:.3f}** |
| `attn_edge` pair in CSR order (`DFGNN_RANKED=0`; round 2's kernels, and what `gt_hyper_forward` callers get) | {ck['gt_hyper_fwd']['avg_us']:.0f} | {ck['gt_bwd']['avg_us']:.0f} | {cs['ms_per_step']:.4f} | {cs['value']:.2e} | 347.9 / 608.9 (§8d) | {ck['gt_hyper_fwd']['frac']:.3f} / {ck['gt_bwd']['frac']:.3f} |
| statistics pair at h = 1 (`DFGNN_STATS=1`; `secondary.row_statistics_pair`) | {sk['gt_hyper_fwd_stats']['avg_us']:.0f} | {sk['gt_bwd_stats']['avg_us']:.0f} | {st['ms_per_step']:.4f} | {st['value']:.2e} | 251.6 / 436.7 | {sk['gt_hyper_fwd_stats']['frac']:.3f} / {sk['gt_bwd_stats']['frac']:.3f} |
| weighted edges, statistics pair (`secondary.weighted_edges`) | {w['fwd_us']:.0f} | {w['bwd_us']:.0f} | — | — | + 62 MB of dense weights per launch | — |
| fp32 VALU kernels (`arithmetic.f32_valu_step`) | {v['fwd_us']:.0f} | {v['bwd_us']:.0f} | — | — | | |

| heads × features (`bench.py --heads`) | statistics pair (ms, what runs) | `attn_edge` pair, same box (ms) | round 2 (ms, its own measurement) |
|---|---|---|---|
| 2 × 64 | {H(2)} | {H(2, '_attn_pair')} | 0.305 |
| 4 × 32 | {H(4)} | {H(4, '_attn_pair')} | 0.384 |
| 8 × 16 | **{H(8)}** | {H(8, '_attn_pair')} | 0.646 |

(Replayed as HIP graphs: {HG(2)} / {HG(2, '_attn_pair')}, {HG(4)} / {HG(4, '_attn_pair')} and {HG(8)} / {HG(8, '_attn_pair')} ms.  The statistics pair also holds `8·m·h` bytes
between forward and backward instead of `4·h·nnz`: 7.7 MB against 201 MB at 8 heads.)

HBM traffic from the PMC passes of the same build (`profiles/r03_pmc_dense_kernels.json`, FETCH_SIZE × 2 + WRITE_SIZE):
rank-ordered forward {T('gt_dense_fwd_ranked_kernel')} MB (algorithmic 276), CSR-ordered forward {T('gt_dense_fwd_kernel')} MB, `gt_dense_bwd_kernel` {T('gt_dense_bwd_kernel')} MB; statistics pair fwd {T('gt_dense_fwd_stats_kernel')} MB (algorithmic 252) / bwd **{T('gt_dense_bwd_stats_kernel')} MB**
(algorithmic 437: the 1.23× is dO, read for dP and again for dV — three images do not fit next to the tile).  By its
own bytes the statistics backward runs at a *lower* fraction than the `attn_edge` one: it moves 100 MB less in about the
same time — a range's time is its image phases in series (§ below), and it has five of them instead of four.  rocprofv3
`--kernel-trace --stats` of the bench (`profiles/r03_bench_kernel_stats.csv`): `gt_dense_fwd_ranked_kernel` {ks['gt_dense_fwd_ranked_kernel'][1]:.1f} µs over {ks['gt_dense_fwd_ranked_kernel'][0]} calls, `gt_dense_bwd_kernel` {ks['gt_dense_bwd_kernel'][1]:.1f} µs over {ks['gt_dense_bwd_kernel'][0]} calls (that run's own events: {pr['roofline']['all_kernels']['gt_hyper_fwd_ranked']['avg_us']:.1f} / {pr['roofline']['all_kernels']['gt_bwd_ranked']['avg_us']:.1f} µs — profiled runs clock 4–8 % lower than plain ones: {pr['ms_per_step']:.4f} ms per step); the CSR-ordered forward {ks['gt_dense_fwd_kernel<true>'][1]:.1f} µs, the statistics kernels {ks['gt_dense_fwd_stats_kernel'][1]:.1f} / {ks['gt_dense_bwd_stats_kernel'][1]:.1f} µs, the weighted ones {ks['gt_dense_fwd_stats_w_kernel'][1]:.1f} / {ks['gt_dense_bwd_stats_w_kernel'][1]:.1f} µs.
'''
print(block)
c = b["secondary"]["c4"]
print("headline", b["ms_per_step"], "%.3e" % b["value"], "hipgraph", b["secondary"]["step_as_hipgraph_ms"], "setup_steps", b.get("setup_steps"))
print("c4", c["ms"], c["single_kernel_ms"], "%.3e" % c["edges_per_s"], c["hbm_frac"], c["gather_GBs"], c4)
print("gat_train", b["secondary"]["gat_train"], "prep", b["secondary"]["preprocess_per_batch_ms"])
print("cpu", "%.3e" % b["cpu_baseline"]["value"], "%.3e" % b["cpu_baseline"]["c1_gat_cora_f64"]["edges_per_s"], "copy", b["roofline"]["measured_copy_GBs"])
print("traffic", b["roofline"]["traffic"], b["roofline"]["traffic_source"])
for l in open(P + "r03_shard_scaling.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print("shards", d["shards"], d["ms_per_step_eager_max_over_shards"], d["ms_per_step_hipgraph_max_over_shards"])
if "--write" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    i, j = s.index("<!-- r03-table-begin -->"), s.index("<!-- r03-table-end -->")
    s = s[:i] + "<!-- r03-table-begin -->\n" + block + s[j:]
    open(p, "w").write(s)
    print("DESIGN.md rewritten")
