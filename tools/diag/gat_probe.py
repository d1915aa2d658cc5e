import os, sys, time
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")): sys.path.insert(0, p)
import torch
import fused_gatconv, fused_gtconv
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev="cuda:0"
g = S.pattern_like(batch_size=1024, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m=g.num_nodes()
ar, ac, X = S.gat_features(m, 1, 128, seed=6, device=dev)
Q,K,V = S.gt_features(m,1,128,seed=100,device=dev)
def ev(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); t1=time.perf_counter(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps, (t1-t0)*1e6/reps
with torch.no_grad():
    print("gat_forward first", ev(lambda: fused_gatconv.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0)))
    # main step a few times (as bench does), then again
    for _ in range(30):
        o, at = fused_gtconv.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        d = fused_gtconv.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, at, o)
    print("gat_forward after steps", ev(lambda: fused_gatconv.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0)))
    del o, at, d
    torch.cuda.empty_cache()
    print("gat_forward after empty_cache", ev(lambda: fused_gatconv.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0)))
    print("gat_inference_hyper", ev(lambda: fused_gatconv.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)))
    print(torch.cuda.memory_summary(abbreviated=True)[:1500])
