import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/df-gnn_amd")
import torch, dfgnn_native
L = dfgnn_native.lib()
print("before cuda init:", [(m, nnz, L.dfgnn_preprocess_ws_bytes(m, nnz)) for m, nnz in ((950, 49000), (120490, 6288908), (700, 20000))], flush=True)
x = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
print("after cuda init:", [(m, nnz, L.dfgnn_preprocess_ws_bytes(m, nnz)) for m, nnz in ((950, 49000), (120490, 6288908), (700, 20000), (951, 48738))], flush=True)
from DFGNN.utils import synthetic as S
g = S.pattern_like(batch_size=8, seed=1).to("cuda:0")
print(g.num_nodes(), g.num_edges(), L.dfgnn_preprocess_ws_bytes(g.num_nodes(), g.num_edges()), L.dfgnn_plan_ints(g.num_nodes(), g.num_edges()))
import dfgnn_preprocess
src, dst = g.edges()
print("dtypes", src.dtype, dst.dtype, src.is_contiguous(), src.data_ptr() % 16, dst.data_ptr() % 16)
for csc in (False, True):
    try:
        out = dfgnn_preprocess.coo_to_hyper(src, dst, g.num_nodes(), csc=csc)
        print("csc", csc, "OK", [int(t.sum()) for t in out][:3])
    except Exception as e:
        print("csc", csc, "FAIL", e)
from DFGNN.layers import preprocess_Hyper_fw_bw
try:
    r = preprocess_Hyper_fw_bw(g)
    print("prep OK")
except Exception as e:
    print("prep FAIL", e)
