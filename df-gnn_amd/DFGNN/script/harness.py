"""Bodies of the two inference harness scripts, with the reference's command lines
(DFGNN/script/test/test_batch_graph.py:11-117, test_full_graph.py:41-160):

    python DFGNN/script/test/test_batch_graph.py --dim 128 --heads 1 --batch-size 1024 --dataset PATTERN --format hyper --conv gt
    python DFGNN/script/test/test_full_graph.py  --dim 128 --heads 1 --dataset cora --format softmax --conv gat

Each format is timed the way the reference does it: the layer's own benchmark() (3 dry + 10 timed calls between device
events) on the fused operator and on the non-fused branch, the reference's check_correct verdict for the first
batches, average times at the end, optional result pickle.  Datasets are the synthetic stand-ins of
DFGNN/utils/datasets.py (nothing can be downloaded here)."""
import os
import pickle

import torch

from DFGNN.layers import Model, load_graphconv_layer, load_prepfunc
from DFGNN.utils import check_correct, preprocess_dglsp
from DFGNN.utils.datasets import GraphDataLoader, load_data_full_graph, load_dataset_fn, mkdir

_SWEEPS = {  # --format all: the variants this build serves out of the reference's sweep lists
    ("batch", "gt"): ["csr", "softmax", "hyper"],
    ("batch", "gat"): ["csr", "softmax", "hyper_v2"],
    ("batch", "agnn"): ["csr", "softmax", "hyper"],
    ("full", "gt"): ["csr", "softmax", "hyper", "tiling"],
    ("full", "gat"): ["csr", "softmax", "hyper_v2", "tiling", "hyper_recompute"],
    ("full", "agnn"): ["csr", "softmax", "hyper", "tiling"],
}


def _report(args, kind, no_fuse, fuse):
    print("----------------------Result------------------------")
    print("no-fuse average time {:.4f} ms".format(sum(no_fuse) / len(no_fuse)))
    print("fuse average time {:.4f} ms".format(sum(fuse) / len(fuse)))
    if args.store_result:
        out_dir = os.path.join(os.getcwd(), "dataset", args.dataset, args.conv)
        mkdir(out_dir)
        tag = f"{args.format}_dim{args.dim}" + (f"_bs{args.batch_size}" if kind == "batch" else "")
        path = os.path.join(out_dir, tag + "_result.pkl")
        with open(path, "wb") as fh:
            pickle.dump([no_fuse, fuse], fh)
        print("store result at", path)
    return sum(no_fuse) / len(no_fuse), sum(fuse) / len(fuse)


def _formats(args, kind):
    return _SWEEPS[(kind, args.conv)] if args.format == "all" else [args.format]


def run_batch_graph(args):
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    dataset, inference_fn = load_dataset_fn(args.dataset, args.data_dir)
    loader = GraphDataLoader(dataset, batch_size=args.batch_size, shuffle=False)
    in_size = dataset[0][0].ndata["feat"].shape[1]
    results = {}
    for fmt in _formats(args, "batch"):
        args.format = fmt
        print("format", fmt)
        model = Model(load_graphconv_layer(args), in_size, args.dim).to(dev)
        no_fuse, fuse = inference_fn(load_prepfunc(args), model, loader, dev)
        results[fmt] = _report(args, "batch", no_fuse, fuse)
    return results


def run_full_graph(args):
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    g = load_data_full_graph(args.dataset, args.data_dir).to(dev)
    deg = torch.bincount(g.edges()[0], minlength=g.num_nodes())
    print(f"# of nodes {g.num_nodes()}  # of edges {g.num_edges()}  avg. degree {deg.float().mean():.2f}  "
          f"max. degree {int(deg.max())}")
    X = g.ndata["feat"]
    results = {}
    for fmt in _formats(args, "full"):
        args.format = fmt
        print("format", fmt)
        model = Model(load_graphconv_layer(args), X.shape[1], args.dim).to(dev).eval()
        A, params = preprocess_dglsp(g), load_prepfunc(args)(g)
        no_fuse, fuse = [], []
        with torch.no_grad():
            model(A, X)                                             # warm-up
            for epoch in range(10):
                logits, t_nofuse = model(A, X)
                logits_fuse, t_fuse = model(params, X, fuse=True)
                if epoch < 2:
                    check_correct(logits[:1000], logits_fuse[:1000], params)
                    check_correct(logits[-1000:], logits_fuse[-1000:], params)
                no_fuse.append(t_nofuse)
                fuse.append(t_fuse)
                print(f"epoch {epoch} non-fused time {t_nofuse:.4f}  fused time {t_fuse:.4f}")
        results[fmt] = _report(args, "full", no_fuse, fuse)
    return results
