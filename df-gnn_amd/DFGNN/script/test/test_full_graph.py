"""Full-graph inference harness -- same command line as the reference's DFGNN/script/test/test_full_graph.py (:41-160):

    python DFGNN/script/test/test_full_graph.py --dim 128 --heads 1 --dataset cora --format softmax --conv gat

Ten epochs of the non-fused branch and the fused operator on the whole graph, the reference's check_correct on the first
and the last 1000 rows of the first two epochs, average times at the end.  --format all sweeps the variants this build
serves.  The graph is a synthetic stand-in of the named dataset (DFGNN/utils/datasets.py)."""
import argparse

import torch

from DFGNN.layers import Model, load_graphconv_layer, load_prepfunc
from DFGNN.script.harness import formats, report
from DFGNN.utils import check_correct, parser_argument, preprocess_dglsp
from DFGNN.utils.datasets import load_data_full_graph


def run_full_graph(args):
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    g = load_data_full_graph(args.dataset, args.data_dir).to(dev)
    deg = torch.bincount(g.edges()[0], minlength=g.num_nodes())
    print(f"# of nodes {g.num_nodes()}  # of edges {g.num_edges()}  avg. degree {deg.float().mean():.2f}  "
          f"max. degree {int(deg.max())}")
    X = g.ndata["feat"]
    results = {}
    for fmt in formats(args, "full"):
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
        results[fmt] = report(args, "full", no_fuse, fuse)
    return results


if __name__ == "__main__":
    run_full_graph(parser_argument(argparse.ArgumentParser(description="full-graph inference")))
