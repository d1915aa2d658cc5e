"""Batched-graph inference harness -- same command line as the reference's DFGNN/script/test/test_batch_graph.py
(:11-117):

    python DFGNN/script/test/test_batch_graph.py --dim 128 --heads 1 --batch-size 1024 --dataset PATTERN --format hyper --conv gt

Every batch of the loader is run through the non-fused branch and the fused operator of the chosen layer (each timed by the
layer's own benchmark(): 3 dry + 10 timed calls between device events), the first batches are compared with the
reference's check_correct, the averages are printed (and pickled with --store-result).  --format all sweeps the variants
this build serves.  The dataset is a synthetic stand-in of the named one (DFGNN/utils/datasets.py)."""
import argparse

import torch

from DFGNN.layers import Model, load_graphconv_layer, load_prepfunc
from DFGNN.script.harness import formats, report
from DFGNN.utils import parser_argument
from DFGNN.utils.datasets import GraphDataLoader, load_dataset_fn


def run_batch_graph(args):
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    dataset, inference_fn = load_dataset_fn(args.dataset, args.data_dir)
    loader = GraphDataLoader(dataset, batch_size=args.batch_size, shuffle=False)
    in_size = dataset[0][0].ndata["feat"].shape[1]
    results = {}
    for fmt in formats(args, "batch"):
        args.format = fmt
        print("format", fmt)
        model = Model(load_graphconv_layer(args), in_size, args.dim).to(dev)
        no_fuse, fuse = inference_fn(load_prepfunc(args), model, loader, dev)
        results[fmt] = report(args, "batch", no_fuse, fuse)
    return results


if __name__ == "__main__":
    run_batch_graph(parser_argument(argparse.ArgumentParser(description="batched-graph inference")))
