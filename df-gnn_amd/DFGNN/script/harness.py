"""What the two inference harness scripts share (the scripts themselves hold the run loops, like the reference's:
DFGNN/script/test/test_batch_graph.py:11-117, test_full_graph.py:41-160), with the reference's command lines:

    python DFGNN/script/test/test_batch_graph.py --dim 128 --heads 1 --batch-size 1024 --dataset PATTERN --format hyper --conv gt
    python DFGNN/script/test/test_full_graph.py  --dim 128 --heads 1 --dataset cora --format softmax --conv gat

Each format is timed the way the reference does it: the layer's own benchmark() (3 dry + 10 timed calls between device
events) on the fused operator and on the non-fused branch, the reference's check_correct verdict for the first
batches, average times at the end, optional result pickle.  Datasets are the synthetic stand-ins of
DFGNN/utils/datasets.py (nothing can be downloaded here)."""
import os
import pickle

from DFGNN.utils.datasets import mkdir

SWEEPS = {  # --format all: the variants this build serves out of the reference's sweep lists
    ("batch", "gt"): ["csr", "softmax", "hyper"],
    ("batch", "gat"): ["csr", "softmax", "hyper_v2"],
    ("batch", "agnn"): ["csr", "softmax", "hyper"],
    ("full", "gt"): ["csr", "softmax", "hyper", "tiling"],
    ("full", "gat"): ["csr", "softmax", "hyper_v2", "tiling", "hyper_recompute"],
    ("full", "agnn"): ["csr", "softmax", "hyper", "tiling"],
}


def report(args, kind, no_fuse, fuse):
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


def formats(args, kind):
    return SWEEPS[(kind, args.conv)] if args.format == "all" else [args.format]


def __getattr__(name):  # run_batch_graph / run_full_graph live in the scripts; kept importable from here
    if name == "run_batch_graph":
        from DFGNN.script.test.test_batch_graph import run_batch_graph
        return run_batch_graph
    if name == "run_full_graph":
        from DFGNN.script.test.test_full_graph import run_full_graph
        return run_full_graph
    raise AttributeError(name)
