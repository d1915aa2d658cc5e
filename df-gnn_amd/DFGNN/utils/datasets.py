"""Synthetic stand-ins for the datasets and the data loader the reference's harness scripts use
(DFGNN/utils/util.py:41-148 load_dataset_fn / load_data_full_graph -- commented out at the reference's snapshot,
SURVEY.md 9 #9 -- and dgl.dataloading.GraphDataLoader).  No dataset can be downloaded here, so a dataset name maps to
a generator with that dataset's published shape statistics (DFGNN/utils/synthetic.py); node features are random.

    dataset, inference_fn = load_dataset_fn("PATTERN", data_dir)        # batched-graph datasets
    g = load_data_full_graph("cora", data_dir)                          # one graph with g.ndata["feat"]
    for batched_g, labels in GraphDataLoader(dataset, batch_size=256): ...
"""
import os

import torch

from . import synthetic as S
from .graph import batch

_FEAT_DIM = 64


class SyntheticGraphDataset:
    """`length` member graphs of a batched-graph dataset; item i is generated on demand from seed + i."""

    def __init__(self, one_graph, length, seed=0):
        self._one, self._n, self._seed = one_graph, int(length), int(seed)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if not 0 <= i < self._n:
            raise IndexError(i)
        g = self._one(self._seed + i)
        gen = torch.Generator().manual_seed(self._seed + i)
        g.ndata["feat"] = torch.randn(g.num_nodes(), _FEAT_DIM, generator=gen)
        return g, torch.zeros(1)


class GraphDataLoader:
    """Consecutive (shuffle=False) batches of a SyntheticGraphDataset as (block-diagonal graph, labels)."""

    def __init__(self, dataset, batch_size=1, shuffle=False):
        if shuffle:
            raise NotImplementedError("the harness scripts use shuffle=False")
        self.dataset, self.batch_size = dataset, int(batch_size)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for b0 in range(0, len(self.dataset), self.batch_size):
            items = [self.dataset[i] for i in range(b0, min(len(self.dataset), b0 + self.batch_size))]
            yield batch([g for g, _ in items]), torch.cat([y for _, y in items])


_BATCHED = {
    # name -> one member graph from a seed
    "PATTERN": lambda seed: S.pattern_like(batch_size=1, seed=seed),
    "CLUSTER": lambda seed: S.pattern_like(batch_size=1, seed=seed, mean_nodes=117.2, std_nodes=20.0, lo=40, hi=190,
                                           mean_deg=36.0),
    "Peptides-struct": lambda seed: S.peptides_like(batch_size=1, seed=seed),
    "Peptides-func": lambda seed: S.peptides_like(batch_size=1, seed=seed),
}


def load_dataset_fn(dataset_name, data_dir=None, length=None):
    """(dataset, inference function) for a batched-graph dataset name of the reference's scripts.  `length` member
    graphs (default: DFGNN_SYNTH_GRAPHS or 2048)."""
    from .util import inference_Graph_level
    if dataset_name not in _BATCHED:
        raise ValueError(f"unknown batched-graph dataset {dataset_name}; synthetic stand-ins: {sorted(_BATCHED)}")
    n = int(length if length is not None else os.environ.get("DFGNN_SYNTH_GRAPHS", 2048))
    return SyntheticGraphDataset(_BATCHED[dataset_name], n), inference_Graph_level


def load_data_full_graph(dataset_name, data_dir=None, scale=None):
    """One full graph with random node features for a dataset name of test_full_graph.py."""
    if dataset_name == "cora":
        g = S.cora_like()
    elif dataset_name == "reddit":
        g = S.reddit_like(scale=float(scale if scale is not None else os.environ.get("DFGNN_REDDIT_SCALE", 1.0)))
    else:
        raise ValueError(f"unknown full-graph dataset {dataset_name}; synthetic stand-ins: cora, reddit")
    g.ndata["feat"] = torch.randn(g.num_nodes(), _FEAT_DIM, generator=torch.Generator().manual_seed(0))
    return g


def mkdir(path):
    os.makedirs(path, exist_ok=True)
