"""Input projection + one conv layer (reference: DFGNN/layers/model.py:29-40,
DFGNN/script/test/test_full_graph.py:20-31).  The dataset-specific encoders of the reference
(ogb AtomEncoder etc.) are replaced by a plain Linear / Embedding chosen by feature dtype."""
import torch.nn as nn


def choose_Inproj(in_size, hidden_size, integer_features=False):
    return nn.Embedding(in_size, hidden_size) if integer_features else nn.Linear(in_size, hidden_size)


class Model(nn.Module):
    def __init__(self, MHAlayer, in_size, hidden_size, integer_features=False):
        super().__init__()
        self.inproj = choose_Inproj(in_size, hidden_size, integer_features)
        self.MHA = MHAlayer

    def forward(self, params, X, fuse=False):
        h = self.inproj(X)
        if h.dim() == 3:  # embedding of several categorical columns
            h = h.sum(1)
        return self.MHA(params, h, fuse)
