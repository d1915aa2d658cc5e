from .agnn_layers import (AGNNConv_csr, AGNNConv_csr_gm, AGNNConv_forward, AGNNConv_hyper,  # noqa: F401
                          AGNNConv_softmax, AGNNConv_softmax_gm, AGNNConv_tiling, AGNNConvDGL)
