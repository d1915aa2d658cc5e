from .gatconv_layer import GATConvDGL  # noqa: F401
from .gatconv_layer_fused import (GATConv_dgNN, GATConv_hyper, GATConv_hyper_ablation,  # noqa: F401
                                  GATConv_hyper_recompute, GATConv_hyper_v2, GATConv_softmax)
from .gatconv_layer_softmax_gm import GATConv_softmax_gm  # noqa: F401
from .gatconv_layer_tiling import GATConv_tiling  # noqa: F401
