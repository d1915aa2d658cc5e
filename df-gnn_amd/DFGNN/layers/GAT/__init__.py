from .gatconv_layer import GATConvDGL  # noqa: F401
from .gatconv_layer_fused import GATConv_dgNN, GATConv_hyper, GATConv_softmax  # noqa: F401
from .gatconv_layer_softmax_gm import GATConv_softmax_gm  # noqa: F401
from .gatconv_layer_tiling import GATConv_tiling  # noqa: F401
