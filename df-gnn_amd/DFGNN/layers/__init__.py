from .AGNN import AGNNConv_forward  # noqa: F401
from .GT.gtconv_layer_forward import SparseMHA_forward  # noqa: F401
from .model import Model, choose_Inproj  # noqa: F401
from .util import (load_graphconv_layer, load_layer_AGNN, load_layer_GAT, load_layer_GT, load_prepfunc,  # noqa: F401
                   preprocess_CSR, preprocess_Hyper, preprocess_Hyper_fw_bw, preprocess_softmax)
