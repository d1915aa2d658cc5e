from .gtconv_layer import SparseMHA  # noqa: F401
from .gtconv_layer_csr_gm import SparseMHA_CSR_GM  # noqa: F401
from .gtconv_layer_forward import SparseMHA_forward, SparseMHA_forward_timing  # noqa: F401
from .gtconv_layer_fused import SparseMHA_CSR, SparseMHA_hyper, SparseMHA_softmax  # noqa: F401
from .gtconv_layer_softmax_gm import SparseMHA_softmax_gm  # noqa: F401
from .gtconv_layer_tiling import SparseMHA_tiling  # noqa: F401
