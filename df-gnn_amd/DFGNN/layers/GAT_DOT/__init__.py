from .dotgatconv_layers import DOTGATConv_csr, DOTGATConv_hyper, DOTGATConv_softmax, DOTGATConvDGL, DotGatConv  # noqa: F401
