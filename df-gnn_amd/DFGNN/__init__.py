"""DFGNN -- Python surface of the MI355X-native fused attention-GNN convolution.
Same package layout as the reference (DFGNN/__init__.py: layers + operators re-exported)."""
from .layers import *  # noqa: F401,F403
from .operators import *  # noqa: F401,F403
