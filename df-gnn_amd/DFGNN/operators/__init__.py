from .fused_gatconv import *  # noqa: F401,F403
from .fused_gtconv import *  # noqa: F401,F403
