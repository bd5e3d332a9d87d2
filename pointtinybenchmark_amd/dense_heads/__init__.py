from .cpr_head import CPRHead  # noqa: F401
from .p2p_head import P2PHead  # noqa: F401
