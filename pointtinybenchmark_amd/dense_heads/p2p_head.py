"""P2PHead (T/mmdet/models/point/dense_heads/p2p_head.py:18-572)."""
from .p2p_head_impl import P2PHead  # noqa: F401
