"""P2PHead (T/mmdet/models/point/dense_heads/p2p_head.py:18-572) -- placeholder import target; the towers,
targets and pseudo-box NMS are built in p2p_head_impl once the CPR path is measured."""
from .p2p_head_impl import P2PHead  # noqa: F401
