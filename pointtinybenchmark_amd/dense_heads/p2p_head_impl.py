import torch.nn as nn

from ..registry import HEADS


@HEADS.register_module()
class P2PHead(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError('P2PHead: under construction')
