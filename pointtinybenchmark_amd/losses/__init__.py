from .mil_loss import MILLoss  # noqa: F401
