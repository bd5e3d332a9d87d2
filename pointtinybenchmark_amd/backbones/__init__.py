from .resnet import ResNet  # noqa: F401
