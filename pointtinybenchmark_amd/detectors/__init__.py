from .locator import BasicLocator  # noqa: F401
