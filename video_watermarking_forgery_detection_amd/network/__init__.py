from .UNet import UNet  # noqa: F401
