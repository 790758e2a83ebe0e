from .lr_schedulers import CosineWarmupLR  # noqa: F401
