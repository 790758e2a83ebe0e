from .dataset import BubbleForecast, DeviceClipStore  # noqa: F401
