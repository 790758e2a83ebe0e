"""Model registry: same contract as bubbleformer/models/_api.py:5-40 (decorator registration, lower-cased lookup,
KeyError listing the available models, ValueError on duplicates)."""
from typing import Any, Callable, List, Optional, TypeVar

import torch.nn as nn

M = TypeVar("M", bound=nn.Module)
MODELS = {}


def register_model(name: Optional[str] = None) -> Callable[[Callable[..., M]], Callable[..., M]]:
    def wrapper(fn: Callable[..., M]) -> Callable[..., M]:
        key = name or fn.__name__
        if key in MODELS:
            raise ValueError(f"Cannot register duplicate model ({key})")
        MODELS[key] = fn
        return fn
    return wrapper


def list_models() -> List[str]:
    print("Available models:")
    return sorted(list(MODELS.keys()))


def get_model(name: str, **config: Any) -> nn.Module:
    name = name.lower()
    try:
        fn = MODELS[name]
    except KeyError as exc:
        raise KeyError(f"Model {name} not found. Available Models: {MODELS.keys()}") from exc
    return fn(**config)
