"""OCEAN (Online multi-modal Causal structure lEArNiNG) per-step model hot path
on AMD MI355X: the reference's Python API over hand-written gfx950 kernels."""
__version__ = "0.1.0"

from .configs.default_config import OCEANConfig, default_config, get_device
from .models import OCEANModel, OnlineLearner, Trainer

from ._hip import set_split_bf16, split_bf16

__all__ = ["OCEANConfig", "default_config", "get_device", "OCEANModel", "Trainer", "OnlineLearner", "set_split_bf16", "split_bf16"]
