"""Configuration dataclasses accepted by the model, learner and trainer.

Field names and default values are those of the reference's
ocean/configs/default_config.py:9-222 (they are read by
ocean/models/ocean_model.py:40-48,64-110 and
ocean/models/training/online_learner.py:256-278), so an OCEANConfig built for
the reference configures this implementation identically.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional

import torch


@dataclass
class ModelConfig:
    dcnn_channels: Optional[List[int]] = None          # -> [64, 128, 256, 512]
    dcnn_kernel_size: int = 3
    dcnn_dilation_rates: Optional[List[int]] = None    # -> [1, 2, 4, 8]
    dcnn_dropout: float = 0.1
    gnn_hidden_dim: int = 128
    gnn_num_heads: int = 4
    gnn_num_layers: int = 2
    gnn_dropout: float = 0.1
    attention_dim: int = 128
    attention_num_heads: int = 8
    attention_dropout: float = 0.1
    fusion_projection_dim: int = 256
    fusion_temperature: float = 0.1
    hidden_dim: int = 128
    output_dim: int = 1
    num_services: int = 100
    temporal_dim: int = 128
    spatial_dim: int = 128
    log_dim: int = 768
    fusion_dim: int = 256

    def __post_init__(self):
        self.dcnn_channels = [64, 128, 256, 512] if self.dcnn_channels is None else self.dcnn_channels
        self.dcnn_dilation_rates = [1, 2, 4, 8] if self.dcnn_dilation_rates is None else self.dcnn_dilation_rates


@dataclass
class DataConfig:
    dataset_path: str = "data/datasets"
    rcaeval_path: Optional[str] = None
    lemma_rca_path: Optional[str] = None
    sequence_length: int = 100
    sliding_window_step: int = 1
    normalization_method: str = "minmax"
    log_template_extractor: str = "drain"
    bert_model_name: str = "bert-base-uncased"
    max_log_length: int = 512
    edge_threshold: float = 0.1
    max_graph_size: int = 1000
    batch_size: int = 32
    num_workers: int = 4
    train_ratio: float = 0.7
    val_ratio: float = 0.15
    test_ratio: float = 0.15


@dataclass
class TrainingConfig:
    num_epochs: int = 100
    learning_rate: float = 0.001
    weight_decay: float = 1e-5
    gradient_clip_norm: float = 1.0
    prediction_loss_weight: float = 1.0
    contrastive_loss_weight: float = 0.1
    online_learning: bool = True
    sliding_window_size: int = 1000
    update_frequency: int = 10
    optimizer: str = "adam"
    scheduler: str = "cosine"
    early_stopping_patience: int = 10
    early_stopping_delta: float = 1e-4
    save_checkpoint_every: int = 10
    checkpoint_dir: str = "checkpoints"


@dataclass
class EvaluationConfig:
    top_k_values: Optional[List[int]] = None
    compute_statistical_significance: bool = True
    significance_level: float = 0.05
    baseline_methods: Optional[List[str]] = None
    ablation_components: Optional[List[str]] = None

    def __post_init__(self):
        self.top_k_values = [1, 3, 5, 10] if self.top_k_values is None else self.top_k_values
        self.baseline_methods = ["random", "pagerank", "clustering"] if self.baseline_methods is None else self.baseline_methods
        self.ablation_components = ["dcnn", "gnn", "attention", "fusion"] if self.ablation_components is None \
            else self.ablation_components


@dataclass
class SystemConfig:
    device: str = "auto"
    gpu_memory_fraction: float = 0.8
    log_level: str = "INFO"
    log_dir: str = "logs"
    wandb_project: Optional[str] = "ocean-rca"
    wandb_entity: Optional[str] = None
    random_seed: int = 42
    deterministic: bool = True
    num_threads: int = 4
    mixed_precision: bool = False


_SECTIONS = {"model": ModelConfig, "data": DataConfig, "training": TrainingConfig,
             "evaluation": EvaluationConfig, "system": SystemConfig}


@dataclass
class OCEANConfig:
    model: Optional[ModelConfig] = None
    data: Optional[DataConfig] = None
    training: Optional[TrainingConfig] = None
    evaluation: Optional[EvaluationConfig] = None
    system: Optional[SystemConfig] = None

    def __post_init__(self):
        for name, cls in _SECTIONS.items():
            if getattr(self, name) is None:
                setattr(self, name, cls())

    def to_dict(self) -> Dict[str, Any]:
        return {name: getattr(self, name).__dict__ for name in _SECTIONS}

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "OCEANConfig":
        return cls(**{name: sec(**config_dict.get(name, {})) for name, sec in _SECTIONS.items()})


def get_device(device_config: str = "auto") -> torch.device:
    """'auto' resolves to the GPU (ROCm exposes it as 'cuda') when present."""
    if device_config != "auto":
        return torch.device(device_config)
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def default_config() -> OCEANConfig:
    return OCEANConfig()
