from .default_config import (DataConfig, EvaluationConfig, ModelConfig, OCEANConfig, SystemConfig, TrainingConfig,
                             default_config, get_device)

__all__ = ["OCEANConfig", "ModelConfig", "DataConfig", "TrainingConfig", "EvaluationConfig", "SystemConfig",
           "default_config", "get_device"]
