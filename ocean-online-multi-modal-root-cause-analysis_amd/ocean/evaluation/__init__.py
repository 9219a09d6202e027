"""Evaluation helpers that sit directly behind the model's outputs."""
from .metrics import PerformanceMetrics

__all__ = ["PerformanceMetrics"]
