from .components import DilatedCNN, GraphFusionModule, GraphNeuralNetwork, MultiFactorAttention
from .ocean_model import OCEANModel
from .training import OnlineLearner

__all__ = ["OCEANModel", "DilatedCNN", "GraphNeuralNetwork", "MultiFactorAttention", "GraphFusionModule", "OnlineLearner"]
