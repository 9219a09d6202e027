from .components import DilatedCNN, GraphFusionModule, GraphNeuralNetwork, MultiFactorAttention
from .ocean_model import OCEANModel
from .training import LossFunction, OnlineLearner, StreamingHandler, Trainer

__all__ = ["OCEANModel", "DilatedCNN", "GraphNeuralNetwork", "MultiFactorAttention", "GraphFusionModule",
           "Trainer", "OnlineLearner", "StreamingHandler", "LossFunction"]
