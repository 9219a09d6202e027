from .dilated_cnn import DilatedCNN, DilatedConvBlock
from .graph_fusion import ContrastiveLearningModule, GraphFusionModule, InfoNCELoss, ProjectionHead
from .graph_neural_network import DeviceGraph, GATConv, GraphNeuralNetwork
from .multi_factor_attention import (CrossModalAttention, MultiFactorAttention, MultiHeadAttention,
                                     ScaledDotProductAttention)

__all__ = ["DilatedCNN", "DilatedConvBlock", "GraphNeuralNetwork", "GATConv", "DeviceGraph",
           "MultiFactorAttention", "CrossModalAttention", "ScaledDotProductAttention", "MultiHeadAttention",
           "GraphFusionModule", "ContrastiveLearningModule", "ProjectionHead", "InfoNCELoss"]
