from .dilated_cnn import DilatedCNN, DilatedConvBlock
from .graph_neural_network import DeviceGraph, GATConv, GraphNeuralNetwork

__all__ = ["DilatedCNN", "DilatedConvBlock", "GraphNeuralNetwork", "GATConv", "DeviceGraph"]
