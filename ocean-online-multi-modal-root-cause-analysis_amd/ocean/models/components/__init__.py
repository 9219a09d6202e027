from .dilated_cnn import DilatedCNN, DilatedConvBlock

__all__ = ["DilatedCNN", "DilatedConvBlock"]
