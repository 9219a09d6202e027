from .data_types import BatchData, DatasetSample, MultiModalFeatures, ServiceGraph

__all__ = ["ServiceGraph", "DatasetSample", "MultiModalFeatures", "BatchData"]
