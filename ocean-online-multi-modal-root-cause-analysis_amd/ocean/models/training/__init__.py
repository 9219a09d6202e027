from .online_learner import AdaptiveLearningRateScheduler, MemoryBuffer, OnlineLearner, SlidingWindow
from .streaming_handler import DataStream, MemoryManager, StreamingHandler, StreamProcessor
from .trainer import EarlyStopping, LearningRateScheduler, LossFunction, Trainer

__all__ = ["OnlineLearner", "MemoryBuffer", "SlidingWindow", "AdaptiveLearningRateScheduler",
           "StreamingHandler", "StreamProcessor", "DataStream", "MemoryManager",
           "Trainer", "LossFunction", "EarlyStopping", "LearningRateScheduler"]
