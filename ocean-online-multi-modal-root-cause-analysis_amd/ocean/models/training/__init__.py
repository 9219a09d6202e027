from .online_learner import AdaptiveLearningRateScheduler, MemoryBuffer, OnlineLearner, SlidingWindow

__all__ = ["OnlineLearner", "MemoryBuffer", "SlidingWindow", "AdaptiveLearningRateScheduler"]
