"""Streaming front-end of the online learner (host logic; no arithmetic).

Classes, constructor arguments, method names, statistics keys and failure
behaviour follow the reference (ocean/models/training/streaming_handler.py:
27-85 DataStream, 88-201 MemoryManager, 204-410 StreamProcessor, 413-642
StreamingHandler): bounded per-stream deques that drop when full, one daemon
worker that every `processing_interval` seconds drains up to `batch_size`
samples per active stream and hands them to OnlineLearner.process_batch in
chunks of `batch_size`, swallowing (and logging) per-batch exceptions.

Sharding over data-parallel ranks (no counterpart in the reference, which is one process): every rank runs its own
handler over the same set of streams; stream number s (in creation order) belongs to rank s mod world and only its
owner buffers its samples.  A polling cycle starts with one small all-gather in which the ranks tell each other how
many chunks they hold and of which sizes; all then run the largest chunk count, a rank that has run out of chunks
taking part with a weightless stand-in batch, so that the collectives inside the update step stay matched.

Addition (off by default): `event_driven=True` wakes the worker as soon as one
stream holds `wake_samples` samples instead of at the next polling tick, so the
delay between a window arriving and its update is the update time, not up to
`processing_interval`.  A cycle does the same work either way; the interval
stays the upper bound on the wait.
"""
from __future__ import annotations

import gc
import json
import logging
import queue
import threading
import time
import weakref
from collections import deque
from datetime import datetime
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import psutil

from ...data.data_types import DatasetSample
from .online_learner import OnlineLearner

logger = logging.getLogger(__name__)


class ShardedStepError(RuntimeError):
    """A data-parallel update step failed on this rank in the middle of a sharded cycle; the rank has left the group."""


class DataStream:
    def __init__(self, stream_id: str, buffer_size: int = 1000):
        self.stream_id, self.buffer_size = stream_id, buffer_size
        self.buffer: deque = deque(maxlen=buffer_size)
        self.is_active = False
        self.total_samples = 0
        self.dropped_samples = 0
        self._on_sample: Optional[Callable[["DataStream"], None]] = None    # set by an event-driven StreamProcessor

    def add_sample(self, sample: Any) -> bool:
        if len(self.buffer) >= self.buffer_size:
            self.dropped_samples += 1
            return False
        self.buffer.append(sample)
        self.total_samples += 1
        notify = self._on_sample
        if notify is not None:
            notify(self)
        return True

    def get_samples(self, max_samples: Optional[int] = None) -> List[Any]:
        if max_samples is None:
            out = list(self.buffer)
            self.buffer.clear()
            return out
        out = []
        while self.buffer and len(out) < max_samples:
            out.append(self.buffer.popleft())
        return out

    def get_stats(self) -> Dict[str, Any]:
        return {"stream_id": self.stream_id, "buffer_size": len(self.buffer), "max_buffer_size": self.buffer_size,
                "total_samples": self.total_samples, "dropped_samples": self.dropped_samples,
                "drop_rate": self.dropped_samples / max(1, self.total_samples), "is_active": self.is_active}


class MemoryManager:
    """Host-memory watchdog: polls psutil and asks registered objects to cleanup() above a threshold."""

    def __init__(self, max_memory_usage: float = 0.8, cleanup_threshold: float = 0.9, check_interval: float = 10.0):
        self.max_memory_usage, self.cleanup_threshold, self.check_interval = max_memory_usage, cleanup_threshold, check_interval
        self.managed_objects: List[weakref.ref] = []
        self.memory_history: deque = deque(maxlen=100)
        self.is_monitoring = False
        self.monitor_thread: Optional[threading.Thread] = None

    def register_object(self, obj: Any):
        self.managed_objects.append(weakref.ref(obj))

    def get_memory_usage(self) -> float:
        return psutil.virtual_memory().percent / 100.0

    def start_monitoring(self):
        if self.is_monitoring:
            return
        self.is_monitoring = True
        self.monitor_thread = threading.Thread(target=self._monitoring_worker, daemon=True)
        self.monitor_thread.start()

    def stop_monitoring(self):
        self.is_monitoring = False
        if self.monitor_thread:
            self.monitor_thread.join(timeout=5.0)

    def _monitoring_worker(self):
        while self.is_monitoring:
            try:
                usage = self.get_memory_usage()
                self.memory_history.append(usage)
                if usage > self.cleanup_threshold:
                    logger.warning("High memory usage: %.2f%%", usage * 100)
                    self.cleanup()
                # sleep in short slices so stop_monitoring() returns promptly
                t_end = time.time() + self.check_interval
                while self.is_monitoring and time.time() < t_end:
                    time.sleep(min(0.1, self.check_interval))
            except Exception as e:
                logger.error("Error in memory monitoring: %s", e)

    def cleanup(self):
        alive = []
        for ref in self.managed_objects:
            obj = ref()
            if obj is not None:
                if hasattr(obj, "cleanup"):
                    obj.cleanup()
                alive.append(ref)
        self.managed_objects = alive
        gc.collect()

    def get_memory_stats(self) -> Dict[str, Any]:
        stats = {"current_usage": self.get_memory_usage(), "max_usage": self.max_memory_usage,
                 "cleanup_threshold": self.cleanup_threshold, "managed_objects_count": len(self.managed_objects),
                 "is_monitoring": self.is_monitoring}
        if self.memory_history:
            h = list(self.memory_history)
            stats.update({"avg_usage": np.mean(h), "max_recent_usage": np.max(h),
                          "usage_trend": np.polyfit(range(len(h)), h, 1)[0] if len(h) > 1 else 0})
        return stats


class StreamProcessor:
    MAX_CHUNKS = 62          # chunks per rank and cycle the agreement message has room for (64 streams x 32 samples)

    def __init__(self, online_learner: OnlineLearner, max_streams: int = 10, processing_interval: float = 1.0,
                 batch_size: int = 32, event_driven: bool = False, wake_samples: Optional[int] = None,
                 data_parallel=None, service_graph=None):
        self.online_learner, self.max_streams = online_learner, max_streams
        # sharding: the ocean.parallel.DataParallel that wires the learner's model (None: single process), and the
        # service graph a rank uses for stand-in batches before it has seen a sample of its own
        self.data_parallel = data_parallel if data_parallel is not None else getattr(online_learner, "data_parallel", None)
        self.rank = self.data_parallel.rank if self.data_parallel is not None else 0
        self.world = self.data_parallel.world if self.data_parallel is not None else 1
        self.service_graph = service_graph
        self._seq_len = 0
        self._created = 0
        self.owner: Dict[str, int] = {}
        self.standin_steps = 0
        self.failed: Optional[BaseException] = None     # set when a collective update step failed on this rank
        self.processing_interval, self.batch_size = processing_interval, batch_size
        self.event_driven = bool(event_driven)
        self.wake_samples = max(1, int(wake_samples if wake_samples is not None else batch_size))
        self._wake = threading.Event()
        self.early_wakeups = 0
        self.streams: Dict[str, DataStream] = {}
        self.processing_queue: queue.PriorityQueue = queue.PriorityQueue()
        self.is_processing = False
        self.processor_thread: Optional[threading.Thread] = None
        self.processed_samples = 0
        self.processing_times: deque = deque(maxlen=1000)
        self.throughput_history: deque = deque(maxlen=100)
        self.memory_manager = MemoryManager()
        self.memory_manager.register_object(self)

    def add_stream(self, stream: DataStream) -> bool:
        if len(self.streams) >= self.max_streams:
            logger.warning("Maximum streams (%d) reached", self.max_streams)
            return False
        if stream.stream_id in self.streams:
            logger.warning("Stream %s already exists", stream.stream_id)
            return False
        self.streams[stream.stream_id] = stream
        self.owner[stream.stream_id] = self._created % self.world      # creation order is the same on every rank
        self._created += 1
        stream.is_active = True
        if self.event_driven:
            stream._on_sample = self._sample_arrived
        return True

    def _sample_arrived(self, stream: DataStream) -> None:
        # producer side (any thread): one length test, and a set() only when a full batch is waiting
        if len(stream.buffer) >= self.wake_samples:
            self._wake.set()

    def _batch_waiting(self) -> bool:
        return any(s.is_active and len(s.buffer) >= self.wake_samples for s in list(self.streams.values()))

    def owns(self, stream_id: str) -> bool:
        return self.owner.get(stream_id, 0) == self.rank

    def remove_stream(self, stream_id: str) -> bool:
        stream = self.streams.pop(stream_id, None)
        self.owner.pop(stream_id, None)
        if stream is None:
            return False
        stream.is_active = False
        stream._on_sample = None
        return True

    def start_processing(self):
        if self.is_processing:
            logger.warning("Stream processing already active")
            return
        self.is_processing = True
        self.processor_thread = threading.Thread(target=self._processing_worker, daemon=True)
        self.processor_thread.start()
        self.memory_manager.start_monitoring()

    def stop_processing(self):
        self.is_processing = False
        self._wake.set()            # an event-driven worker may be waiting on it
        if self.processor_thread:
            self.processor_thread.join(timeout=10.0)
        self.memory_manager.stop_monitoring()

    def process_once(self) -> int:
        """One polling cycle: drain every active stream and update the model.  Returns samples processed."""
        t0 = time.time()
        samples: List[DatasetSample] = []
        # sharded: a cycle's agreement message has room for MAX_CHUNKS chunks per rank, so no more than that many
        # samples leave the stream buffers (the rest stay where they are for the next cycle)
        room = self.MAX_CHUNKS * self.batch_size if self.world > 1 else None
        for stream in list(self.streams.values()):
            if stream.is_active and self.owns(stream.stream_id):
                take = self.batch_size if room is None else min(self.batch_size, room - len(samples))
                if take <= 0:
                    break
                samples.extend(stream.get_samples(take))
        if self.failed is not None:
            raise ShardedStepError("this rank left the data-parallel group after a failed update step") from self.failed
        if self.world > 1:
            done = self._process_samples_sharded(samples)
            if done == 0 and not samples:
                return 0
        elif not samples:
            return 0
        else:
            done = self._process_samples(samples)
        self.processed_samples += done
        dt = time.time() - t0
        self.processing_times.append(dt)
        if dt > 0:
            self.throughput_history.append(done / dt)
        return done

    def _processing_worker(self):
        while self.is_processing:
            try:
                t0 = time.time()
                self._wake.clear()
                self.process_once()
                remaining = self.processing_interval - (time.time() - t0)
                if self.event_driven:
                    # samples that arrived during the update may already make a batch: then no wait at all
                    if self.is_processing and remaining > 0 and not self._batch_waiting():
                        if self._wake.wait(timeout=remaining) and self.is_processing:
                            self.early_wakeups += 1
                    elif remaining > 0:
                        self.early_wakeups += 1
                    continue
                while self.is_processing and remaining > 0:
                    time.sleep(min(0.05, remaining))
                    remaining = self.processing_interval - (time.time() - t0)
            except ShardedStepError as e:        # this rank is out of step with its peers for good
                logger.error("Stream processing stopped: %s", e)
                self.is_processing = False
                return
            except Exception as e:
                logger.error("Error in stream processing: %s", e)
                time.sleep(1.0)

    def _process_samples(self, samples: List[DatasetSample]) -> int:
        done = 0
        batches = [samples[i:i + self.batch_size] for i in range(0, len(samples), self.batch_size)]
        stages = getattr(self.online_learner, "pinned_staging", False)   # a learner with the reference's signature: no hint
        for k, batch in enumerate(batches):
            try:
                if stages and k + 1 < len(batches):
                    # the following batch is staged on the device while this update runs
                    self.online_learner.process_batch(batch, prefetch=batches[k + 1])
                else:
                    self.online_learner.process_batch(batch)
                done += len(batch)
            except Exception as e:          # the reference logs and moves on (streaming_handler.py:355-356)
                logger.error("Error processing batch: %s", e)
        return done

    def _process_samples_sharded(self, samples: List[DatasetSample]) -> int:
        """One cycle of a rank among `world`: agree on the number of update steps (the largest chunk count of any rank)
        and on every rank's chunk sizes, then run that many steps - real chunks while this rank has some, weightless
        stand-in batches after that.  Every rank calls this once per cycle, with or without samples.

        What can fail on one rank alone - collating a chunk of samples (ragged shapes, a sample without a tensor) -
        happens BEFORE the agreement: such a chunk is logged and dropped as the reference drops a failing batch
        (streaming_handler.py:355-356) and never enters the step count.  An update step itself is collective (BatchNorm
        statistics, gathered rows, the gradient all-reduce): a rank that fails inside one cannot be skipped over without
        leaving its peers waiting in a collective it never issues, so that error is not swallowed - the rank aborts its
        process group (the peers' pending collectives then raise instead of blocking), stops processing and raises
        ShardedStepError here and on every later cycle."""
        import torch
        dp = self.data_parallel
        chunks = [samples[i:i + self.batch_size] for i in range(0, len(samples), self.batch_size)]
        batches = []
        for chunk in chunks:
            try:
                batches.append((len(chunk), chunk[0].graph, torch.stack([s.metrics for s in chunk]),
                                torch.stack([s.logs for s in chunk]), torch.stack([s.label for s in chunk])))
            except Exception as e:
                logger.error("Error processing batch: %s", e)
        if batches:
            self._seq_len = int(batches[0][2].shape[1])
            if self.service_graph is None:
                self.service_graph = batches[0][1]
        # message: [chunk count, sequence length, chunk sizes ...]; one all-gather over the ranks
        msg = torch.zeros(2 + self.MAX_CHUNKS, dtype=torch.int64)
        msg[0], msg[1] = len(batches), self._seq_len
        for k, b in enumerate(batches):
            msg[2 + k] = b[0]
        dev = self.online_learner.model.device
        table = torch.empty(self.world, msg.numel(), dtype=torch.int64, device=dev)
        dp.transport.all_gather(table, msg.to(dev))
        table = table.cpu()
        n_steps = int(table[:, 0].max())
        if n_steps == 0:
            return 0
        seq_len = int(table[:, 1].max())
        done = 0
        try:
            for k in range(n_steps):
                rows = [int(table[r, 2 + k]) for r in range(self.world)]
                dp.set_step_rows(rows)
                try:
                    if k < len(batches):
                        n, graph, metrics, logs, labels = batches[k]
                        self.online_learner.process_tensors(metrics, graph, logs, labels)
                        done += n
                    else:
                        if self.service_graph is None:
                            raise RuntimeError("a rank without samples needs the service graph for its stand-in batch: "
                                               "pass service_graph= to StreamingHandler / StreamProcessor")
                        self.online_learner.process_standin(seq_len, self.service_graph)
                        self.standin_steps += 1
                except Exception as e:
                    self._leave_group(e)
                    raise ShardedStepError(f"update step {k} of a sharded cycle failed on rank {self.rank}: {e}") from e
        finally:
            dp.set_step_rows(None)
        return done

    def _leave_group(self, exc: BaseException) -> None:
        """A collective update step failed on this rank: stop processing and abort the process group so that the other
        ranks' collectives end with an error instead of waiting for messages this rank will never send."""
        self.failed = exc
        self.is_processing = False
        logger.error("Rank %d: update step failed inside a sharded cycle (%s); aborting the process group", self.rank, exc)
        abort = getattr(self.data_parallel.transport, "abort", None)
        if abort is not None:
            try:
                abort()
            except Exception as e:          # nothing more to do than say so: the peers then end on their own timeout
                logger.error("Rank %d: aborting the process group failed: %s", self.rank, e)

    def get_processing_stats(self) -> Dict[str, Any]:
        stats: Dict[str, Any] = {"total_processed_samples": self.processed_samples,
                                 "active_streams": sum(1 for s in self.streams.values() if s.is_active),
                                 "total_streams": len(self.streams), "is_processing": self.is_processing,
                                 "memory_stats": self.memory_manager.get_memory_stats()}
        if self.event_driven:
            stats.update({"event_driven": True, "wake_samples": self.wake_samples, "early_wakeups": self.early_wakeups})
        if self.world > 1:
            stats.update({"rank": self.rank, "world": self.world, "standin_steps": self.standin_steps,
                          "owned_streams": sum(1 for sid in self.streams if self.owns(sid))})
        if self.processing_times:
            t = list(self.processing_times)
            stats.update({"avg_processing_time": np.mean(t), "max_processing_time": np.max(t), "processing_time_std": np.std(t)})
        if self.throughput_history:
            t = list(self.throughput_history)
            stats.update({"avg_throughput": np.mean(t), "max_throughput": np.max(t), "current_throughput": t[-1]})
        stats["streams"] = {sid: s.get_stats() for sid, s in self.streams.items()}
        return stats

    def cleanup(self):
        while not self.processing_queue.empty():
            try:
                self.processing_queue.get_nowait()
            except queue.Empty:
                break
        for s in self.streams.values():
            s.buffer.clear()
        self.processing_times.clear()
        self.throughput_history.clear()


class StreamingHandler:
    def __init__(self, online_learner: OnlineLearner, max_concurrent_streams: int = 10, processing_interval: float = 1.0,
                 enable_memory_management: bool = True, event_driven: bool = False, wake_samples: Optional[int] = None,
                 data_parallel=None, service_graph=None):
        self.online_learner = online_learner
        self.max_concurrent_streams, self.processing_interval = max_concurrent_streams, processing_interval
        self.enable_memory_management = enable_memory_management
        self.stream_processor = StreamProcessor(online_learner=online_learner, max_streams=max_concurrent_streams,
                                                processing_interval=processing_interval, event_driven=event_driven,
                                                wake_samples=wake_samples, data_parallel=data_parallel,
                                                service_graph=service_graph)
        self.stream_callbacks: Dict[str, Callable] = {}
        self.is_active = False
        self.start_time: Optional[datetime] = None
        self.total_processed = 0

    def start(self):
        if self.is_active:
            logger.warning("Streaming handler already active")
            return
        self.is_active = True
        self.start_time = datetime.now()
        self.stream_processor.start_processing()

    def stop(self):
        self.is_active = False
        self.stream_processor.stop_processing()

    def create_stream(self, stream_id: str, buffer_size: int = 1000, callback: Optional[Callable] = None) -> bool:
        if not self.stream_processor.add_stream(DataStream(stream_id, buffer_size)):
            return False
        if callback:
            self.stream_callbacks[stream_id] = callback
        return True

    def remove_stream(self, stream_id: str) -> bool:
        ok = self.stream_processor.remove_stream(stream_id)
        if ok:
            self.stream_callbacks.pop(stream_id, None)
        return ok

    def add_sample_to_stream(self, stream_id: str, sample: DatasetSample) -> bool:
        stream = self.stream_processor.streams.get(stream_id)
        if stream is None:
            logger.warning("Stream %s not found", stream_id)
            return False
        if not self.stream_processor.owns(stream_id):     # sharded: another rank buffers and processes this stream
            return False
        ok = stream.add_sample(sample)
        cb = self.stream_callbacks.get(stream_id)
        if cb is not None:
            try:
                cb(stream_id, sample, ok)
            except Exception as e:
                logger.error("Error in stream callback: %s", e)
        if ok:
            self.total_processed += 1
        return ok

    def get_comprehensive_stats(self) -> Dict[str, Any]:
        uptime = (datetime.now() - self.start_time).total_seconds() if self.start_time else 0
        handler = {"is_active": self.is_active, "start_time": self.start_time.isoformat() if self.start_time else None,
                   "total_processed": self.total_processed, "uptime_seconds": uptime}
        if self.start_time and uptime > 0:
            handler["avg_throughput"] = self.total_processed / uptime
        return {"handler_stats": handler, "processor_stats": self.stream_processor.get_processing_stats(),
                "learner_stats": self.online_learner.get_performance_summary()}

    def health_check(self) -> Dict[str, Any]:
        health: Dict[str, Any] = {"status": "healthy", "issues": [], "warnings": []}
        if not self.is_active:
            health["issues"].append("Streaming handler not active")
            health["status"] = "unhealthy"
        mem = self.stream_processor.memory_manager.get_memory_stats()
        if mem["current_usage"] > mem["cleanup_threshold"]:
            health["warnings"].append(f"High memory usage: {mem['current_usage']:.2%}")
        stats = self.stream_processor.get_processing_stats()
        dropped = sum(s.get("dropped_samples", 0) for s in stats.get("streams", {}).values())
        if dropped > 0:
            health["warnings"].append(f"Total dropped samples: {dropped}")
        if "avg_throughput" in stats and stats["avg_throughput"] < 1.0:
            health["warnings"].append("Low processing throughput")
        return health

    def save_state(self, filepath: str):
        state = {"total_processed": self.total_processed,
                 "start_time": self.start_time.isoformat() if self.start_time else None,
                 "stream_configs": {sid: {"buffer_size": s.buffer_size, "total_samples": s.total_samples,
                                          "dropped_samples": s.dropped_samples}
                                    for sid, s in self.stream_processor.streams.items()},
                 "processing_stats": self.stream_processor.get_processing_stats()}
        self.online_learner.save_checkpoint(filepath.replace(".json", "_learner.pt"))
        with open(filepath, "w") as f:
            json.dump(state, f, indent=2, default=float)

    def load_state(self, filepath: str):
        with open(filepath, "r") as f:
            state = json.load(f)
        self.total_processed = state.get("total_processed", 0)
        if state.get("start_time"):
            self.start_time = datetime.fromisoformat(state["start_time"])
        try:
            self.online_learner.load_checkpoint(filepath.replace(".json", "_learner.pt"))
        except FileNotFoundError:
            logger.warning("Learner checkpoint not found for %s", filepath)
