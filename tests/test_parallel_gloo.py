"""world_size-2 gloo tests (CPU) of the data-parallel host logic in
ocean/parallel.py: the flat gradient bucket, the all-gather whose backward is a
reduce-scatter (in-batch InfoNCE across ranks), and the BatchNorm statistics
protocol.  Each check compares 2 ranks on half batches against one process on
the full batch, using the CPU oracle's InfoNCE as the reference."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_batch_reference():
    from oracle import ocean_ref as R
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 6, generator=g)
    wa = torch.randn(5, 6, generator=g).requires_grad_(True)
    wp = torch.randn(5, 6, generator=g).requires_grad_(True)
    loss = R.infonce(x @ wa.t(), y @ wp.t(), None, 0.1)
    loss.backward()
    return x, y, wa.detach(), wp.detach(), loss.detach(), wa.grad, wp.grad


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ocean.parallel import GradBucket, all_gather_rows
        x, y, wa0, wp0, loss_full, ga_full, gp_full = _full_batch_reference()
        Bl = x.shape[0] // world
        xs, ys = x[rank * Bl:(rank + 1) * Bl], y[rank * Bl:(rank + 1) * Bl]
        wa, wp = wa0.clone().requires_grad_(True), wp0.clone().requires_grad_(True)
        a = F.normalize(xs @ wa.t(), dim=1)
        p_all = F.normalize(all_gather_rows(ys @ wp.t()), dim=1)          # [B_global, P], autograd-aware
        logits = a @ p_all.t() / 0.1
        labels = torch.arange(Bl) + rank * Bl                              # this rank's row offset
        loss = F.cross_entropy(logits, labels)                             # local mean
        loss.backward()
        GradBucket().all_reduce_mean([wa.grad, wp.grad])
        lt = loss.detach().clone()
        dist.all_reduce(lt)
        ok_loss = torch.allclose(lt / world, loss_full, atol=1e-5)
        ok_grad = torch.allclose(wa.grad, ga_full, atol=1e-5) and torch.allclose(wp.grad, gp_full, atol=1e-5)
        # BatchNorm statistics protocol: fp64 [sum, sum of squares] blocks all-reduced, then finalised identically
        u = torch.randn(16, 4, generator=torch.Generator().manual_seed(3))
        us = u[rank * 8:(rank + 1) * 8].double()
        stats = torch.cat([us.sum(0), (us * us).sum(0)])
        dist.all_reduce(stats)
        n = 16.0
        mean, var = stats[:4] / n, stats[4:] / n - (stats[:4] / n) ** 2
        ok_bn = torch.allclose(mean.float(), u.mean(0), atol=1e-6) and torch.allclose(var.float(), u.var(0, unbiased=False), atol=1e-6)
        # DataParallel wiring on a stand-in model: broadcast + hooks
        out[rank] = (ok_loss, ok_grad, ok_bn)
    finally:
        dist.destroy_process_group()


def test_two_rank_equals_full_batch():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: (True, True, True), 1: (True, True, True)}


def _ragged_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ocean.parallel import GradBucket, Transport, all_gather_rows_ragged
        x, y, wa0, wp0, loss_full, ga_full, gp_full = _full_batch_reference()
        results = []
        # 5 + 3 rows; then 8 + 0, rank 1 taking part with a one-row stand-in that must count for nothing
        for counts in ([5, 3], [8, 0]):
            lo = sum(counts[:rank])
            n = counts[rank]
            xs, ys = (x[lo:lo + n], y[lo:lo + n]) if n > 0 else (x[:1] * 3.0, y[:1] * -2.0)
            wa, wp = wa0.clone().requires_grad_(True), wp0.clone().requires_grad_(True)
            a = F.normalize(xs @ wa.t(), dim=1)
            p_all = F.normalize(all_gather_rows_ragged(ys @ wp.t(), counts), dim=1)
            assert p_all.shape[0] == sum(counts)
            logits = a @ p_all.t() / 0.1
            labels = torch.arange(xs.shape[0]) + (lo if n > 0 else 0)
            weight = world * n / float(sum(counts))                       # DataParallel.loss_weight
            loss = F.cross_entropy(logits, labels) * weight
            loss.backward()
            GradBucket().all_reduce_mean([wa.grad, wp.grad])
            lt = loss.detach().clone()
            dist.all_reduce(lt)
            results.append(bool(torch.allclose(lt / world, loss_full, atol=1e-5) and torch.allclose(wa.grad, ga_full, atol=1e-5)
                                and torch.allclose(wp.grad, gp_full, atol=1e-5)))
        # a table that promises more rows than the rank holds is refused before any collective
        try:
            all_gather_rows_ragged(torch.zeros(1, 3), [2, 2] if rank == 0 else [2, 2], Transport(None))
            results.append(False)
        except ValueError:
            results.append(True)
        out[rank] = tuple(results)
    finally:
        dist.destroy_process_group()


def test_ragged_gather_of_in_batch_negatives_equals_the_union_batch():
    """Unequal row counts per rank (sharded streams): anchors are scored against the rows of ALL ranks, gathered by their
    true counts; with the row-count loss weights the rank-averaged gradient is the union batch's (graph_fusion.py:144-157
    evaluated by one process on all rows)."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ragged_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: (True, True, True), 1: (True, True, True)}


def _wire_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocean
        from ocean.parallel import DataParallel
        cfg = ocean.default_config()
        cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates = [8, 16], [1, 2]
        cfg.model.hidden_dim = cfg.model.attention_dim = 16
        cfg.model.attention_num_heads, cfg.model.gnn_num_heads, cfg.model.fusion_projection_dim = 4, 2, 32
        torch.manual_seed(100 + rank)                      # ranks start with different weights
        model = ocean.OCEANModel(cfg)

        class L:
            grad_sync = None
        learner = L()
        dp = DataParallel(model, learner)
        w = model.prediction_head[0].weight.detach().clone()
        gathered = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        bank = model.fusion_module.contrastive_module.memory_bank.clone()
        gb = [torch.zeros_like(bank) for _ in range(world)]
        dist.all_gather(gb, bank)
        enc, cm = model.temporal_encoder, model.fusion_module.contrastive_module
        hooks = (enc.stats_all_reduce is not None and enc.stats_world_size == world and cm.gather_positives is not None
                 and cm.rank_row_offset(4) == 4 * rank and learner.grad_sync is not None)
        for p in model.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        dp.sync_grads()
        avg = all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in model.parameters())
        out[rank] = (same, torch.equal(gb[0], gb[1]), hooks, avg)
    finally:
        dist.destroy_process_group()


def test_data_parallel_wiring_broadcasts_and_averages():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_wire_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: (True, True, True, True), 1: (True, True, True, True)}


class _ShardStubLearner:
    """Records what the sharded stream processor asks of it (no model arithmetic)."""

    class _M:
        device = torch.device("cpu")

    def __init__(self):
        self.model, self.calls = self._M(), []

    fail_on_rows = None                # a step with this many rows raises (a failure inside a collective step)

    def process_tensors(self, metrics, graph, logs, labels):
        if metrics.shape[0] == self.fail_on_rows:
            raise RuntimeError("kernel fault stand-in")
        self.calls.append(("batch", int(metrics.shape[0])))
        return {}

    def process_standin(self, seq_len, graph):
        self.calls.append(("standin", seq_len))
        return {}


class _ShardStubDP:
    def __init__(self):
        from ocean.parallel import Transport
        self.transport = Transport()
        self.aborted = 0
        self.transport.abort = self._abort          # the test's process group must survive to be torn down in order
        self.rank, self.world = self.transport.rank, self.transport.world
        self.rows = []

    def _abort(self):
        self.aborted += 1

    def set_step_rows(self, rows):
        self.rows.append(rows)


def _shard_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from ocean.models.training import StreamingHandler
        learner, dp = _ShardStubLearner(), _ShardStubDP()
        h = StreamingHandler(learner, max_concurrent_streams=4, processing_interval=0.05, data_parallel=dp, service_graph="G")
        for s in range(4):
            assert h.create_stream(f"s{s}", buffer_size=100)
        sp = h.stream_processor
        owned = [sid for sid in sp.streams if sp.owns(sid)]
        sample = lambda: SimpleNamespace(metrics=torch.zeros(9, 3), logs=torch.zeros(5), label=torch.zeros(4), graph="G")
        # rank 0 receives 2 x 35 samples on its streams (s0, s2), rank 1 10 on one of its (s1); samples offered to a stream
        # another rank owns are turned away
        accepted = 0
        for sid, n in (("s0", 35), ("s2", 35), ("s1", 10)):
            for _ in range(n):
                accepted += bool(h.add_sample_to_stream(sid, sample()))
        done1 = sp.process_once()          # rank 0: 32 + 32 taken -> chunks 32, 32; rank 1: 10 -> one chunk
        done2 = sp.process_once()          # rank 0: the remaining 3 + 3; rank 1: nothing -> one stand-in
        done3 = sp.process_once()          # nobody has anything: no step at all
        out[rank] = {"owned": owned, "accepted": accepted, "done": (done1, done2, done3), "calls": learner.calls,
                     "rows": dp.rows, "standins": sp.get_processing_stats()["standin_steps"]}
    finally:
        dist.destroy_process_group()


def test_sharded_streams_agree_on_step_counts():
    """Streams are dealt to ranks round-robin; per cycle the ranks run the same number of update steps (the largest
    chunk count), a rank that has run out taking part with stand-in batches, and every rank knows every rank's row
    count of every step (ocean/models/training/streaming_handler.py:303-358 is the single-process loop this extends)."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_shard_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["owned"] == ["s0", "s2"] and r1["owned"] == ["s1", "s3"]
    assert r0["accepted"] == 70 and r1["accepted"] == 10
    assert r0["done"] == (64, 6, 0) and r1["done"] == (10, 0, 0)
    assert r0["calls"] == [("batch", 32), ("batch", 32), ("batch", 6)]
    assert r1["calls"] == [("batch", 10), ("standin", 9), ("standin", 9)]
    # both ranks saw the same per-step row tables, and switched the weighting off again after each cycle
    assert r0["rows"] == r1["rows"] == [[32, 10], [32, 0], None, [6, 0], None]
    assert r0["standins"] == 0 and r1["standins"] == 2


def _shard_failure_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from ocean.models.training import StreamingHandler
        from ocean.models.training.streaming_handler import ShardedStepError, StreamProcessor
        learner, dp = _ShardStubLearner(), _ShardStubDP()
        h = StreamingHandler(learner, max_concurrent_streams=4, processing_interval=0.05, data_parallel=dp, service_graph="G")
        for s in range(4):
            assert h.create_stream(f"s{s}", buffer_size=5000)
        sp = h.stream_processor
        good = lambda T=9: SimpleNamespace(metrics=torch.zeros(T, 3), logs=torch.zeros(5), label=torch.zeros(4), graph="G")
        # cycle 1: rank 0 holds one chunk whose samples do not stack (two sequence lengths) and one good chunk of 7;
        # rank 1 holds one good chunk of 3.  The ragged chunk is dropped before the ranks agree: one step, rows [7, 3].
        if rank == 0:
            for i in range(32):
                h.add_sample_to_stream("s0", good(9 if i else 8))
            for _ in range(7):
                h.add_sample_to_stream("s2", good())
        else:
            for _ in range(3):
                h.add_sample_to_stream("s1", good())
        done1 = sp.process_once()
        rows1 = list(dp.rows)
        # cycle 2: more samples wait on rank 0 than one cycle's message has room for; the surplus stays buffered
        sp.MAX_CHUNKS, room = 2, 2 * sp.batch_size
        if rank == 0:
            for _ in range(40):
                h.add_sample_to_stream("s0", good())
                h.add_sample_to_stream("s2", good())
        done2 = sp.process_once()
        left = sum(len(s.buffer) for s in sp.streams.values())
        done3 = sp.process_once()
        sp.MAX_CHUNKS = StreamProcessor.MAX_CHUNKS
        # cycle 4: rank 1's step fails inside the (collective) update: not swallowed
        learner.fail_on_rows = 5 if rank == 1 else None
        for _ in range(5):
            h.add_sample_to_stream("s0" if rank == 0 else "s1", good())
        err = again = None
        try:
            sp.process_once()
        except ShardedStepError as e:
            err = str(e)
        if rank == 1:
            try:
                sp.process_once()
            except ShardedStepError as e:
                again = str(e)
        out[rank] = {"done": (done1, done2, done3), "rows1": rows1, "left": left, "room": room, "err": err, "again": again,
                     "aborted": dp.aborted, "failed": sp.failed is not None, "calls": learner.calls}
    finally:
        dist.destroy_process_group()


def test_sharded_streams_failures_and_surplus():
    """What one rank can get wrong alone (a chunk that does not collate) is dropped before the ranks agree on the cycle's
    steps; samples beyond the agreement message's room stay in their buffers; an error inside a collective update step
    is raised, marks the processor failed and aborts the group instead of leaving the peers in unmatched collectives."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_shard_failure_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["rows1"] == r1["rows1"] == [[7, 3], None]
    assert r0["done"][0] == 7 and r1["done"][0] == 3
    assert r0["done"][1] == r0["room"] and r0["left"] == 80 - r0["room"] and r0["done"][2] == 80 - r0["room"]
    assert r0["err"] is None and not r0["failed"] and r0["aborted"] == 0
    assert r1["err"] is not None and "rank 1" in r1["err"] and r1["failed"] and r1["aborted"] == 1
    assert r1["again"] is not None
