"""The batch stager must hand the model the same values as the reference's three torch.stack calls
(ocean/models/training/online_learner.py:378-393), reuse its buffers, and never reuse one too early."""
from datetime import datetime

import pytest
import torch


def _samples(n, seed, T=7, F=5, D=24, S=6, graph=None):
    from ocean.data import DatasetSample
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        lab = torch.zeros(S)
        lab[int(torch.randint(0, S, (1,), generator=g))] = 1.0
        out.append(DatasetSample(metrics=torch.randn(T, F, generator=g), logs=torch.randn(D, generator=g), graph=graph,
                                 label=lab, timestamp=datetime.now()))
    return out


def test_collate_equals_stack_and_ring_reuses_buffers():
    from ocean.staging import BatchStager
    st = BatchStager("cpu", depth=2)
    a, b, c = _samples(3, 1), _samples(3, 2), _samples(3, 3)
    ta = st.collate(a)
    for got, name in zip(ta, ("metrics", "logs", "label")):
        assert torch.equal(got, torch.stack([getattr(s, name) for s in a]))
    tb = st.collate(b)
    assert ta[0].data_ptr() != tb[0].data_ptr()                 # second slot of the ring
    assert torch.equal(ta[0], torch.stack([s.metrics for s in a]))  # first slot untouched so far
    tc = st.collate(c)
    assert tc[0].data_ptr() == ta[0].data_ptr()                 # depth 2: the third batch reuses the first slot
    assert torch.equal(tc[2], torch.stack([s.label for s in c]))
    # a different batch size gets its own ring
    td = st.collate(_samples(5, 4))
    assert td[0].shape[0] == 5 and td[0].data_ptr() not in (ta[0].data_ptr(), tb[0].data_ptr())


def test_shape_ring_count_is_bounded():
    from ocean.staging import BatchStager
    st = BatchStager("cpu", depth=1, max_shapes=2)
    for n in (1, 2, 3, 4):
        st.collate(_samples(n, n))
    assert len(st._rings) == 2


def test_ragged_and_non_fp32_batches_are_left_to_the_plain_collate():
    from ocean.staging import BatchStager
    s = _samples(2, 5)
    assert BatchStager.stageable(s)
    s[1].logs = torch.randn(3, 24)
    assert not BatchStager.stageable(s)
    s = _samples(2, 6)
    s[0].metrics = s[0].metrics.double()
    assert not BatchStager.stageable(s)
    # a mismatch in the middle of the batch passes the screen and is caught while stacking
    s = _samples(5, 7)
    s[2].logs = torch.randn(3, 24)
    assert BatchStager.stageable(s) and BatchStager("cpu").collate(s) is None
    s = _samples(5, 8)
    s[2].metrics = s[2].metrics.double()
    assert BatchStager("cpu").collate(s) is None


@pytest.mark.gpu
def test_process_batch_staged_equals_stacked():
    """Six consecutive updates from host samples, staged (pinned ring, depth 2) vs stacked: the same updates."""
    import ocean
    from ocean.data import ServiceGraph
    from ocean.models.training import OnlineLearner
    from oracle import synth

    def learner(staged):
        cfg = ocean.OCEANConfig()
        cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates = [8, 16], [1, 2]
        cfg.model.hidden_dim = cfg.model.attention_dim = 16
        cfg.model.attention_num_heads, cfg.model.gnn_num_heads = 4, 2
        cfg.model.fusion_projection_dim, cfg.model.num_services = 32, 6
        cfg.data.metrics_features, cfg.data.log_embedding_dim, cfg.data.graph_features = 5, 24, 7
        m = ocean.OCEANModel(cfg)
        m.load_state_dict(synth.fill_state_dict(m.state_dict(), 3))
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        m = m.cuda()
        m.contrastive_negatives = "in_batch"
        lrn = OnlineLearner(m, cfg)
        lrn.pinned_staging = staged
        return m, lrn

    graph = ServiceGraph(adjacency_matrix=synth.random_digraph(6, 2.0, 4),
                         node_features=torch.randn(6, 7, generator=torch.Generator().manual_seed(1)))
    (ma, la), (mb, lb) = learner(True), learner(False)
    batches = [_samples(4 if i != 3 else 3, 100 + i, graph=graph) for i in range(6)]   # one odd-sized batch in the middle
    for i, batch in enumerate(batches):
        # from the third update on, the staged learner is also told its next batch (collated under the running step)
        ra = la.process_batch(batch, prefetch=batches[i + 1] if 2 <= i < 5 else None)
        assert (la._prefetched is not None) == (2 <= i < 5)
        rb = lb.process_batch(batch)
        assert ra["update_count"] == rb["update_count"] and ra["batch_size"] == rb["batch_size"]
        assert abs(ra["batch_loss"] - rb["batch_loss"]) <= 1e-5 * max(1.0, abs(ra["batch_loss"]))
    assert la._stager is not None and lb._stager is None and la._prefetched is None
    # a prefetched batch that is not the one processed next is dropped, not used
    other = _samples(4, 999, graph=graph)
    la.process_batch(batches[0], prefetch=batches[1])
    ra, rb = la.process_batch(other), (lb.process_batch(batches[0]), lb.process_batch(other))[1]
    assert abs(ra["batch_loss"] - rb["batch_loss"]) <= 1e-5 * max(1.0, abs(ra["batch_loss"]))
    # same inputs, same kernels; the split reductions add in a run-dependent order, which Adam amplifies a little
    for (n, p), q in zip(ma.named_parameters(), mb.parameters()):
        scale = max(q.double().abs().max().item(), 1e-3)
        assert (p.double() - q.double()).abs().max().item() <= 2e-3 * scale, n


def test_replay_pool_follows_the_memory_buffer_through_evictions():
    """Rows are taken on add, handed back on eviction (all three retention strategies), and a gathered batch holds
    the samples' own bytes in the order MemoryBuffer.sample chose."""
    import numpy as np
    from ocean.models.training.online_learner import MemoryBuffer
    from ocean.staging import ReplayPool
    for strategy in ("importance", "fifo", "random"):
        buf = MemoryBuffer(max_size=5, retention_strategy=strategy)
        pool = ReplayPool("cpu", 5)
        buf.on_evict = pool.evict
        np.random.seed(3)
        samples = _samples(12, 40)
        for k, s in enumerate(samples):
            buf.add(s, importance_score=float((7 * k) % 11) + 0.5)
            assert pool.put(s)
            assert len(pool) == len(buf) == min(k + 1, 5)
            assert all(b in pool for b in buf.buffer)
        assert sorted(pool._free + list(pool._slot_of.values())) == list(range(5))
        batch = buf.sample(3)
        m, l, y = pool.gather(batch)
        assert torch.equal(m, torch.stack([s.metrics for s in batch]))
        assert torch.equal(l, torch.stack([s.logs for s in batch]))
        assert torch.equal(y, torch.stack([s.label for s in batch]))
        held = {id(b) for b in buf.buffer}
        evicted = [s for s in samples if id(s) not in held]
        assert pool.gather([batch[0], evicted[0]]) is None         # a batch with an unpooled sample is not served
    # samples that do not fit the pool's row shape, or arrive when it is full, are left to the host path
    pool = ReplayPool("cpu", 2)
    a, b, c = _samples(3, 41)
    odd = _samples(1, 42, T=9)[0]
    assert pool.put(a) and not pool.put(odd) and pool.put(b) and not pool.put(c)
    assert pool.put(a) and len(pool) == 2                          # already pooled: still one row
    pool.evict(a)
    assert pool.put(c) and pool.gather([c, b]) is not None and pool.gather([a]) is None
    pool.clear()
    assert len(pool) == 0 and len(pool._free) == 2


@pytest.mark.gpu
def test_experience_replay_from_the_device_pool_equals_replay_from_host_samples():
    """Two learners see the same stream of process_sample calls and then replay with the same random draws; one
    gathers its replay batches from HBM, the other collates them from the host samples as the reference does
    (ocean/models/training/online_learner.py:433-453).  Same batches, same updates."""
    import numpy as np
    import ocean
    from ocean.data import ServiceGraph
    from ocean.models.training import OnlineLearner
    from oracle import synth

    def learner(device_replay):
        cfg = ocean.OCEANConfig()
        cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates = [8, 16], [1, 2]
        cfg.model.hidden_dim = cfg.model.attention_dim = 16
        cfg.model.attention_num_heads, cfg.model.gnn_num_heads = 4, 2
        cfg.model.fusion_projection_dim, cfg.model.num_services = 32, 6
        cfg.data.metrics_features, cfg.data.log_embedding_dim, cfg.data.graph_features = 5, 24, 7
        cfg.training.sliding_window_size = 6                      # memory buffer capacity: forces evictions below
        m = ocean.OCEANModel(cfg)
        m.load_state_dict(synth.fill_state_dict(m.state_dict(), 3))
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        m = m.cuda()
        m.contrastive_negatives = "in_batch"
        lrn = OnlineLearner(m, cfg)
        lrn.device_replay = device_replay
        return m, lrn

    graph = ServiceGraph(adjacency_matrix=synth.random_digraph(6, 2.0, 4),
                         node_features=torch.randn(6, 7, generator=torch.Generator().manual_seed(1)))
    (ma, la), (mb, lb) = learner(True), learner(False)
    stream = _samples(9, 77, graph=graph)
    for s in stream:                                              # update_model=False: identical scores on both sides
        ra, rb = la.process_sample(s, update_model=False), lb.process_sample(s, update_model=False)
        assert abs(ra["importance_score"] - rb["importance_score"]) <= 1e-6 * max(1.0, abs(rb["importance_score"]))
    assert len(la.memory_buffer) == 6 and len(la._replay_pool) == 6 and lb._replay_pool is None
    assert [id(s) for s in la.memory_buffer.buffer] == [id(s) for s in lb.memory_buffer.buffer]
    for k in range(4):
        np.random.seed(10 + k)
        ra = la.experience_replay(4)
        np.random.seed(10 + k)
        rb = lb.experience_replay(4)
        assert ra["replay"] and rb["replay"] and ra["batch_size"] == rb["batch_size"] == 4
        assert abs(ra["batch_loss"] - rb["batch_loss"]) <= 1e-5 * max(1.0, abs(rb["batch_loss"]))
    assert la.experience_replay(7) is None                        # fewer samples than asked for: the reference's None
    for (n, p), q in zip(ma.named_parameters(), mb.parameters()):
        scale = max(q.double().abs().max().item(), 1e-3)
        assert (p.double() - q.double()).abs().max().item() <= 2e-3 * scale, n
    la.reset()
    assert la._replay_pool is None and len(la.memory_buffer) == 0
