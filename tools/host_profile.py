"""Host-side cost of one replayed update step: cProfile over N calls of OnlineLearner.process_tensors with use_hip_graph
(the GPU is idle between the loss read-back of one step and the graph launch of the next for as long as the host takes).
usage: python tools/host_profile.py [BATCH] [SEQ] [SERVICES] [STEPS]    (OCEAN_HOST_PROFILE_EAGER=1: separate launches instead)"""
import cProfile
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd")); sys.path.insert(0, ROOT)
import torch
import bench
import ocean
from ocean.data import ServiceGraph
from ocean.models.training import OnlineLearner

B, T, N, K = (int(a) for a in (sys.argv[1:5] + ["8", "64", "64", "300"][len(sys.argv) - 1:]))
cfg = ocean.default_config(); cfg.model.num_services = N
torch.manual_seed(42)
model = ocean.OCEANModel(cfg).cuda(); model.contrastive_negatives = "in_batch"
learner = OnlineLearner(model, cfg); learner.use_hip_graph = os.environ.get('OCEAN_HOST_PROFILE_EAGER') != '1'
metrics, logs, nf, adj, labels = bench.synth_inputs(B, T, N, N, 0, torch.device("cuda"))
graph = ServiceGraph(adjacency_matrix=adj, node_features=nf)
for _ in range(10):
    learner.process_tensors(metrics, graph, logs, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    learner.process_tensors(metrics, graph, logs, labels)
torch.cuda.synchronize()
print(f"B={B} T={T} N={N}: {(time.perf_counter() - t0) / K * 1e6:.1f} us per step ({'replayed' if learner.use_hip_graph else 'eager'})")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    learner.process_tensors(metrics, graph, logs, labels)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
