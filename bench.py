"""Headline benchmark: samples/sec of the OCEAN fwd+bwd online step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one online update of OCEANModel on one batch resident in HBM:
forward, compute_loss (BCE + 0.1 * 3 in-batch InfoNCE), backward,
clip_grad_norm_(1.0), Adam - what OnlineLearner.process_batch runs, including
its one loss read-back per step.  Workload (BASELINE.json configs[2]): 256
services (directed Erdos-Renyi graph, mean in-degree 8), seq_len 128, 12
metric features, 768-d log embeddings, GLOBAL batch 1024 split evenly over
the ranks (strong scaling; --weak keeps 1024 per GPU instead), reference
default hyper-parameters (dropout 0.1 everywhere, so every graph replica is
computed), fp32.

N > 1: one process per GPU, launched by torch.distributed.run (RANK /
LOCAL_RANK / WORLD_SIZE in the environment) or, when those are absent, by
this script itself, which then starts N rank processes of itself before it
touches any GPU.  Ranks are data-parallel over RCCL: gradient arena
all-reduce, all-reduced BatchNorm statistics, all-gathered InfoNCE positives;
the whole step, collectives included, is replayed as one HIP graph.  Rank 0
prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X dense bf16 matrix peak (--split-bf16 only)
PEAK_HBM_GBS = 8000.0


def synth_inputs(B, T, N, S, rank, device):
    g = lambda s: torch.Generator().manual_seed(s)
    metrics = torch.randn(B, T, 12, generator=g(1000 + rank))
    logs = torch.randn(B, 768, generator=g(2000 + rank))
    nf = torch.randn(N, 12, generator=g(7))
    adj = (torch.rand(N, N, generator=g(11)) < 8.0 / N).float()
    adj.fill_diagonal_(0.0)
    labels = torch.zeros(B, S)
    labels[torch.arange(B), torch.randint(0, S, (B,), generator=g(3000 + rank))] = 1.0
    return metrics.to(device), logs.to(device), nf, adj, labels.to(device)


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
LIB_FILE = os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd", "ocean", "libocean_hip.so")


def file_sha16(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def recorded_traffic(kernel):
    """HBM-side bytes per launch of `kernel` (fetch + write) from rocprofv3 --pmc passes over this same command
    (tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as gfx950
    requires).  The counters cannot be read from inside this process, so the figure comes from a recording - and only
    from one made with the very kernel library and bench script that are running now (their hashes are stored with the
    recording); otherwise None: a stale figure would describe other kernels."""
    if not os.path.exists(TRAFFIC_FILE):
        return None
    rec = json.load(open(TRAFFIC_FILE))
    if not isinstance(rec, dict) or rec.get("lib_sha16") != file_sha16(LIB_FILE) or rec.get("bench_sha16") != file_sha16(__file__):
        return None
    key = kernel.replace(" ", "")
    n = tot = 0.0
    for r in rec.get("kernels", []):
        if r["kernel"].replace(" ", "") == key:
            n += r["launches"]
            tot += r["launches"] * (r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"])
    return tot / n if n else None


def cpu_baseline(model, nf, adj, T, S, budget_s=25.0):
    """The CPU oracle's update step (PyTorch CPU eager, fp32) on a bounded batch, timed with the recipe of
    the reference's profiler (3 warm-up + timed iterations, perf_counter)."""
    from oracle import ocean_ref as R
    Bc = 32
    sd = R.require_grad({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    g = lambda s: torch.Generator().manual_seed(s)
    metrics, logs = torch.randn(Bc, T, 12, generator=g(1)), torch.randn(Bc, 768, generator=g(2))
    labels = torch.zeros(Bc, S)
    labels[torch.arange(Bc), torch.randint(0, S, (Bc,), generator=g(3))] = 1.0
    tk = [k for k in R.trainable_keys(sd) if sd[k].is_floating_point()]
    state = {k: {} for k in tk}

    def step():
        for k in tk:
            sd[k].grad = None
        out = R.model_forward(sd, metrics, adj, nf, logs, training=True, negatives="in_batch", dedup_gnn=False)
        R.compute_loss(out, labels)["total_loss"].backward()
        wg = [k for k in tk if sd[k].grad is not None]
        R.clip_grad_norm([sd[k].grad for k in wg], 1.0)
        with torch.no_grad():
            R.adam_step([sd[k] for k in wg], [sd[k].grad for k in wg], [state[k] for k in wg])

    def timed(fn, budget):
        for _ in range(3):                               # 3 warm-up + up to 10 timed iterations (evaluation/profiler.py:62-111)
            fn()
        t0 = time.perf_counter()
        n = 0
        while n < 10 and time.perf_counter() - t0 < budget:
            fn()
            n += 1
        return (time.perf_counter() - t0) / n, n

    dt, n = timed(step, budget_s * 0.7)

    def eval_forward():
        with torch.no_grad():
            R.model_forward(sd, metrics, adj, nf, logs, training=False)

    dte, ne = timed(eval_forward, budget_s * 0.3)
    return {"value": Bc / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} update steps of batch {Bc} (same graph, T={T}, S={S}, in-batch InfoNCE, every graph replica computed as "
                      f"the reference does; dropout off, i.e. slightly less work than the GPU step), {dt * 1e3:.0f} ms/step",
            "eval_forward": {"value": Bc / dte, "unit": "samples/s",
                             "sample": f"{ne} eval-mode forward passes of batch {Bc}, {dte * 1e3:.0f} ms each"}}


def launch_ranks(n, argv):
    """Start n rank processes of this script (this process has not touched a GPU and does not), wait for them and
    return the worst exit code.  Rank 0's stdout (the JSON line) passes through; every rank's stderr goes to its own
    file (OCEAN_BENCH_LOG_DIR, default a fresh temporary directory), and the tail of a failing rank's file is repeated
    on this process's stderr with the rank named.  A run that exceeds OCEAN_BENCH_TIMEOUT seconds (default 900) of wall
    clock - a collective some rank never entered - is ended here with exit code 124 and the list of ranks that were
    still running, instead of waiting for the collective's own timeout."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    log_dir = os.environ.get("OCEAN_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="ocean_bench_")
    os.makedirs(log_dir, exist_ok=True)
    limit = float(os.environ.get("OCEAN_BENCH_TIMEOUT", "900"))
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        logs.append(os.path.join(log_dir, f"rank{r}.err"))
        with open(logs[-1], "wb") as err:
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stderr=err))

    def tail(r, lines=25):
        try:
            with open(logs[r], "rb") as f:
                text = f.read().decode(errors="replace").splitlines()[-lines:]
        except OSError:
            text = []
        print(f"[bench.py] rank {r} stderr ({logs[r]}), last lines:", file=sys.stderr)
        for ln in text:
            print(f"[rank {r}] {ln}", file=sys.stderr)

    # wait for all of them; when one fails, the others (stuck in a collective that will never complete) are stopped
    # instead of being waited for until the collective's own timeout
    worst, live, t0 = 0, list(range(n)), time.time()
    while live:
        time.sleep(0.2)
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.remove(r)
            if rc != 0 and worst == 0:
                worst = rc
                print(f"[bench.py] rank {r} exited with code {rc}; stopping ranks {live}", file=sys.stderr)
                tail(r)
                for other in live:
                    procs[other].terminate()
        if live and worst == 0 and time.time() - t0 > limit:
            worst = 124
            print(f"[bench.py] no result after {limit:.0f} s; ranks still running: {live} (stopping them)", file=sys.stderr)
            for r in live:
                tail(r, 12)
                procs[r].terminate()
    sys.stderr.flush()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="global batch, split evenly over the ranks (per GPU with --weak)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --batch samples on every GPU")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--services", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (by default the step is replayed as a HIP graph)")
    ap.add_argument("--split-bf16", nargs="?", const="all", default=None, choices=["all", "backward"],
                    help="secondary, labelled line (never the headline): the large convolution products on fp32 operands split into "
                         "two bf16 values, three bf16 matrix instructions per product, fp32 accumulation; 'backward' keeps the "
                         "forward pass fp32")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # stdout carries the ONE JSON line and nothing else: libraries that print there (the RCCL and gloo banners) are sent
    # to stderr by pointing descriptor 1 at it; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    # OCEAN_BENCH_REHEARSE=1: every rank uses GPU 0 and the collectives are staged through the host over gloo, so that
    # the N > 1 control flow (phased BatchNorm statistics, gathers, gradient arena, barriers, max over ranks) can be
    # run on a one-GPU box.  Its numbers mean nothing: the ranks share one GPU.
    rehearse = os.environ.get("OCEAN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1 or rehearse:
        # a rank stuck in a collective dumps every thread's stack into its stderr and exits non-zero, under whichever
        # launcher it runs (this script's own or torch.distributed.run)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("OCEAN_BENCH_RANK_TIMEOUT", os.environ.get("OCEAN_BENCH_REHEARSE_TIMEOUT", "600"))), exit=True)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # OCEAN_BENCH_DIST=1 takes the data-parallel code path (RCCL process group, collectives) with a single rank as
    # well: a smoke test of that path on a one-GPU box
    use_dist = world > 1 or os.environ.get("OCEAN_BENCH_DIST") == "1"
    transport = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:        # (single rank without a launcher) a free port, not a fixed one a run that just ended may still hold
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    import ocean
    from ocean import _hip
    from ocean.data import ServiceGraph
    from ocean.models.training import OnlineLearner

    if args.batch % world and not args.weak:
        raise SystemExit(f"bench.py: global batch {args.batch} is not a multiple of {world} ranks")
    B = args.batch if args.weak else args.batch // world
    T, N = args.seq_len, args.services
    cfg = ocean.default_config()
    cfg.model.num_services = N
    torch.manual_seed(42)
    model = ocean.OCEANModel(cfg).to(device)
    model.contrastive_negatives = "in_batch"
    learner = OnlineLearner(model, cfg)
    if args.split_bf16:
        ocean.set_split_bf16(args.split_bf16)
    if use_dist:
        from ocean.parallel import DataParallel, HostStagedTransport
        transport = HostStagedTransport() if rehearse else None
        dp = DataParallel(model, learner, transport=transport)
        transport = dp.transport
    learner.use_hip_graph = not args.no_graph         # with ranks on RCCL the recorded step contains the collectives
    metrics, logs, nf, adj, labels = synth_inputs(B, T, N, N, rank, device)
    graph = ServiceGraph(adjacency_matrix=adj, node_features=nf)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    recorded = True
    for _ in range(max(args.warmup, 3 if learner.use_hip_graph else 0)):   # the graph is recorded on the third step
        try:
            learner.process_tensors(metrics, graph, logs, labels)
        except RuntimeError as e:
            if not (use_dist and "recording the update step" in str(e)):
                raise
            # every rank runs the same code on the same shapes, so a recording that fails fails on all of them: they
            # all go on with separate launches (the learner has switched recording off)
            recorded = False
            print(f"[rank {rank}] update step not recordable with its collectives, launching eagerly: {e.__cause__}", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        learner.process_tensors(metrics, graph, logs, labels)
    barrier()
    dt = time.perf_counter() - t0
    graphed = bool(learner._graphed)
    # Roofline kernel timing: the library brackets each GEMM launch with HIP events on the launch stream.  That costs
    # host time per launch and sees nothing under HIP-graph replay, so it runs over a few eagerly launched steps after
    # the timed region (on every rank: the steps contain the collectives).
    prof_steps = max(3, args.steps // 4)
    learner.use_hip_graph = False
    _hip.profile_enable(True)
    for _ in range(prof_steps):
        learner.process_tensors(metrics, graph, logs, labels)
    torch.cuda.synchronize()
    prof = _hip.profile_summary()
    _hip.profile_enable(False)
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        transport.all_reduce(t, "max")
        dt = float(t.item())

    if rank == 0:
        dom = max(prof, key=lambda r: r["total_ms"])
        achieved = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS
        if args.split_bf16:          # three bf16 instructions are issued per product counted in dom["flops"]
            achieved, peak = 3.0 * achieved, PEAK_BF16_MFMA_TFLOPS
        gemm_ms = sum(r["total_ms"] for r in prof) / prof_steps
        out = {
            "metric": "samples/sec fwd+bwd online step (256 svc, seq 128)",
            "value": world * B * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": ("bf16x3 (fp32 operands split hi + lo, fp32 accumulate)" + (" in the backward pass, f32 forward" if args.split_bf16 == "backward" else "")) if args.split_bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2] (C3): {N} services (ER graph, in-degree 8), global batch {world * B} "
                                   f"({B} per GPU), seq_len {T}, "
                                   "online update step = fwd + BCE/InfoNCE(in-batch) loss + bwd + clip(1.0) + Adam, "
                                   "reference default hyper-parameters (dropout 0.1: every graph replica computed)",
                       "global_batch": world * B, "batch_per_gpu": B, "seq_len": T, "services": N,
                       "parallelism": f"dp{world}" + (" (rehearsal: all ranks on one GPU, gloo)" if rehearse else ""),
                       "launch": "hip-graph replay" + (" (collectives inside the graph)" if use_dist else "") if graphed else "eager",
                       "arithmetic": ("SECONDARY LINE, not the headline: large convolution products"
                                      + (" of the backward pass" if args.split_bf16 == "backward" else "")
                                      + " as hi.hi + hi.lo + lo.hi of bf16 halves (about 16 bits per product), everything else fp32; "
                                      "roofline counts the bf16 instructions issued"
                                      if args.split_bf16 else "fp32 throughout")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": recorded_traffic(dom["kernel"]) if (B, T, N, world, use_dist, bool(args.split_bf16)) == (1024, 128, 256, 1, False, False) else None,
                         "traffic_unit": "bytes/launch (HBM-side fetch+write, separate rocprofv3 --pmc passes of the default command; "
                                         "null unless recorded with this very library and script, and for any other shape)",
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"], "kernel": dom["kernel"],
                         "launches_per_step": dom["launches"] / prof_steps,
                         "avg_launch_ms": dom["total_ms"] / dom["launches"],
                         "all_gemm_ms_per_step": gemm_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, nf, adj, T, N)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
