"""Can two ranks share one GPU over RCCL on this box?  (They cannot with stock settings: the library refuses duplicate
devices.)  usage: python tools/try_rccl_one_gpu.py  - prints what happened; exits 0 either way."""
import os
import subprocess
import sys

if "RANK" not in os.environ:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=50)
        except subprocess.TimeoutExpired:
            p.kill()
            out = "(timed out)"
        print(f"--- rank {r} exit {p.returncode}\n" + "\n".join(out.strip().splitlines()[-6:]))
    sys.exit(0)

import torch
import torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=int(os.environ["RANK"]), world_size=2)
    t = torch.ones(4, device="cuda") * (1 + int(os.environ["RANK"]))
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("all_reduce over two ranks on one GPU:", t.tolist())
    dist.destroy_process_group()
except Exception as e:      # noqa: BLE001 - the point is to report whatever the library says
    print("refused:", type(e).__name__, str(e).splitlines()[0][:300])
