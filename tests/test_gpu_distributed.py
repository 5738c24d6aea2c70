"""RCCL on the MI355X box (the path's ONE collective is an all-gather of generated ids).  The driver's 1-GPU lease cannot
measure scaling, but it can prove what an 8-GPU lease depends on: `backend="nccl"` (= RCCL on ROCm) initialises here with
HSA_ENABLE_IPC_MODE_LEGACY=0, and `gather_tokens` moves int64 DEVICE tensors through `all_gather_into_tensor` -- a world of one
short-circuits unless force_collective is set, so the test sets it.  Runs in a child process (a process group is process-global
state; the rest of the suite stays without one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "visual-chinese-llama-alpaca_amd"))
from visualcla.distributed import gather_tokens, shard_range
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
assert dist.get_backend() == "nccl"
g = torch.Generator().manual_seed(3)
toks = torch.randint(0, 49958, (64, 128), generator=g).cuda()           # one rank's share of configs[3]: 64 requests x 128 new ids
assert toks.dtype == torch.int64 and toks.is_cuda
same = gather_tokens(toks)                                                # world 1: no collective
assert same is toks
out = gather_tokens(toks, n_total=64, n_cols=128, pad_id=0, force_collective=True)    # ONE all_gather_into_tensor through RCCL
torch.cuda.synchronize()
assert out.shape == (64, 128) and out.dtype == torch.int64 and out.is_cuda and torch.equal(out, toks)
ragged = gather_tokens(toks[:37, :100], pad_id=-1, force_collective=True)             # shape exchange + gather (two collectives)
assert torch.equal(ragged, toks[:37, :100])
padded = gather_tokens(toks[:5, :9], n_total=5, n_cols=12, pad_id=-7, force_collective=True)   # early-EOS shard: columns padded
assert padded.shape == (5, 12) and torch.equal(padded[:, :9], toks[:5, :9]) and bool((padded[:, 9:] == -7).all())
# the bench's timing collectives on the same backend
t = torch.tensor([1.25], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
assert float(t) == 1.25
import ctypes
loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
print("RCCL_OK", sorted(set(os.path.basename(p) for p in loaded)))
dist.destroy_process_group()
"""


def test_rccl_world1_gathers_device_int64_tokens(tmp_path):
    import socket
    with socket.socket() as sk:                       # a free rendezvous port (a fixed one may be taken on a shared box)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "librccl" in r.stdout, r.stdout          # the collective really went through RCCL's shared object
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "a") as f:
        f.write("rccl world-1: nccl backend initialised, gather_tokens(force_collective) int64 device tensors OK; " + r.stdout.strip().splitlines()[-1] + "\n")
