"""The path's collective on RCCL itself (backend "nccl" on ROCm): a process group of ONE rank on the GPU box -- the largest group a 1-GPU box
can form -- through the same helpers the N-GPU bench / evaluate.py use (siu3r_amd/distributed.py).  The sharding logic and the world-size-2
arithmetic are covered on CPU with gloo (tests/test_distributed_cpu.py); what this adds is that RCCL initialises, that the statistics
vector travels as a DEVICE tensor (_collective_device) and that barrier / max-over-ranks run on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
from siu3r_amd import distributed as D
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
assert dist.get_backend() == "nccl" and D._collective_device(dev) == dev
v = D.pack_stats(dict(n_pairs=3, n_images=6, sum_psnr=150.0, n_segments=7, label_checksum=12345.0))
g = D.all_gather_stats(v, device=dev)
assert g.shape == (1, len(D.STAT_KEYS)) and torch.equal(g[0], v), g
r = D.reduce_stats(g)
assert r["n_pairs"] == 3 and abs(r["psnr"] - 25.0) < 1e-12
assert D.max_over_ranks(1.25, device=dev) == 1.25
D.barrier()
torch.cuda.synchronize()
from siu3r_amd.metrics import MetricAccumulator
acc = MetricAccumulator(num_classes=3)
vec = torch.as_tensor(acc.to_vector(), dtype=torch.float64)  # the evaluator's per-rank vector (9 + 12 C doubles)
out = [torch.empty_like(vec.to(dev))]
dist.all_gather(out, vec.to(dev))
assert torch.equal(out[0].cpu(), vec)
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_collective_helpers_on_rccl_with_one_rank():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CODE], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
