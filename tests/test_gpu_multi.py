"""The multi-GPU code paths on the one-GPU box: RCCL (backend nccl) with a one-rank group, bench.py's own rank spawning, and
the bench line's contract fields.  N > 1 ranks on separate GPUs are the driver's to run (SCALE); these tests make sure that what
it will launch has executed before: process-group setup with device_id, all-gather / all-reduce / barrier on GPU tensors through
RCCL, shard.make_ba_exchange's nccl branch, and `bench.py --gpus N` starting N ranks by itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_NCCL_WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import numpy as np, torch, torch.distributed as dist
import orbhip, shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
# shard.make_ba_exchange, nccl branch: slot `rank` of the exchange buffer all-gathered in place through RCCL
stride = 1000
xbuf = torch.zeros(stride, dtype=torch.float64, device="cuda")
xbuf[:600] = torch.arange(600, dtype=torch.float64, device="cuda") * 0.5
ex = shard.make_ba_exchange(xbuf, stride)
ex(2, 600)
assert torch.equal(xbuf[:600].cpu(), torch.arange(600, dtype=torch.float64) * 0.5) and float(xbuf[600:].abs().sum()) == 0.0
# the ORB path's only exchange: fixed-size per-frame records, and the max over ranks of the timed region
rec = torch.arange(24, dtype=torch.int32, device="cuda").view(12, 2)
allrec = shard.allgather_records(rec)
assert tuple(allrec.shape) == (1, 12, 2) and torch.equal(allrec[0], rec)
assert shard.max_over_ranks(3.25, device="cuda") == 3.25
dist.barrier()
# a real extraction next to the live process group (one HIP runtime serves torch, RCCL and liborbhip)
ctx = orbhip.Context(0); ext = orbhip.Extractor(ctx, 500, 1.2, 8, 20, 7)
r = ext.extract_host(orbhip.synth_frames(320, 240, 2, seed=3), (0, 0))
assert len(r[0][0]) > 300
ext.close(); ctx.close()
dist.destroy_process_group()
print("NCCL_OK")
"""


def _env(**kw):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(kw)
    return e


def test_one_rank_nccl_group_runs_the_rccl_paths(tmp_path):
    script = tmp_path / "w.py"
    script.write_text("ROOT = %r\n" % ROOT + _NCCL_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=_env(), timeout=400)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, r.stdout[-3000:]


SMALL = ["--steps", "2", "--warmup", "1", "--batch", "32", "--ba-graphs", "4", "--ba-steps", "1", "--pose-frames", "32",
         "--stereo-pairs", "8", "--no-cpu-baseline"]


def _line(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-3000:]
    return json.loads(lines[0])


def test_bench_one_rank_over_rccl():
    """bench.py with the process group forced on for one rank: init_process_group("nccl", device_id=...), the rank census, barriers,
    max-over-ranks and the record all-gather all run through RCCL on this box's GPU."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=_env(ORBHIP_BENCH_FORCE_DIST="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == [0] and d["config"]["collective_backend"] == "nccl"
    assert d["config"]["records_gathered"] == 32 and d["value"] > 0 and d["ba"]["value"] > 0
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert d["roofline"]["kernel"].startswith("k_") and 0 < d["roofline"]["frac"] < 1


def test_bench_gpus2_spawns_two_ranks_itself():
    """`python bench.py --gpus 2` with no torch.distributed environment must start two ranks by itself (the driver's SCALE command
    shape).  Rehearsal on one card: both ranks pinned to GPU 0, collectives over gloo; the line must say n_gpus 2, list both ranks
    and count both ranks' records."""
    env = _env(ORBHIP_BENCH_BACKEND="gloo", ORBHIP_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ba-sharded-graphs", "2"] + SMALL,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == [0, 1] and d["config"]["frames_total"] == 64
    assert d["config"]["records_gathered"] == 64 and d["scaling"] == "weak"
    assert "error" not in d["ba_sharded"] and d["ba_sharded"]["ranks"] == 2 and d["ba_sharded"]["all_gathers"] > 3
    assert "cpu_baseline" not in d
