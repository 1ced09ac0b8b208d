"""Two REAL ranks of klang_amd.ShardedSynthBank with the HIP bank: one process per rank, the global event
stream given to both, one all-reduce of the [2][n] block per step — against a single-process bank with the same events.  With two GPUs the
ranks take cuda:0 / cuda:1 and the `nccl` (RCCL) backend; on a one-GPU box both ranks share cuda:0 and reduce through `gloo`
(KLG_BENCH_ONE_GPU's arrangement).  The summation order differs (two partial mixes), hence the mix tolerance of test_gpu_parity."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import klang_amd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
two = torch.cuda.device_count() >= 2
dev = rank if two else 0
torch.cuda.set_device(dev)
dist.init_process_group("nccl" if two else "gloo", device_id=torch.device("cuda", dev) if two else None)
S, P, N, B = 5, 16, 256, 12                                     # 5 synth instances: an uneven 3 + 2 split
bank = klang_amd.ShardedSynthBank(%(patch)r, S, P, max_block=N, rank=rank, world=world, device=dev)
rng = np.random.default_rng(11)
events = [(int(rng.integers(0, 4)), int(rng.integers(0, S)), int(rng.integers(40, 90)), float(rng.uniform(0.3, 1.0)), int(rng.integers(1, 1 << 30))) for _ in range(40)]
out = np.zeros((B, 2, N), np.float32)
for b in range(B):
    for (at, sy, p, vel, seed) in events:
        if at == b: bank.note_on(sy, p, vel, seed=seed)
        if at + 5 == b: bank.note_off(sy, p)
    if b == 6 and %(patch)r == "supersaw": bank.set_control(1, 0, 0.5)
    mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()                               # (no stream given = the bank's own, non-blocking stream: torch's fill on the default stream is not ordered with it)
    bank.bank.process_device(mix.data_ptr(), N, None)
    torch.cuda.synchronize()
    if two:
        dist.all_reduce(mix)
    else:
        host = mix.cpu(); dist.all_reduce(host); mix = host
    out[b] = mix.cpu().numpy()
if rank == 0: np.save(sys.argv[1], out)
dist.barrier(); bank.close(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("patch", ["sub2a", "supersaw", "fm4"])
def test_two_ranks_with_the_hip_bank_equal_one_bank(patch, tmp_path):
    import klang_amd
    script = tmp_path / "rank.py"
    script.write_text(RANK % {"root": ROOT, "patch": patch})
    res = tmp_path / "mix.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29577",
                        str(script), str(res)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(res)
    # the same events on ONE bank
    S, P, N, B = 5, 16, 256, 12
    bank = klang_amd.SynthBank(patch, synths=S, notes=P, max_block=N)
    rng = np.random.default_rng(11)
    events = [(int(rng.integers(0, 4)), int(rng.integers(0, S)), int(rng.integers(40, 90)), float(rng.uniform(0.3, 1.0)), int(rng.integers(1, 1 << 30))) for _ in range(40)]
    want = np.zeros((B, 2, N), np.float32)
    for b in range(B):
        for (at, sy, p, vel, seed) in events:
            if at == b:
                bank.random(seed); bank.note_on(sy, p, vel)
            if at + 5 == b:
                bank.note_off(sy, p)
        if b == 6 and patch == "supersaw":
            bank.set_control(1, 0, 0.5)
        blk = np.zeros((2, N), np.float32)
        bank.process(blk)
        want[b] = blk
    bank.close()
    peak = float(np.max(np.abs(want)))
    assert peak > 0.05
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-5 * peak * np.sqrt(S * P) * 4


def test_bench_py_two_ranks_end_to_end():
    """`python bench.py --gpus 2`: bench.py spawns its own ranks (torch.distributed.run on 127.0.0.1), every rank builds its shard and its
    event script, blocks go through the ring of four buffers with one asynchronous all-reduce each, rank 0 prints ONE JSON line.  On a
    one-GPU box KLG_BENCH_ONE_GPU=1 puts both ranks on cuda:0 over gloo — a functional run of the N > 1 code path as a whole, not a measurement."""
    import json
    import torch
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env["KLG_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--voices", str(375 * 256)],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    assert np.isfinite(out["value"]) and out["value"] > 0
    assert out["config"]["voices_alive_after_last_block"] == out["config"]["voices_alive_expected"]
    assert out["config"]["mix_checksum"] > 0
    assert "all-reduce" in out["config"]["parallelism"]
    # the line proves what it ran on: ranks counted THROUGH the communicator, the devices they sit on, the collective timed by itself, every rank's kernel,
    # and the efficiency against the one-GPU value of the same invocation
    assert out["rccl_ranks_seen"] == 2 and out["backend"] in ("nccl", "gloo")
    assert out["functional_test_only"] == (env.get("KLG_BENCH_ONE_GPU") == "1") and out["devices_distinct"] == (1 if out["functional_test_only"] else 2)
    assert out["allreduce_us_per_block"] > 0 and len(out["per_rank_kernel_ms"]) == 2 and all(ms > 0 for ms in out["per_rank_kernel_ms"])
    assert len(out["value_one_gpu_same_invocation"]) == 2 and 0 < out["weak_scaling_efficiency"] < 1.5


def test_bench_py_eight_ranks_end_to_end():
    """`python bench.py --gpus 8` as the driver's scaling run launches it, end to end, before there is an 8-GPU node to run it on: eight ranks (on a box with
    fewer GPUs all on cuda:0 over gloo, KLG_BENCH_ONE_GPU=1: a functional run of the 8-way path, not a measurement), eight shards with their scripts, the
    ring of four buffers, one asynchronous all-reduce per block, ONE JSON line from rank 0 whose aggregate counts all eight shards."""
    import json
    import torch
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 8:
        env["KLG_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--voices", str(375 * 256)],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 4 and out["scaling"] == "weak"
    assert np.isfinite(out["value"]) and out["value"] > 0
    assert out["config"]["voices_alive_after_last_block"] == out["config"]["voices_alive_expected"]
    assert out["config"]["mix_checksum"] > 0 and "x8" in out["config"]["parallelism"]



def test_bench_py_in_library_mode():
    """`python bench.py --gpus 4 --in-library`: ONE process, klg_init(ids), the bank sharded inside the library, the shards' blocks combined by the library's own
    reduction (RCCL between distinct GPUs; a device-side add where KLG_BENCH_ONE_GPU=1 puts every shard on cuda:0: the functional run a one-GPU box can make)."""
    import json
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 4:
        env["KLG_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--in-library", "--steps", "6", "--warmup", "2", "--voices", str(375 * 512)],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["mode"] == "in-library" and out["steps"] == 6 and np.isfinite(out["value"]) and out["value"] > 0
    assert out["config"]["mix_checksum"] > 0 and "x4" in out["config"]["parallelism"]
    one = env.get("KLG_BENCH_ONE_GPU") == "1"
    assert out["shards"] == 4 and out["functional_test_only"] == one and out["devices_distinct"] == (1 if one else 4) and out["rccl_ranks_seen"] == (0 if one else 4)
    assert len(out["per_rank_kernel_ms"]) == 4 and all(ms > 0 for ms in out["per_rank_kernel_ms"]) and out["value_one_gpu_same_invocation"] > 0 and 0 < out["weak_scaling_efficiency"] < 1.5
    assert (out["allreduce_us_per_block"] == 0) == one                                       # (shards that share a GPU are added on the device: there is no collective to time)


def test_bench_py_refuses_to_report_more_gpus_than_it_ran_on():
    """--gpus 2 on a one-GPU box without the functional-test switch: no JSON line, a message — a scaling number is never made from ranks that share a GPU."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    env = {k: v for k, v in os.environ.items() if k != "KLG_BENCH_ONE_GPU"}
    for extra in ([], ["--in-library"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--voices", str(375 * 256)] + extra,
                           env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], (r.stdout[-500:], r.stderr[-500:])
        assert "GPU(s)" in r.stderr


FX_RANK = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import klang_amd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
two = torch.cuda.device_count() >= 2
dev = rank if two else 0
torch.cuda.set_device(dev)
dist.init_process_group("gloo")                                 # (only for the rendezvous and the final barrier: the effect path has no collective)
K, N, B = 37, 256, 10                                           # 37 instances: 19 + 18
bank = klang_amd.ShardedFxBank(%(patch)r, K, max_block=N, rank=rank, world=world, device=dev)
rng = np.random.default_rng(23)
dials = [(int(rng.integers(0, B)), int(rng.integers(0, K)), int(rng.integers(0, 3)), float(rng.uniform(.1, .9))) for _ in range(30)]
out = np.zeros((B, bank.hi - bank.lo, 2, N), np.float32)
ts = torch.cuda.Stream()
with torch.cuda.stream(ts):
    for b in range(B):
        for (at, k, i, v) in dials:
            if at == b: bank.set_control(k, i, v)
        io = (rng.random((K, 2, N), dtype=np.float32) - 0.5).astype(np.float32)
        d = torch.from_numpy(np.ascontiguousarray(bank.local(io))).cuda()
        bank.process_device(d.data_ptr(), N, ts.cuda_stream)
        ts.synchronize()
        out[b] = d.cpu().numpy()
np.save(sys.argv[1] + f".{rank}.npy", out)
dist.barrier(); bank.close(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("patch", ["pingpong", "reverb"])
def test_two_ranks_of_effect_instances_equal_one_bank(patch, tmp_path):
    """SURVEY §8e for the effect banks: instances split 19 + 18 over two ranks, no collective; row for row the two ranks' output is the one bank's, bit for bit."""
    import klang_amd
    script = tmp_path / "fx_rank.py"
    script.write_text(FX_RANK % {"root": ROOT, "patch": patch})
    res = tmp_path / "fx"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29578",
                        str(script), str(res)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.concatenate([np.load(f"{res}.{rk}.npy") for rk in range(2)], axis=1)
    K, N, B = 37, 256, 10
    bank = klang_amd.FxBank(patch, K, max_block=N)
    rng = np.random.default_rng(23)
    dials = [(int(rng.integers(0, B)), int(rng.integers(0, K)), int(rng.integers(0, 3)), float(rng.uniform(.1, .9))) for _ in range(30)]
    for b in range(B):
        for (at, k, i, v) in dials:
            if at == b:
                bank.set_control(k, i, v)
        io = (rng.random((K, 2, N), dtype=np.float32) - 0.5).astype(np.float32)
        bank.process(io)
        assert np.array_equal(got[b].view(np.uint32), io.view(np.uint32)), (patch, b)
    assert float(np.abs(got).max()) > 0.05
    bank.close()
