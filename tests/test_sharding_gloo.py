"""N > 1 path on CPU: two processes over gloo (world_size 2) shard a batch by rows, transform their shards with
no data-path collective, and agree with the oracle on the gathered result; the timing/checksum reductions that
bench.py performs over RCCL are exercised over gloo.  The per-rank compute stand-in is the kernel-body emulator
(tests/emu) because there is no GPU here; on the GPU box the same code path runs the HIP library."""
import os
import socket
import subprocess

import numpy as np
import pytest

from helpers import compare_vectors, random_signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


def test_shard_rows_cover_batch_exactly():
    from rustfft_amd.sharding import all_shards, shard_rows

    for batch in (0, 1, 7, 8, 1024, 8192, 1000003):
        for world in (1, 2, 3, 4, 8):
            shards = all_shards(batch, world)
            assert shards[0][0] == 0 and shards[-1][1] == batch
            for (lo, hi), (lo2, hi2) in zip(shards, shards[1:]):
                assert hi == lo2 and lo <= hi and lo2 <= hi2
    assert shard_rows(8192, 8, 3) == (3072, 4096)  # BASELINE config 5: 1024 rows per GPU


def _worker(rank, world, port, n, batch, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rustfft_amd
    from rustfft_amd import _native
    from rustfft_amd.sharding import process_sharded, reduce_max, reduce_sum, shard_rows

    lib = _native.load(os.path.join(EMU_DIR, "libmi355fft_emu.so"))
    planner = rustfft_amd.FftPlannerHip(np.complex64, lib=lib)
    fft = planner.plan_fft_forward(n)
    x = random_signal(n * batch, np.complex64)  # every rank derives the same synthetic batch from the seed
    lo, hi = shard_rows(batch, world, rank)
    mine = x[lo * n:hi * n].copy()
    dist.barrier()
    process_sharded(fft, mine)
    dist.barrier()
    elapsed = reduce_max(0.001 * (rank + 1), dist)
    checksum = reduce_sum(float(np.abs(mine).sum()), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, mine))
    if rank == 0:
        full = np.concatenate([g[2] for g in sorted(gathered, key=lambda t: t[0])])
        np.save(os.path.join(out_dir, "full.npy"), full)
        np.save(os.path.join(out_dir, "meta.npy"), np.array([elapsed, checksum]))
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_batch(tmp_path, oracle):
    import torch.multiprocessing as mp

    subprocess.check_call(["make", "-C", EMU_DIR, "-j", "8", "-s"])
    n, batch, world = 4096, 5, 2  # ragged: 3 + 2 rows
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n, batch, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    elapsed, checksum = np.load(tmp_path / "meta.npy")
    x = random_signal(n * batch, np.complex64)
    want = x.copy()
    oracle.plan(np.complex64, n, 0).process(want)
    assert compare_vectors(want, full)
    assert elapsed == pytest.approx(0.002)  # MAX over ranks
    assert checksum == pytest.approx(float(np.abs(full).sum()), rel=1e-6)


def _edge_worker(rank, world, port, n, batch, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rustfft_amd
    from rustfft_amd import _native, sharding

    sharding._MAX_MSG_ELEMS = 3000  # several point-to-point operations per peer, ragged last one
    lib = _native.load(os.path.join(EMU_DIR, "libmi355fft_emu.so"))
    fft = rustfft_amd.FftPlannerHip(np.complex64, lib=lib).plan_fft_forward(n)

    class NumpyFft:  # the emulator takes numpy slices; the edges move torch tensors
        def process(self, t):
            a = t.numpy()
            fft.process(a)

    full = torch.from_numpy(random_signal(n * batch, np.complex64)) if rank == 0 else None
    out, times = sharding.process_from_root(NumpyFft(), full, n, batch, dist, root=0, device="cpu", dtype=torch.complex64)
    assert set(times) == {"scatter_s", "compute_s", "gather_s"}
    if rank == 0:
        np.save(os.path.join(out_dir, "edge.npy"), out.numpy())
    dist.destroy_process_group()


def test_scatter_transform_gather_from_one_rank(tmp_path, oracle):
    """SURVEY section 8(e) edges: the batch originates on rank 0, rows go out with grouped send/recv, every rank transforms
    its rows, rows come back -- world size 2 over gloo, ragged shards (3 + 2 rows), multi-chunk messages."""
    import torch.multiprocessing as mp

    subprocess.check_call(["make", "-C", EMU_DIR, "-j", "8", "-s"])
    n, batch, world = 2048, 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_edge_worker, args=(world, port, n, batch, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "edge.npy")
    want = random_signal(n * batch, np.complex64)
    oracle.plan(np.complex64, n, 0).process(want)
    assert compare_vectors(want, got)


def test_bench_self_spawn_launcher():
    """`python bench.py --gpus 2` without torchrun starts one process per rank with the torchrun environment contract;
    --selftest-spawn runs that launcher on CPU (gloo): every rank joins the group and rank 0 prints ONE JSON line."""
    import json
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-spawn"], capture_output=True, text=True,
                         timeout=300, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # (gloo itself chats on stdout)
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d == {"selftest": "spawn", "n_gpus": 2, "rank_sum": 1, "local_ranks": [0, 1]}
