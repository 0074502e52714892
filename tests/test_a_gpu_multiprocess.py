"""Runs FIRST among the -m gpu tests (file name order), before this pytest process has touched the GPU: it starts
fresh child processes -- one per rank, all on the box's single GPU -- that render interleaved tiles through the C
ABI and gather them with rustraytracer_amd.dist.FilmGather over gloo (tools/mp_film_check.py).  This is the
multi-rank path of bench.py --gpus N run as a system, minus RCCL (a one-GPU box cannot host two RCCL ranks)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_on_one_gpu_gather_the_one_rank_film(ranks):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + 17 * ranks),
           os.path.join(ROOT, "tools", "mp_film_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["ranks"] == ranks and rec["bit_identical"] and rec["rays_sum"] == rec["rays_one_rank"]


def _gpu_count():
    # torch.cuda.device_count() does not initialise the GPU in this process (it reads the driver's device list)
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
def test_rccl_comes_up_with_one_rank():
    """What a one-GPU box can say about the RCCL side of bench.py --gpus N: the nccl backend of this image initialises a
    communicator on the GPU, an all_reduce on device tensors runs through it, and the rank script ends cleanly (one rank:
    FilmGather has nothing to exchange).  The 2 / 4 / 8-rank versions below need that many GPUs."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29689",
           os.path.join(ROOT, "tools", "mp_film_check.py"), "--backend", "nccl"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["ranks"] == 1 and rec["backend"] == "nccl" and rec["bit_identical"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_ranks_on_distinct_gpus_gather_over_rccl(ranks):
    """Arms itself on a multi-GPU box: one fresh process per GPU, backend nccl (= RCCL over xGMI), FilmGather on device
    tensors -- the path bench.py --gpus N takes (render.rs:49-71's tile farm).  Skipped on a one-GPU box."""
    if _gpu_count() < ranks:
        pytest.skip(f"needs {ranks} GPUs, this box has {_gpu_count()}")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + 13 * ranks),
           os.path.join(ROOT, "tools", "mp_film_check.py"), "--backend", "nccl"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["ranks"] == ranks and rec["backend"] == "nccl" and rec["bit_identical"]
    assert rec["rays_sum"] == rec["rays_one_rank"]
