"""CPU suite, part 4: the multi-GPU path (SRS / coefficient sharding with an all_gather of
partial points and a one-element division carry, poly_commit_amd/sharded.py) run as a real
multi-process torch.distributed job on the gloo backend, world_size 2 and 3."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_commit_open_gloo(world):
    import oracle_lib as O
    O.lib()   # build the oracle once, before the ranks race for it
    port = 29650 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_sharded_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("OK") == world


@pytest.mark.gpu
@pytest.mark.parametrize("precompute", [False, True])
def test_sharded_commit_open_device_engine_two_ranks(precompute):
    """The same two-rank job with the DEVICE engine (both ranks on GPU 0, gloo for the exchanges): shard
    evaluation (pc_hip_poly_eval), carry composition, one division scan with the carry, MSMs at the
    shifted base offsets, with and without the SRS window table -- against the single-process oracle."""
    import oracle_lib as O
    O.lib()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29661", os.path.join(HERE, "_sharded_worker.py"),
           "--engine", "hip"] + (["--precompute"] if precompute else [])
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("OK") == 2
