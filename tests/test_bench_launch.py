"""bench.py --gpus N: the launch path (N rank processes, 127.0.0.1 rendezvous, one JSON line from rank 0).

CPU tests drive the real launcher with the hot path switched off (MPSR_BENCH_RENDEZVOUS_ONLY: gloo); the GPU test runs
two ranks of the real step on one device (MPSR_BENCH_SHARE_GPU: ranks meet over gloo, both compute on cuda:0).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks(n):
    r = _run(["--gpus", str(n)], {"MPSR_BENCH_RENDEZVOUS_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # rank 0 only
    assert lines[0]["n_gpus"] == n and lines[0]["max_rank_plus_1"] == n  # every rank took part


def test_under_a_launcher_each_process_is_one_rank():
    # what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` does to each process
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, MPSR_BENCH_RENDEZVOUS_ONLY="1")
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    assert [len(_json_lines(o[0])) for o in outs] == [1, 0]
    assert _json_lines(outs[0][0])[0]["n_gpus"] == 2


def test_gpus_must_match_the_launcher():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MPSR_BENCH_RENDEZVOUS_ONLY": "1"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_a_failed_rank_fails_the_launch():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU: the ranks fail with 'needs a GPU'")
    r = _run(["--gpus", "2"], {})
    assert r.returncode != 0 and "rank(s) failed" in r.stderr


@pytest.mark.gpu
def test_two_ranks_of_the_real_step_on_one_gpu():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--points", "256", "--no-roofline",
              "--cpu-sample", "0"], {"MPSR_BENCH_SHARE_GPU": "1"}, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    res = lines[0]
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["global_batch"] == 16
    assert res["value"] > 0 and abs(res["value"] - 16 * 2 / (res["ms_per_step"] * 2e-3)) < 0.01 * res["value"]
