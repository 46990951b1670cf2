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


@pytest.mark.parametrize("n", [2, 3, 8])  # (8: the node the driver's scaling run uses)
def test_gpus_n_starts_n_ranks(n):
    r = _run(["--gpus", str(n)], {"MPSR_BENCH_RENDEZVOUS_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # rank 0 only
    res = lines[0]
    assert res["n_gpus"] == n and res["max_rank_plus_1"] == n  # every rank took part
    # the objects an N > 1 line carries, produced here by the same functions on stand-in workloads
    proof = res["rank_proof"]
    assert proof["allreduce_of_ones"] == n and proof["world_size"] == n
    assert [r["rank"] for r in proof["ranks"]] == list(range(n)) and len({r["pid"] for r in proof["ranks"]}) == n
    assert res["per_rank_ms"] == [1.0 + r for r in range(n)]
    ts = res["training_step"]
    for key in ("ms_per_step", "ms_per_step_without_allreduce", "exposed_allreduce_ms", "allreduce_alone_ms",
                "gradients", "allreduce", "loss_per_step"):
        assert key in ts, key
    assert ts["ranks"] == n and "per-rank batch statistics" in ts["what"]
    # the stand-in gradient is rank + 1 everywhere: averaged over ranks by the reducer
    assert ts["loss_per_step"][-1] == pytest.approx(sum(range(1, n + 1)) / n, abs=0.06)
    emd = res["emd"]["all_ranks"]
    assert emd["ranks_ok"] == n and emd["fused_clouds_per_s_sum"] == 10.0 * sum(range(1, n + 1))


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


def test_a_rank_that_dies_early_does_not_leave_its_peers_waiting():
    # rank 1 exits before the rendezvous; rank 0 would sit in it until the store timeout (the launcher
    # must poll every rank and stop the others on the first failure)
    import time
    t0 = time.monotonic()
    r = _run(["--gpus", "2"], {"MPSR_BENCH_RENDEZVOUS_ONLY": "1", "MPSR_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0 and "(1, 7)" in r.stderr and "stopped" in r.stderr
    assert time.monotonic() - t0 < 90


def test_a_hanging_extra_does_not_cost_the_headline():
    # once `value` exists a watchdog is armed: extras that do not finish by the deadline are dropped, the line is
    # printed as it stands (marked), every rank exits 0
    r = _run(["--gpus", "2"], {"MPSR_BENCH_RENDEZVOUS_ONLY": "1", "MPSR_BENCH_TEST_HANG_EXTRAS": "1",
                               "MPSR_BENCH_TEST_DEADLINE": "3"}, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["extras_timed_out_after_s"] == 3.0
    assert "rank_proof" not in lines[0]


def test_a_rank_dying_during_the_extras_does_not_cost_the_headline():
    #  a non-zero rank that dies while the extras run makes the launcher SIGTERM the others: rank 0, whose
    # headline is already measured, prints it (marked) from its signal handler instead of dying silently
    import signal
    import time
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MPSR_BENCH_RENDEZVOUS_ONLY="1", MPSR_BENCH_TEST_HANG_EXTRAS="1", MPSR_BENCH_TEST_DEADLINE="300",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "1"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    try:
        time.sleep(12)  # rendezvous + headline, then the extras "hang"
        assert p.poll() is None
        p.send_signal(signal.SIGTERM)
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    lines = _json_lines(out)
    # (the status says the run was terminated -- 128 + SIGTERM -- while the headline line still comes out)
    assert p.returncode == 128 + signal.SIGTERM and len(lines) == 1, (p.returncode, out[-500:], err[-1000:])
    assert lines[0]["extras_incomplete"] == "terminated_by_signal_15" and "emitting it" in err


def test_sigterm_to_the_launcher_stops_the_ranks():
    import signal
    import time
    import psutil
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MPSR_BENCH_TEST_HANG="1")
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    try:
        kids = []
        for _ in range(600):  # (up to 30 s: the interpreter start of the launcher can be slow on a loaded host)
            kids = psutil.Process(p.pid).children()
            if len(kids) == 2:
                break
            time.sleep(0.05)
        assert len(kids) == 2
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=30)
        assert p.returncode != 0
        gone, alive = psutil.wait_procs(kids, timeout=15)
        assert not alive
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.gpu
def test_two_ranks_of_the_real_step_on_one_gpu():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--points", "256", "--cpu-sample", "0"],
             {"MPSR_BENCH_SHARE_GPU": "1"}, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    res = lines[0]
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["global_batch"] == 16
    assert res["value"] > 0 and abs(res["value"] - 16 * 2 / (res["ms_per_step"] * 2e-3)) < 0.01 * res["value"]
    # the four N > 1 objects: rank proof, per-rank times, the real-gradient training step through the reducer, and
    # the EMD share of every rank
    assert res["rank_proof"]["allreduce_of_ones"] == 2 and len(res["rank_proof"]["ranks"]) == 2
    assert len(res["per_rank_ms_per_step"]) == 2 and max(res["per_rank_ms_per_step"]) <= res["ms_per_step"] * 1.05
    ts = res["training_step"]
    assert "error" not in ts, ts
    # N > 1 default (r06): the both-trunk net -- SURVEY 8(d) cfg4's 100 M-parameter / 401 MB gradient is what the reducer
    # carries (the reference trains both ResNet-101 trunks: net_builder.py:44-52, core/trainer.py:71-81)
    assert ts["ranks"] == 2 and ts["params"] >= 100204832 and ts["grad_bytes"] >= 400819328 and ts["ms_per_step"] > 0
    assert "both ResNet-101 trunks" in ts["trainable"]
    assert "exposed_allreduce_ms" in ts and "ms_per_step_without_allreduce" in ts
    emd = res["emd"]
    assert "error" not in emd and emd["all_ranks"]["ranks_ok"] == 2, emd
    assert 0 < res["roofline"]["frac"] <= 1.0
