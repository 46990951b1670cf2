"""RCCL on the hardware that exists (r05 review, "What's missing" 1): a ONE-rank "nccl" process group on the lease's GPU
runs the N > 1 step's real collective sequence -- ReverseBucketReducer in both modes through the
`dist.get_backend() == "nccl"` branch of monopsr_amd/core/trainer.py (reduce_scatter_tensor followed at once by
all_gather_into_tensor on RCCL's stream, async) over the full both-trunk flat gradient, a training step through it,
gather_instances / reduce_metric_sums -- in a child process (tools/rccl_selftest.py), so that this pytest process never
holds a process group.  The gloo tests (tests/test_trainer_reducer_gloo.py, world sizes 2-3) cover the arithmetic of
N > 1; this one covers the library, the branch and the sizes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_runs_the_n_rank_collective_sequence():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selftest.py")], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    print(json.dumps(out))
    assert out["backend"] == "nccl" and out["world_size"] == 1 and out["rccl_version"]
    # the both-trunk buffer: >= the reference's 100,204,832 trainable parameters (401 MB), padded layout
    assert out["params"] >= 100204832 and out["grad_bytes"] == 4 * out["params"] >= 400819328
    nb = out["reducer_rccl"]["buckets"]
    assert nb == -(-out["grad_bytes"] // (64 << 20))
    assert out["reducer_rccl"]["issued"] == {"all_reduce": nb}
    # "direct": every bucket divides by a world size of one -> scatter + gather per bucket, gather right behind scatter
    assert out["reducer_direct"]["issued"] == {"all_gather_into_tensor": nb, "reduce_scatter_tensor": nb}
    assert out["reducer_direct"]["issue_order_head"][:2] == ["reduce_scatter_tensor", "all_gather_into_tensor"]
    for mode in ("rccl", "direct"):
        rec = out["reducer_" + mode]
        assert rec["buffer_bit_identical"], mode  # one rank: sum == identity, average divides by one
        assert rec["grads_finite"] and all(v == v and abs(v) < 1e30 for v in rec["loss"]), (mode, rec)
        assert sum(rec["step_issued"].values()) == (nb if mode == "rccl" else 2 * nb)
        # same step, same weights, with and without the collectives: equal up to the atomics' summation order
        assert rec["grad_max_rel_diff_vs_no_collectives"] < 1e-3, (mode, rec)
        # the trainer probed the hardware queues (r06): the weight-gradient stream it kept is one a waiting collective does
        # not hold up (one of four candidates; every rank makes the same five probe calls)
        hq = rec["hardware_queues"]
        assert len(hq["wgrad_stream_candidates_held_up"]) == 4
        assert hq["wgrad_stream_candidates_held_up"][hq["wgrad_stream_picked"]] is False
        assert hq["caller_stream_held_up_by_collectives"] in (True, False)
    assert out["gather_instances_identity"] and out["reduce_metric_sums"] == [1.5, 2.0]
