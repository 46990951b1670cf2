"""The committed evidence must be self-consistent (r03 review, item 5): bench.py's `roofline.frac` -- executed
multiply-adds of the replayed matrix-pipe launches / launch time / peak -- has to agree with the in-situ PMC figure
(SQ_VALU_MFMA_BUSY_CYCLES over the same kernel set, tools/pmc_reduce.py) collected at the same commit, and the timed
replay must not exceed the step it replays.  CPU-only: reads the newest round's files under profiles/."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest(pattern):
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert paths, pattern
    return paths[-1]


def _line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_roofline_frac_agrees_with_the_pmc_busy_fraction():
    bench = _newest("r*_bench_default_run.json")
    rnd = re.match(r"(r\d+)_", os.path.basename(bench)).group(1)
    pmc = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd)))
    r = _line(bench)
    roof = r["roofline"]
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] <= 1.0
    # (both are fractions of TIME at the nominal clock: GRBM_GUI_ACTIVE advances at a fixed ~2.45 GHz -- tools/pmc_reduce.py)
    ins = roof.get("in_situ")
    if isinstance(ins, dict) and "frac_in_situ" in ins:
        # r06: the line carries the step's own kernel records (torch.profiler).  Traced records and the counter collection
        # are the same kind of measurement (every launch under a tracer) and must agree; the event-timed replay runs the
        # same launches warm and back to back and sits between the traced figure and the one scaled to the untraced step
        # (0.03: the counter passes read their counters around every launch on top of the tracing -- r06's committed pair:
        # 0.7605 traced, 0.7362 counted, 0.7788 replayed, 0.791 scaled; r05's cold replay against its counters: 0.745 / 0.747)
        assert abs(ins["frac_in_situ"] - pmc["mfma_busy"]) <= 0.03, (ins["frac_in_situ"], pmc["mfma_busy"])
        # (no ordering between the two: the counter passes are 3-step runs on a board that has not reached its
        # power-limited clock yet, the traced steps follow the full timed loop -- the round's four collections read
        # 0.7605 / 0.7362, 0.7556 / 0.7508, 0.752 / 0.7413, 0.7423 / 0.7509, 0.7533 / 0.7426 and 0.7543 / 0.7473 traced / counted)
        assert ins["frac_in_situ"] - 0.01 <= roof["frac"] <= ins["frac_in_situ_scaled_to_untraced"] + 0.01, (roof["frac"], ins)
        assert 0.9 < ins["matrix_share_of_kernel_time"] < 1.0 and ins["matrix_launches_per_step"] == roof["launches_per_step"]
    else:
        assert abs(roof["frac"] - pmc["mfma_busy"]) <= 0.02, (roof["frac"], pmc["mfma_busy"])
    assert roof["kernel_ms_per_step"] <= r["ms_per_step"]
    assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    # the kernels the line names are the kernels the PMC collection saw
    for k in ("pw_conv_kernel", "wino3", "wino4_conv_kernel"):
        assert k in roof["kernel"] and any(name.startswith(k) for name in pmc["kernels"])


def test_bench_line_carries_the_contract_and_the_round_objects():
    r = _line(_newest("r*_bench_default_run.json"))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["dtype"] == "f32" and r["vs_baseline"] is None and "workload" in r["config"]
    assert abs(r["value"] - r["config"]["global_batch"] * 1e3 / r["ms_per_step"]) / r["value"] < 1e-3
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    if "cfg5_step" in r:
        assert r["cfg5_step"]["ms_per_step"] > r["ms_per_step"]  # forward + EMD loss costs more than forward + Chamfer
