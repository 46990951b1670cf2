"""A/B of the bench step under DeviceNet.forward_instances' host-side options, interleaved in one process:
filter cache off / on (mpsr_net_opts.filter_cache) x heads on the main stream / on a second stream next to the map decoder.
    python tools/step_opts_ab.py [--rounds 5] [--steps 10]
Also checks that all variants give bit-identical outputs."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from monopsr_amd.core import device_net as dn  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
device = torch.device("cuda", 0)
net = dn.DeviceNet(W.synthetic_weights(seed=0), device=device)
inp, _ = bench.make_inputs(args.batch, 1024, 0, device)
step = bench.Step(net, inp, 1024)
step()
caches = dict(net.fcache)


class NoCache:
    def opts(self, key=None, event=None):
        from monopsr_amd import _lib
        o = _lib.NetOpts()
        o.ready_event = event
        return o


hi = torch.cuda.Stream(priority=-1)  # overlap == 2: the step on a high-priority stream, the heads' stream at normal priority


def configure(cache, overlap):
    net.fcache = dict(caches) if cache else {k: NoCache() for k in caches}
    step.overlap_heads = overlap > 0
    step.run_on = hi if overlap == 2 else None


_call = bench.Step.__call__
_fwd = bench.Step.forward_net


def _on(fn):
    def wrapped(self, *a):
        s = getattr(self, "run_on", None)
        if s is None:
            return fn(self, *a)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            r = fn(self, *a)
        torch.cuda.current_stream().wait_stream(s)
        return r
    return wrapped


bench.Step.__call__ = _on(_call)
variants = [(1, 0), (1, 1), (1, 2), (0, 0)]
ref = None
for v in variants:
    configure(*v)
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    step.run_on = None
    xyz, heads = step.forward_net()
    torch.cuda.synchronize()
    got = (xyz.clone(), heads["centroids"].clone(), heads["alpha_bins"].clone())
    if ref is None:
        ref = got
    else:
        assert all(torch.equal(a, b) for a, b in zip(ref, got)), "variant %r changes the outputs" % (v,)
samples = {v: [] for v in variants}
for _ in range(args.rounds):
    for v in variants:
        configure(*v)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        samples[v].append((time.perf_counter() - t0) / args.steps * 1e3)
for v in variants:
    s = sorted(samples[v])
    print("filter cache %d, heads on second stream %d (2: + the step on a high-priority stream): %.3f ms per step (median of %d; min %.3f)"
          % (v[0], v[1], s[len(s) // 2], len(s), s[0]))
print("outputs bit-identical across the variants")
