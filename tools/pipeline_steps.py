"""Experiment: consecutive benchmark steps (independent batches of 256) issued alternately on S HIP streams, each with
its own scratch, so that one step's launch edges and small kernels overlap another step's matrix work.
    python tools/pipeline_steps.py [--streams 1,2,3] [--steps 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from monopsr_amd.core import device_net as dn  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,3")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    net = dn.DeviceNet(W.synthetic_weights(seed=0), device=dev)
    inp, _ = bench.make_inputs(args.batch, 1024, 0, dev)
    for rnd in range(2):
        for S in [int(s) for s in args.streams.split(",")]:
            steps = [bench.Step(net.clone_with_own_scratch() if i else net, inp, 1024) for i in range(S)]
            streams = [torch.cuda.Stream() for _ in range(S)]
            for st, sm in zip(steps, streams):
                with torch.cuda.stream(sm):
                    st()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                with torch.cuda.stream(streams[k % S]):
                    steps[k % S]()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("streams %d: %.2f ms/step  %.0f crops/s" % (S, 1e3 * dt / args.steps, args.batch * args.steps / dt))


if __name__ == "__main__":
    main()
