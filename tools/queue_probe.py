"""Which of a training rank's streams share a hardware queue?  Creates a one-rank RCCL communicator (or runs under
torch.distributed.run on N GPUs), then reports: is the default stream held up by a waiting collective
(device_net.blocked_by_collectives), is a freshly picked side stream, and what `GPU_MAX_HW_QUEUES` is.
    python tools/queue_probe.py [--streams-first N]   (N: side streams created BEFORE the communicator, as bench.py has)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from monopsr_amd.core import device_net as dn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams-first", type=int, default=0)
    args = ap.parse_args()
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    main_stream = torch.cuda.current_stream(dev)
    early = []
    for _ in range(args.streams_first):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
        early.append(s)
    side_before = dn.concurrent_stream(dev, main_stream) if args.streams_first else None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    dist.all_reduce(torch.ones(1, device=dev))
    torch.cuda.synchronize()
    side_after = dn.concurrent_stream(dev, main_stream)
    rep = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "world": world,
           "streams_created_before_the_communicator": args.streams_first,
           "default_stream_blocked_by_collectives": [dn.blocked_by_collectives(main_stream) for _ in range(3)],
           "side_stream_picked_after_blocked": [dn.blocked_by_collectives(side_after) for _ in range(3)]}
    if side_before is not None:
        rep["side_stream_picked_before_blocked"] = [dn.blocked_by_collectives(side_before) for _ in range(3)]
    cands = [torch.cuda.Stream() for _ in range(8)]
    rep["eight_fresh_streams_blocked"] = [dn.blocked_by_collectives(c) for c in cands]
    if rank == 0:
        out.write(json.dumps(rep) + "\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
