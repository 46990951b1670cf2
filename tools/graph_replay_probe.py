"""Probe of the captured training step (InstanceTrainer.capture_step) on this ROCm / torch stack: a replay of the ~1100-node
HIP graph that directly follows a device-wide synchronisation came back with non-finite weight gradients in 10-30 % of
the cases (forward intact; eager steps never) until _StepGraph.step() put eager launches in front of every replay.
    python tools/graph_replay_probe.py <mode> [reps]
mode is a string of switches: "graph" (captured step; otherwise eager), "restore" (8 steps, then the scenario, then one
more step, instead of 9 plain steps) with "nosave" + one of "synconly" / "sleep" / "sleeponly" / "synckern" /
"syncalloc" / "cpu" (what happens between the eighth step and the ninth) or "saveonly" (tr.save without tr.restore),
"noclone", "spin"; A/B switches of the library: "nothin", "nofc", "nosplit", "nobank", "nowgw", "ungrouped", "nopw",
"nowino".  Set _StepGraph.TICKS = 0 (monopsr_amd/core/trainer.py) to see the failure: e.g.
    python tools/graph_replay_probe.py graph_restore 40              -> 12 bad of 40 without the workaround, 0 of 80 with it
    python tools/graph_replay_probe.py graph_restore_nosave_synconly 40 -> 7 of 40 / 0 of 80"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib
from monopsr_amd.core import config_utils, train_net, trainer
from monopsr_amd.core import weights as W
lib = _lib.lib()
B, div = 4, 4
cfg = config_utils.default_config()
opt = cfg.train_config.optimizer.adam_optimizer
opt.learning_rate_type, opt.initial_learning_rate = 'exponential_decay', 2e-5
opt.decay_steps, opt.decay_factor, opt.staircase = 2, 0.8, True
opt.use_moving_average, opt.moving_average_decay = True, 0.9
def batch(i):
    rng = np.random.default_rng(500 + i)
    y1, x1 = rng.uniform(0, 150, B), rng.uniform(0, 1000, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(20, 200, B), x1 + rng.uniform(20, 200, B)], 1).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s = dict(rgb_image_crops=dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
             full_img_feature_crop=dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0).astype(np.float32)),
             boxes_2d=dev(boxes), cam_p=dev(np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.2], [0, 0, 1, 0.003]], np.float32)),
             est_view_angs=dev(rng.uniform(-0.6, 0.6, (B, 1)).astype(np.float32)), class_indices=dev(np.ones((B, 1), np.int32)),
             mean_lwh=dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))), prop_cen_z_offset=dev(np.full((B,), 2.178, np.float32)))
    s.update(trainer.synthetic_ground_truth(s, seed=600 + i))
    return s
data = [batch(i) for i in range(8)]
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if "noticks" in mode:
    trainer._StepGraph.TICKS = 0
graph = "graph" in mode
if "nothin" in mode: lib.mpsr_debug_set_thin_conv(0)
if "nofc" in mode: lib.mpsr_debug_set_fc_split_rows(0)
if "nosplit" in mode: lib.mpsr_debug_set_wino3z_split(0)
if "nowgw" in mode: lib.mpsr_debug_set_wgrad_winograd(0)
if "ungrouped" in mode: lib.mpsr_debug_set_wgrad_grouped(0)
if "nopw" in mode: lib.mpsr_debug_set_conv_pointwise(0)
if "nowino" in mode: lib.mpsr_debug_set_conv_winograd(0)
bad = 0; runs = 0
KEEP = torch.ones((1 << 20,), device="cuda")
NOCLONE = "noclone" in mode
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    net = train_net.TrainNet(W.synthetic_weights(seed=77, width_div=div), width_div=div, dgrad_bank=("nobank" not in mode))
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config)
    if graph: tr.capture_step(warmup=2)
    if "restore" in mode:
        import tempfile
        for i in range(8):
            float(tr.step(data[i]))
        with tempfile.TemporaryDirectory() as ckpt:
            prefix = tr.save(ckpt) if "nosave" not in mode else None
            if prefix is None:
                if "synckern" in mode:   # sync, then kernel launches on existing memory (no allocation)
                    torch.cuda.synchronize()
                    for _ in range(200):
                        KEEP.mul_(1.0)
                elif "syncalloc" in mode:  # sync, then allocations without kernels
                    torch.cuda.synchronize()
                    junk = [torch.empty((net.params.numel(),), device="cuda") for _ in range(4)]
                    del junk
                elif "synconly" in mode:
                    torch.cuda.synchronize()
                elif "sleeponly" in mode:
                    import time
                    time.sleep(0.6)
                elif "shortsleep" in mode:
                    import time
                    torch.cuda.synchronize()
                    time.sleep(0.02)
                elif "sleep" in mode:
                    import time
                    torch.cuda.synchronize()
                    time.sleep(0.6)
                elif "cpu" in mode:
                    junk = [net.params.cpu(), net.adam_m.cpu(), net.adam_v.cpu()]
                    del junk
                else:
                    junk = [torch.randn((net.params.numel(),), device="cuda") for _ in range(4)]
                    del junk
            if NOCLONE:
                before = pb = mb = vb = KEEP
            else:
                before = tr.optimizer.shadow.clone()
                pb, mb, vb = net.params.clone(), net.adam_m.clone(), net.adam_v.clone()
            ref_cpu = [t.cpu() for t in (before, pb, mb, vb)] if not NOCLONE else []
            ranges = [(t.data_ptr(), t.data_ptr() + t.numel() * 4) for t in (before, pb, mb, vb)]
            if "saveonly" not in mode and prefix is not None:
                tr.restore(prefix)
            same = (True,) if NOCLONE else (torch.equal(net.params, pb), torch.equal(net.adam_m, mb), torch.equal(net.adam_v, vb), torch.equal(tr.optimizer.shadow, before))
            if "spin" in mode:  # keep the GPU busy while the host enqueues the graph
                with torch.cuda.stream(tr._graph.side):
                    torch.cuda._sleep(20_000_000)
            l = float(tr.step(data[0]))
            runs += 1
            fin = [bool(torch.isfinite(x).all()) for x in (net.params, net.adam_m, net.adam_v, tr.optimizer.shadow, net.grads)]
            changed = [not torch.equal(t.cpu(), r) for t, r in zip((before, pb, mb, vb), ref_cpu)] if not NOCLONE else []
            if any(changed):
                print("rep", rep, "clones modified by the replay:", changed)
                g = tr._graph
                for nm, t in [("static:" + k, v) for k, v in g.static.items() if torch.is_tensor(v)] + [("pinned%d" % i, t) for i, t in enumerate(g.pinned)] + [("loss", g.loss), ("lr_t", g.lr_t), ("bank", net.dgrad_bank.flat)]:
                    a0, a1 = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
                    for j, (b0, b1) in enumerate(ranges):
                        if a0 < b1 and b0 < a1:
                            print("   overlap of clone", j, "with", nm, hex(a0), hex(a1))
            if not all(fin) or not np.isfinite(l) or not all(same):
                bad += 1
                g = net.grads
                status = "".join("x" if (not bool(torch.isfinite(L.dw).all()) or (L.db is not None and not bool(torch.isfinite(L.db).all()))) else "." for L in net.layers)
                print("   layers (x = non-finite grad):", status, "n_trunk", net.n_trunk, "n_dec", net.n_dec)
                for li, L in enumerate(net.layers):
                    if not bool(torch.isfinite(L.dw).all()) or (L.db is not None and not bool(torch.isfinite(L.db).all())):
                        print("   first bad layer", li, "cin", L.cin, "cout", L.cout, "k", L.kh, "dil", L.dilation,
                              "dw bad" if not bool(torch.isfinite(L.dw).all()) else "db bad",
                              int((~torch.isfinite(L.dw)).sum()), "of", L.dw.numel())
                        break
                print("rep", rep, "loss", l, "restored equal", same, "finite params/m/v/shadow/grads", fin)
        continue
    for i in range(int(os.environ.get("NSTEPS", "9"))):
        if "everysync" in mode: torch.cuda.synchronize()
        if "emptycache" in mode: torch.cuda.empty_cache()
        l = float(tr.step(data[i % 8]))
        runs += 1
        if not np.isfinite(l) or not bool(torch.isfinite(net.params).all()):
            bad += 1
            # which part of the flat buffer?
            nz = (~torch.isfinite(net.params)).nonzero().flatten()
            print("non-finite at rep %d step %d loss %r; %d params, first idx %s" % (rep, i, l, nz.numel(), nz[:3].tolist()))
            for li, L in enumerate(net.layers):
                if not bool(torch.isfinite(L.w).all()) or (L.b is not None and not bool(torch.isfinite(L.b).all())):
                    print("   layer", li, L.cin, L.cout, L.kh, "w bad" if not bool(torch.isfinite(L.w).all()) else "b bad")
                    break
            break
print(mode, "steps", runs, "bad runs", bad)
