#!/usr/bin/env python3
"""Benchmark of the YOGO hot path on MI355X: full training step (forward + loss + backward + AdamW) on synthetic
772x1032 grayscale batches, base_model, 7 classes -- BASELINE.json's metric "training images/sec".

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  A "step" is one optimisation step over one per-GPU batch resident in HBM (weak scaling:
per-GPU batch fixed, the images are sharded over ranks, one RCCL all-reduce of the flat gradient per step).
`roofline` is measured live with HIP events (torch.cuda.Event on the stream the kernels are launched on) around every
launch of the dominant kernel -- conv_bf16_ws_kernel (the stride-1 bf16 convolutions with 128 output channels: persistent, wavefront-specialised) by
default, conv_igemm_f32_kernel<4,2> under --dtype f32;
`cpu_baseline` times the CPU oracle (oracle/yogo_oracle.py: the reference's algorithm on torch CPU ops) on a bounded
sample of the same workload -- a reported baseline, never the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, NUM_CLASSES = 772, 1032, 7
ANCHOR_W, ANCHOR_H = 0.0425, 0.0555
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same table: dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0           # same table, "HBM3E peak BW" (6.29 TB/s measured copy)
TRAIN_GFLOP_PER_IMG = 66.48


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(batch: int = 8):
    """the CPU oracle (oracle/yogo_oracle.py = the reference's algorithm on torch CPU ops) on the host cores
    of this box, bounded samples of the same workload: (i) eval forward + decode, (ii) full train step, (iii) the per-image
    format_preds loop.  2 warm-up + 5 timed iterations, median; the thread count is the one that maximises the oracle's own
    throughput (a sweep over a forward pass: median of 3 per thread count), reported next to the core count and to the all-cores figure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import yogo_oracle as O

    cores = len(os.sched_getaffinity(0))
    spec = O.arch("base_model", NUM_CLASSES)
    sd = O.init_state(spec, seed=0)
    for k in list(sd):   # trained-like running statistics keep the eval-mode activations finite (hard parts)
        if k.endswith("running_var"):
            sd[k] = torch.full_like(sd[k], 5000.0)
    x = O.synthetic_images(batch, H, W, seed=0)
    Sx, Sy = O.grid_size(spec, H, W)
    lab = O.synthetic_labels(batch, Sx, Sy, K=64, num_classes=NUM_CLASSES, seed=1)

    def fwd():
        with torch.no_grad():
            return O.yogo_forward(x, sd, spec, ANCHOR_W, ANCHOR_H, inference=True)

    best_t, best_thr, sweep = None, cores, {}
    for thr in sorted({t for t in (4, 8, 16, 32, 64, cores) if t <= cores}):
        torch.set_num_threads(thr)
        fwd()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fwd()
            ts.append(time.perf_counter() - t0)
        dt = _median(ts)
        sweep[thr] = round(batch / dt, 2)
        if best_t is None or dt < best_t:
            best_t, best_thr = dt, thr
    torch.set_num_threads(best_thr)

    def timed(fn, warm=2, reps=5):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return _median(ts)

    t_fwd = timed(fwd)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    step_no = [0]

    def one():
        step_no[0] += 1
        leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
        sdl = dict(sd)
        sdl.update(leaf)
        ns = {}
        out = O.yogo_forward(x, sdl, spec, ANCHOR_W, ANCHOR_H, train=True, new_stats=ns)
        loss, _ = O.yogo_loss(out, lab)
        loss.backward()
        g = O.clamp_grads({k: v.grad for k, v in leaf.items()})
        lr = O.cosine_lr(step_no[0] - 1, 3e-4, 1000, 3e-5)
        for k in names:
            p, m, v = O.adamw_step(sd[k], g[k], state[k][0], state[k][1], step_no[0], lr)
            sd[k], state[k] = p.detach(), (m, v)
        sd.update(ns)

    t_step = timed(one)
    # (iii) the reference's per-image post-processing loop (yogo/infer.py:45,73) on realistic predictions (100 objects per image)
    preds = O.synthetic_predictions(64, Sx, Sy, NUM_CLASSES, K=100, seed=2)
    t_fmt = timed(lambda: [O.format_preds(p) for p in preds], warm=1, reps=3)
    return {"value": round(batch / t_step, 3), "unit": "images/s", "cores": best_thr, "cores_available": cores, "kind": "port",
            "sample": f"oracle train step (fwd+loss+bwd+clamp+AdamW), fp32, batch {batch}: median of 5 after 2 warm-up steps, "
                      f"{best_thr} torch threads (fastest of a sweep; {cores} cores visible)",
            "eval_forward_decode_images_per_s": round(batch / t_fwd, 3),
            "thread_sweep_eval_forward_images_per_s": {str(k): v for k, v in sweep.items()},   # median of 3 per thread count; the last key = all visible cores
            "format_preds_loop_realistic_images_per_s": round(64 / t_fmt, 1)}


def _timed_gpu(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def inference_extras(model, dev, B: int = 256):
    """BASELINE configs[4] (secondary to the training metric): `yogo infer` at batch 256 -- eval forward + box decode + batched
    threshold / NMS (format_preds_batched) end to end on synthetic images (the network of this run, after its training steps),
    (`YOGO.forward_raw` -> the fused decode + threshold + NMS kernel), and the NMS kernel alone on 'dense' predictions (93 % of the cells fire, what a random-init network gives: the worst
    case) and on 'realistic' ones (100 objects per image).  Roofline of decode / NMS: HBM, algorithmic bytes 0.60 MB per image
    (SURVEY.md 8d)."""
    from yogo_amd.synthetic import synthetic_dense_predictions, synthetic_images, synthetic_predictions
    from yogo_amd.utils import format_preds_batched

    model.eval()
    model.inference = True
    x = synthetic_images(B, H, W, device=dev, seed=7)
    out = {"batch": B}
    with torch.no_grad():
        ms32 = _timed_gpu(lambda: model(x[:64]))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ms16 = _timed_gpu(lambda: model(x))

            def e2e():
                return format_preds_batched(model.forward_raw(x))   # decode inside the threshold + NMS kernel's loads
            # the end-to-end leg runs on a network that FIRES at a stated rate: the head's objectness bias is shifted so that 0.8 % of
            # the grid cells (~100 per image, the 'realistic' post-process workload) pass the 0.5 threshold on these images; the
            # forward's arithmetic does not depend on the value of a bias, the post-process now has its realistic amount of work
            head = [mod for mod in model.modules() if isinstance(mod, torch.nn.Conv2d)][-1]
            raw4 = model.forward_raw(x).raw[:, 4].float().flatten()
            shift = float(torch.quantile(raw4[torch.randperm(raw4.numel(), device=raw4.device)[:1 << 20]], 1.0 - 0.008))
            with torch.no_grad():
                head.bias[4] -= shift
            ms_e2e = _timed_gpu(e2e)
            kept = float(e2e()[2].float().mean())
            fire = float((model(x)[:, 4] > 0.5).float().mean())   # share of the grid cells above the objectness threshold on THIS network
            with torch.no_grad():
                head.bias[4] += shift
        dense = synthetic_dense_predictions(B, model.Sx, model.Sy, NUM_CLASSES, device=dev)
        ms_nd = _timed_gpu(lambda: format_preds_batched(dense), reps=2)
        real = synthetic_predictions(B, model.Sx, model.Sy, NUM_CLASSES, K=100, device=dev)
        ms_nr = _timed_gpu(lambda: format_preds_batched(real))
        from yogo_amd.synthetic import raw_from_predictions
        from yogo_amd.utils.prediction_formatting import RawPredictions

        rp = RawPredictions(raw_from_predictions(real, model._Cxs, model._Cys, *model._decode_scalars()[:2]), model._Cxs, model._Cys, *model._decode_scalars(), True)
        ms_fr = _timed_gpu(lambda: format_preds_batched(rp), reps=20)
        raw = torch.randn(B, 5 + NUM_CLASSES, model.Sy, model.Sx, device=dev)
        dec = torch.empty_like(raw)
        from yogo_amd import _hip

        aw, ah, wm, hm = model._decode_scalars()
        ms_dec = _timed_gpu(lambda: _hip.call("yogo_decode_fwd", raw, dec, model._Cxs, model._Cys, B, 5 + NUM_CLASSES, model.Sy, model.Sx,
                                              aw, ah, wm, hm, 1, _hip.stream_ptr()), reps=20)
    img_bytes = (5 + NUM_CLASSES) * model.Sy * model.Sx * 4
    out["forward_decode_fp32_images_per_s"] = round(64 / ms32 * 1e3, 1)
    out["forward_decode_bf16_images_per_s"] = round(B / ms16 * 1e3, 1)
    out["end_to_end_bf16_forward_decode_nms_images_per_s"] = round(B / ms_e2e * 1e3, 1)
    out["end_to_end_fire_rate"] = round(fire, 4)
    out["end_to_end_kept_per_image"] = round(kept, 1)
    out["end_to_end_network"] = (f"the network of this run after its training steps, objectness bias of the head calibrated on these images: {fire:.4f} of the "
                                 f"grid cells pass the 0.5 threshold, {kept:.1f} boxes per image survive threshold + NMS (the 'realistic' post-process "
                                 "workload; the 'dense' one, where 93 % of the cells fire, is timed apart below)")
    out["threshold_nms_dense_images_per_s"] = round(B / ms_nd * 1e3, 1)
    out["threshold_nms_realistic_images_per_s"] = round(B / ms_nr * 1e3, 1)
    out["fused_decode_threshold_nms_realistic_images_per_s"] = round(B / ms_fr * 1e3, 1)
    out["roofline_decode"] = {"bound": "hbm", "achieved": round(2 * B * img_bytes / ms_dec / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(2 * B * img_bytes / ms_dec / 1e6 / HBM_PEAK_GBS, 4), "avg_call_ms": round(ms_dec, 4)}
    out["roofline_nms_realistic"] = {"bound": "hbm", "achieved": round(B * img_bytes / ms_nr / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(B * img_bytes / ms_nr / 1e6 / HBM_PEAK_GBS, 4), "avg_call_ms": round(ms_nr, 4)}
    out["roofline_fused_decode_nms_realistic"] = {"bound": "hbm", "achieved": round(B * img_bytes / ms_fr / 1e6, 1), "peak": HBM_PEAK_GBS,
                                                  "unit": "GB/s", "frac": round(B * img_bytes / ms_fr / 1e6 / HBM_PEAK_GBS, 4),
                                                  "avg_call_ms": round(ms_fr, 4),
                                                  "replaces_ms": round(ms_dec + ms_nr, 4)}
    out["note"] = ("eval-mode base_model at batch 256 (fp32 forward at 64); dense = 93 % of the 12 513 cells fire with overlapping boxes "
                   "(NMS worst case, bound by the greedy suppression chain), realistic = 100 objects per image")
    model.train()
    model.inference = False
    return out


def fp32_forward_loss(dev, B: int = 64):
    """BASELINE configs[1]: fp32 forward + loss at batch 64 (train-mode forward with BatchNorm batch statistics, decode, fused loss
    kernel), with the fp32-MFMA fraction of the 128-channel convolutions (conv_igemm_f32_kernel<4, 2, false>, peak 157.3 TFLOP/s)"""
    from yogo_amd import _hip
    from yogo_amd.engine import get_engine
    from yogo_amd.model import YOGO
    from yogo_amd.synthetic import synthetic_images, synthetic_labels
    from yogo_amd.yogo_loss import YOGOLoss

    torch.manual_seed(0)
    m = YOGO((H, W), ANCHOR_W, ANCHOR_H, NUM_CLASSES).to(dev)
    m.train()
    x = synthetic_images(B, H, W, device=dev, seed=300)
    lab = synthetic_labels(B, m.Sx, m.Sy, K=64, num_classes=NUM_CLASSES, device=dev, seed=301)
    L = YOGOLoss().to(dev)
    eng = get_engine(m.model)

    def run():
        with torch.no_grad():
            L(m(x), lab)

    ms = _timed_gpu(run, reps=3)
    eng.prof = []
    run()
    torch.cuda.synchronize()
    sel = [e for e in eng.prof if e[0] == "fwd" and e[2] == 4]
    t = sum(e[4].elapsed_time(e[5]) for e in sel)
    fl = sum(e[3] for e in sel)
    eng.prof = None
    tf = fl / max(t, 1e-9) / 1e9
    return {"images_per_s": round(B / ms * 1e3, 1), "ms": round(ms, 3), "batch": B, "dtype": "f32",
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_f32_kernel<4, 2, false> (forward of the 128-channel 3x3 layers)",
                         "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}}


def spawn_ranks(n: int, argv) -> int:
    """run `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a CHILD process and relay its output"""
    import socket
    import subprocess

    with socket.socket() as sk:   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:   # rank 0 prints the one JSON line; everything else the ranks print goes to stderr
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE configs[2]/[3]: 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the secondary inference measurements (profiling runs)")
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"],
                    help="bf16 (default, BASELINE configs[2]): bf16 activations/gradients + bf16 MFMA, fp32 master weights and "
                         "statistics; f32: fp32 storage + exact fp32 MFMA")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) or gloo (rehearsal of the N>1 path on one GPU)")
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl"],
                    help="gradient exchange transport: torch.distributed (backend above) or librccl called directly through the C ABI")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce after backward instead of the overlapped two-part exchange")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, like the reference's `yogo train` does
        # (yogo/train.py:645-656 counts the devices and mp.spawn()s one process per GPU).  This parent never touches the GPU
        # (no torch.cuda call at all), never exec()s, relays rank 0's JSON line and exits with the children's status.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    dev_index = local_rank % torch.cuda.device_count()   # == local_rank on a full node; ranks share the card in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; reporting n_gpus={world}", file=sys.stderr)

    from yogo_amd.model import YOGO
    from yogo_amd.synthetic import synthetic_images, synthetic_labels
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    torch.manual_seed(0)   # reference init under manual_seed(0); identical on every rank
    model = YOGO((H, W), ANCHOR_W, ANCHOR_H, NUM_CLASSES).to(dev)
    model.train()
    B = args.batch
    trainer = HipTrainer(model, YOGOLoss().to(dev), total_steps=args.steps + args.warmup + 47, half=(args.dtype == "bf16"),
                         comm=args.comm, overlap=not args.no_overlap)
    trainer.broadcast_parameters()
    imgs = synthetic_images(B, H, W, device=dev, seed=100 + rank)
    labels = synthetic_labels(B, model.Sx, model.Sy, K=64, num_classes=NUM_CLASSES, device=dev, seed=200 + rank)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # device spin-up (untimed, before the W warm-up steps): a cold MI355X ramps its clocks and the caching allocator grows its
    # pools over the first steps -- on this pool the first dozen steps of a fresh process run up to 2x slower.  Step until
    # three consecutive steps agree within 5 % (at most 40 steps); every rank runs the same count so collectives stay matched.
    spin = []
    for i in range(40):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        trainer.step(imgs, labels)
        torch.cuda.synchronize()
        spin.append(time.perf_counter() - ts)
        done = len(spin) >= 6 and max(spin[-3:]) < 1.05 * min(spin[-3:])
        if world > 1:
            flag = torch.tensor([1.0 if done else 0.0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            done = bool(flag.item() > 0.5)
        if done:
            break
    for _ in range(args.warmup):
        trainer.step(imgs, labels)
    torch.cuda.synchronize()
    barrier()
    # the timed region carries HIP events around the DOMINANT kernel's launches only (the roofline's `achieved`); the per-kind
    # breakdown of every convolution launch comes from a few untimed steps after it
    dominant_tag = 34 if args.dtype == "bf16" else 4
    trainer.engine.prof = []
    # (tag 35: the stride-2 128 -> 128 forward, the one launch of north_star's K5-K8 forward set that is not the dominant kernel)
    trainer.engine.prof_only = {dominant_tag, 35} if args.dtype == "bf16" else {dominant_tag}
    # per step: host time to ENQUEUE the step (perf_counter around trainer.step, no sync) and one HIP event pair on the launch
    # stream -- whether the timed region was GPU-bound or waiting for the host shows in the record itself
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    enq = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        ts = time.perf_counter()
        trainer.step(imgs, labels)
        enq.append(time.perf_counter() - ts)
        ev[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    gpu_step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    prof = trainer.engine.prof
    loss_rec = trainer.loss_components()
    nb = max(1, min(args.steps, 5))
    trainer.engine.prof = []
    trainer.engine.prof_only = None
    for _ in range(nb):
        trainer.step(imgs, labels)
    torch.cuda.synchronize()
    barrier()
    prof_all = trainer.engine.prof
    trainer.engine.prof = None

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # how many ranks the GRADIENT transport actually joins: a ones-vector summed through the trainer's own exchange (the same
    # exchange_begin / exchange_end the step uses: torch.distributed "nccl" = RCCL, or librccl through the C ABI under --comm rccl)
    transport = "none (one rank)"
    rccl_ranks = 1
    if world > 1:
        keep = trainer.flat.grad[:64].clone()
        trainer.flat.grad[:64].fill_(1.0)
        trainer.exchange_begin(0, 64)
        trainer.exchange_end()
        torch.cuda.synchronize()
        joined = int(round(float(trainer.flat.grad[:64].min().item())))
        trainer.flat.grad[:64].copy_(keep)
        is_rccl = args.comm == "rccl" or args.backend == "nccl"
        transport = ("librccl via yogo_comm_*" if args.comm == "rccl" else f"torch.distributed {args.backend}") + f": ones-vector all-reduce summed to {joined}"
        rccl_ranks = joined if is_rccl else 0

    if rank == 0:
        # ---- roofline of the dominant kernel, from the in-loop HIP events ---------------------------------------
        def ms_of(es):
            return sum(e[4].elapsed_time(e[5]) for e in es)

        by_kind = {}
        for kind in ("fwd", "dgrad", "wgrad"):
            es = [e for e in prof_all if e[0] == kind]
            t_ms = ms_of(es)
            by_kind[kind] = {"ms_per_step": round(t_ms / nb, 3),
                             "tflops": round(sum(e[3] for e in es) / max(t_ms, 1e-9) / 1e9, 2)}
        if os.environ.get("YOGO_BENCH_VERBOSE"):
            agg = {}
            for e in prof_all:
                k = (e[0], e[1])
                a = agg.setdefault(k, [0.0, 0.0, 0.0])
                a[0] += e[4].elapsed_time(e[5])
                a[1] += e[3]
                a[2] += e[6]
            for (kind, layer), (t_ms, fl2, by2) in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
                print(f"[bench] layer {layer} {kind:6s} {t_ms / nb:8.3f} ms/step  {fl2 / t_ms / 1e9:7.2f} TFLOP/s  "
                      f"{by2 / t_ms / 1e6:8.1f} GB/s", file=sys.stderr)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        tj = {}
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
            except Exception:
                tj = {}
        if args.dtype == "bf16":
            # dominant kernel: conv_bf16_ws_kernel = every stride-1 convolution with 128 output channels (forward
            # of layers 3/5/6, data gradient of layers 5/6).  Algorithmic FLOPs per launch: 2*B*Cout*Cin*k*k*OH*OW.
            sel = [e for e in prof if e[0] in ("fwd", "dgrad") and e[2] == 34]
            ms = ms_of(sel)
            fl = sum(e[3] for e in sel)
            achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            # (the kernel has one instantiation per epilogue: <0> for layers 5 / 6, <7> for layer 3 -- weight their launches)
            tw = [(v.get("hbm_bytes_per_launch"), v.get("launches_fetch", 0)) for k, v in tj.items() if k.startswith(("conv_bf16_ws_kernel", "conv_bf16_ws16_kernel"))]
            tw = [(a, n) for a, n in tw if a is not None and n]
            traffic = round(sum(a * n for a, n in tw) / sum(n for _, n in tw), 1) if tw else None
            allc = [e for e in prof_all if e[0] in ("fwd", "dgrad") and e[2] in (30, 34)]
            roof = {"bound": "mfma", "kernel": "conv_bf16_ws16_kernel + conv_bf16_ws_kernel<3|7> (persistent wavefront-specialised stride-1 bf16 convolutions with 128 "
                                               "output channels: forward of layers 3/5/6, data gradient of layers 5/6)",
                    "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "calls_timed": len(sel), "avg_call_ms": round(ms / max(1, len(sel)), 4),
                    "traffic_source": "profiles/traffic.json: HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + "
                                      "WRITE_SIZE, tools/collect_profile.sh) -- a committed constant, not measured in this run",
                    "algorithmic_gbs": round(sum(e[6] for e in sel) / max(ms, 1e-9) / 1e6, 1),
                    "all_bf16_conv_gbs": round(sum(e[6] for e in allc) / max(ms_of(allc), 1e-9) / 1e6, 1)}
        else:
            sel = [e for e in prof if e[0] in ("fwd", "dgrad") and e[2] == 4]
            ms = ms_of(sel)
            fl = sum(e[3] for e in sel)
            achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic = tj.get("conv_igemm_f32_kernel<4,2,false>", {}).get("hbm_bytes_per_launch")
            roof = {"bound": "mfma", "kernel": "conv_igemm_f32_kernel<4,2> (fwd+dgrad of the 128-channel 3x3 layers)",
                    "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc passes), not measured in this run",
                    "calls_timed": len(sel), "avg_call_ms": round(ms / max(1, len(sel)), 4)}
        value = world * B * args.steps / dt
        rec_ns = None
        if args.dtype == "bf16":
            # what north_star names: ">= 70 % of gfx950 MFMA peak on the 3x3 conv GEMM" on the forward of the 64->128 / 128->128 3x3
            # layers (SURVEY.md K5-K8 = yogo/model_defns.py:49-65, layers 3..6 here), stride-2 layer 4 included -- Sigma FLOPs / Sigma
            # HIP-event time over those launches of the timed region
            ns = [e for e in prof if e[0] == "fwd" and 3 <= e[1] <= 6]
            ns_ms = ms_of(ns)
            ns_tf = sum(e[3] for e in ns) / max(ns_ms, 1e-9) / 1e9
            per = {}
            for e in ns:
                a = per.setdefault(e[1], [0.0, 0.0])
                a[0] += e[4].elapsed_time(e[5])
                a[1] += e[3]
            rec_ns = {"bound": "mfma", "kernels": "forward of layers 3-6 (K5-K8: conv 64->128 s1, 128->128 s2, 128->128 s1 x 2)",
                      "achieved": round(ns_tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ns_tf / BF16_MFMA_PEAK_TFLOPS, 4),
                      "calls_timed": len(ns), "ms_per_step": round(ns_ms / max(1, args.steps), 4),
                      "per_layer_tflops": {str(k): round(v[1] / max(v[0], 1e-9) / 1e9, 1) for k, v in sorted(per.items())}}
        rec = {
            "metric": "training images/sec (772x1032 gray)", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: full train step (fwd+loss+bwd+clamp+AdamW), base_model, 772x1032x1 uint8, "
                                   "7 classes" + ("" if args.dtype == "bf16" else " -- run at fp32 storage+arithmetic"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": roof,
            "roofline_north_star_fwd": rec_ns,
            "host_enqueue_ms_per_step": round(1e3 * _median(enq), 3),
            "host_enqueue_ms_per_step_max": round(1e3 * max(enq), 3),
            "gpu_ms_per_step_events": round(_median(gpu_step_ms), 3),
            "gpu_ms_per_step_events_min_max": [round(min(gpu_step_ms), 3), round(max(gpu_step_ms), 3)],
            "rccl_ranks": rccl_ranks,
            "gradient_transport": transport,
            "step_tflops": round(value * TRAIN_GFLOP_PER_IMG / 1e3, 2),
            "conv_breakdown": by_kind,
            "loss": round(loss_rec["loss"], 4),
        }
        if world == 1 and not args.no_inference:
            rec["inference"] = inference_extras(model, dev)
            rec["configs1_fp32_forward_loss"] = fp32_forward_loss(dev)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
