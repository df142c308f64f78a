"""World-size-2 data-parallel exchange on CPU (gloo) THROUGH HipTrainer's own methods (broadcast_parameters, the backward hook
that starts the head-side part of the all-reduce, exchange_begin / exchange_end): per-rank clamp BEFORE the exchange, SUM
all-reduce in two parts, 1/world scale -- reproduces the mean of the ranks' clamped gradients (DDP semantics), and the flat
parameter buffer really backs the module's parameters.  The per-rank gradients come from the CPU oracle (test
infrastructure); the kernels themselves are covered by the GPU tests (tests/test_gpu_dp.py runs HipTrainer.step on two ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "oracle")]
    import yogo_oracle as O
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer, cosine_lr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    Himg, Wimg, C, Bper = 48, 64, 3, 2
    model = YOGO((Himg, Wimg), 0.0425, 0.0555, C, model_func=__import__("yogo_amd.model_defns", fromlist=["x"]).get_model_func("quarter_filters"))
    if rank == 1:   # different weights on rank 1 until the broadcast
        with torch.no_grad():
            for p_ in model.parameters():
                p_.add_(1.0)
    tr = HipTrainer(model, total_steps=10)          # no kernels run here: construction + the exchange methods are host logic
    flat = tr.flat
    assert tr.world == world and 0 < tr.split_off < flat.total
    tr.broadcast_parameters()                        # rank-0 weights and buffers everywhere (what DDP does at construction)
    w0 = flat.flat.clone()
    dist.broadcast(w0, src=0)
    assert torch.equal(w0, flat.flat)
    assert all(p.data_ptr() >= flat.flat.data_ptr() and p.data_ptr() < flat.flat.data_ptr() + 4 * flat.total for p in model.parameters())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    spec = O.arch("quarter_filters", C)
    names = [k for k, _ in model.named_parameters()]

    def grads_for(x, lab):
        leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
        sdl = dict(sd)
        sdl.update(leaf)
        out = O.yogo_forward(x, sdl, spec, 0.0425, 0.0555, train=True)
        loss, _ = O.yogo_loss(out, lab)
        loss.backward()
        return O.clamp_grads({k: v.grad for k, v in leaf.items()}, 1.0)

    xs = O.synthetic_images(world * Bper, Himg, Wimg, seed=5)
    labs = O.synthetic_labels(world * Bper, model.Sx, model.Sy, K=4, num_classes=C, seed=6)
    mine = grads_for(xs[rank * Bper:(rank + 1) * Bper], labs[rank * Bper:(rank + 1) * Bper])
    for n, p in zip(names, model.parameters()):
        flat.grad_views[id(p)].copy_(mine[n])
    # the exchange exactly as HipTrainer.step drives it: the backward hook fires when the middle layer's gradients are done
    # (head-side part starts travelling), the rest follows after backward, exchange_end joins and returns the mean's scale
    for i in range(len(tr.engine.layers) - 1, -1, -1):
        tr._on_layer_done(i)
        if i == tr.split_layer:
            assert len(tr._pending) == 1
    tr.exchange_begin(0, tr.split_off)
    scale = tr.exchange_end()
    assert scale == 1.0 / world and not tr._pending
    got = flat.grad * scale
    # reference: every rank's clamped gradient averaged (DDP semantics: hooks clamp per rank, then mean)
    all_g = [grads_for(xs[r * Bper:(r + 1) * Bper], labs[r * Bper:(r + 1) * Bper]) for r in range(world)]
    want = torch.cat([sum(g[n] for g in all_g).flatten() / world for n in names])
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-7)
    assert abs(cosine_lr(0, 3e-4, 10, 3e-5) - 3e-4) < 1e-12 and abs(cosine_lr(10, 3e-4, 10, 3e-5) - 3e-5) < 1e-12
    # BatchNorm buffers: ranks drift apart in training (per-rank batch statistics, no SyncBN), HipTrainer.broadcast_buffers hands
    # everyone rank 0's before evaluation (DDP's broadcast_buffers, yogo/train.py:155-159): one flat broadcast
    with torch.no_grad():
        for name_, b_ in model.named_buffers():
            if "running_mean" in name_:
                b_.fill_(float(rank) + 0.25)
            elif "running_var" in name_:
                b_.fill_(2.0 + rank)
            elif "num_batches" in name_:
                b_.fill_(7 + rank)
    gen0 = tr.engine.generation
    tr.broadcast_buffers()
    for name_, b_ in model.named_buffers():
        if "running_mean" in name_:
            assert bool((b_ == 0.25).all()), name_
        elif "running_var" in name_:
            assert bool((b_ == 2.0).all()), name_
        elif "num_batches" in name_:
            assert int(b_) == 7 and b_.dtype == torch.long, name_
    assert tr.engine.generation == gen0 + 1      # folded inference weights derive from these buffers: caches must be dropped
    flat.publish_grads()
    assert all(p.grad is not None and p.grad.data_ptr() == flat.grad_views[id(p)].data_ptr() for p in model.parameters())
    if rank == 0:
        open(os.path.join(tmp, "ok"), "w").write("ok")
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
