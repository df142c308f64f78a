"""worker of tests/test_gpu_dp.py: one data-parallel rank running HipTrainer.step on its shard (both ranks share cuda:0, gloo
carries the exchange).  usage: dp_worker.py RANK WORLD PORT OUTDIR HALF COMM"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def main():
    rank, world, port, outdir, half, comm = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == "1", sys.argv[6]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import yogo_oracle as O
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Himg, Wimg, C, Bper = 96, 128, 5, 2
    torch.manual_seed(7 + rank)                       # ranks start from DIFFERENT weights: broadcast_parameters must fix that
    model = YOGO((Himg, Wimg), 0.0425, 0.0555, C).cuda()
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=10, half=half, comm=comm)
    tr.broadcast_parameters()
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    xs = O.synthetic_images(world * Bper, Himg, Wimg, seed=5)
    labs = O.synthetic_labels(world * Bper, model.Sx, model.Sy, K=5, num_classes=C, seed=6)
    x, lab = xs[rank * Bper:(rank + 1) * Bper].cuda(), labs[rank * Bper:(rank + 1) * Bper].cuda()
    from yogo_amd import _hip

    _hip.launch_log(True)
    tr.step(x, lab)
    torch.cuda.synchronize()
    log = _hip.read_launch_log()
    _hip.launch_log(False)
    sd1 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr.broadcast_buffers()     # what the Trainer does in front of validation: rank 0's BatchNorm statistics everywhere
    torch.cuda.synchronize()
    torch.save({"sd0": sd0, "grad_sum": tr.flat.grad.cpu(), "flat": tr.flat.flat.cpu(), "loss": tr.loss_components(),
                "sd1": sd1, "sd2": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "split_off": tr.split_off, "launch_log": log},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
