"""``Trainer`` / ``do_training`` -- the driver behind `yogo train` (yogo/train.py:43-665) around the HIP training step.

Same life cycle as the reference's ``Trainer``: dataset definition -> model (fresh or ``--from-pretrained``) -> dataloaders ->
optimiser tools -> run directory; epochs of training steps; validation every 4th epoch with ``best.pth`` / ``latest.pth``
checkpoints (same dict keys, yogo/train.py:280-293); final test of the best checkpoint through ``Metrics``; one process per GPU
for data-parallel runs (``mp.spawn``, yogo/train.py:654-656).  What runs inside is different: a step is ``HipTrainer.step``
(hand-written HIP forward / loss / backward / AdamW on one flat parameter buffer, RCCL all-reduce overlapped with backward
instead of torch DDP), batches arrive already on the device from ``yogo_amd.yogo_dataloader``.

The compute and the loaders sit behind two small seams (``backend_factory`` / ``loader_factory``).  The product binds them to
the HIP kernels -- there is no CPU compute path, ``do_training`` refuses to start without a GPU like the reference does
(yogo/train.py:645-650); the CPU plumbing test of BASELINE configs[0] binds them to the oracle FROM THE TEST.
wandb is optional: without it every ``wandb.log`` record goes to ``<run dir>/log.jsonl``.
"""
from __future__ import annotations

import json
import os
import sys
import warnings
from copy import deepcopy
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Tuple, Union

import torch

from yogo_amd.dataset_definition_file import DatasetDefinition
from yogo_amd.metrics import Metrics
from yogo_amd.model import YOGO
from yogo_amd.model_defns import get_model_func
from yogo_amd.utils.default_hyperparams import DefaultHyperparams as df
from yogo_amd.utils.utils import get_free_port
from yogo_amd.yogo_loss import YOGOLoss

WandbConfig = dict


class RunLog:
    """wandb when it is installed and wanted, a JSON-lines file in the run directory otherwise"""

    def __init__(self, run_dir: Path, config: dict, use_wandb: bool) -> None:
        self.path = Path(run_dir) / "log.jsonl"
        self.wandb = None
        if use_wandb:
            try:
                import wandb   # type: ignore

                wandb.init(config=config, entity=config.get("wandb_entity"), project=config.get("wandb_project"), name=config.get("name"),
                           notes=config.get("note"), tags=config.get("tags"))
                self.wandb = wandb
            except Exception as e:   # not installed / no network: fall back to the file
                warnings.warn(f"wandb unavailable ({e}); logging to {self.path}")

    def log(self, record: dict, step: Optional[int] = None) -> None:
        if self.wandb is not None:
            self.wandb.log(record, step=step)
            return
        clean = {k: (float(v) if isinstance(v, (int, float)) or (torch.is_tensor(v) and v.numel() == 1) else str(type(v).__name__))
                 for k, v in record.items()}
        with open(self.path, "a") as f:
            f.write(json.dumps({"step": step, **clean}) + "\n")

    def finish(self) -> None:
        if self.wandb is not None:
            self.wandb.finish()


class HipBackend:
    """the product's compute: HipTrainer for the step, the HIP forward + loss kernel for evaluation"""

    def __init__(self, net: YOGO, config: dict, total_steps: int, device) -> None:
        from yogo_amd.train import HipTrainer

        self.net, self.device = net, device
        self.loss = YOGOLoss(no_obj_weight=config["no_obj_weight"], iou_weight=config["iou_weight"],
                             label_smoothing=config["label_smoothing"]).to(device)
        self.opt = HipTrainer(net, self.loss, learning_rate=config["learning_rate"], weight_decay=config["weight_decay"],
                              total_steps=total_steps, decay_factor=config["decay_factor"], half=bool(config["half"]))
        self.opt.broadcast_parameters()
        self.half = bool(config["half"])

    def train_step(self, imgs: torch.Tensor, labels: torch.Tensor) -> Dict[str, float]:
        self.opt.step(imgs, labels)
        return self.opt.loss_components()

    def current_lr(self) -> float:
        return self.opt.current_lr()

    @torch.no_grad()
    def eval_batch(self, imgs: torch.Tensor, labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.half):
            outputs = self.net(imgs)
        loss, _ = self.loss(outputs, labels)
        return outputs, loss.detach()

    def optimizer_state_dict(self) -> dict:
        return self.opt.state_dict()

    def sync_eval_state(self) -> None:
        """rank 0's BatchNorm running statistics to every rank before evaluation (DDP's broadcast_buffers, yogo/train.py:155-159)"""
        self.opt.broadcast_buffers()

    def set_global_step(self, step: int) -> None:
        self.opt.global_step = int(step)


class Trainer:
    def __init__(self, config: WandbConfig, _rank: int = 0, _world_size: int = 1,
                 backend_factory: Optional[Callable] = None, loader_factory: Optional[Callable] = None) -> None:
        self.config = config
        self.device = f"cuda:{_rank}"
        self._rank, self._world_size = _rank, _world_size
        self.Sx: Optional[int] = None
        self.Sy: Optional[int] = None
        self.model_save_dir: Optional[Path] = None
        self.dataset_definition: Optional[DatasetDefinition] = None
        self.epoch = 0
        self.global_step = 0
        self.min_val_loss = float("inf")
        self._initialized = False
        self._backend_factory = backend_factory or (lambda net, cfg, steps, dev: HipBackend(net, cfg, steps, dev))
        self._loader_factory = loader_factory
        self.log: Optional[RunLog] = None

    @classmethod
    def train_from_ddp(cls, _rank: int, _world_size: int, config: WandbConfig) -> "Trainer":
        trainer = cls(config, _rank=_rank, _world_size=_world_size)
        trainer.init()
        trainer.train()
        return trainer

    # ---- set-up -----------------------------------------------------------------------------------------------------------
    def init(self) -> None:
        self._init_process_group()
        self._init_dataset_definition()
        self._init_model()
        self._init_dataset()
        self._init_training_tools()
        self._init_run_dir()
        self._initialized = True

    def _init_process_group(self) -> None:
        if self._world_size > 1 and not torch.distributed.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(self.config["master_port"])
            torch.cuda.set_device(self._torch_device())
            torch.distributed.init_process_group(backend=self.config.get("dist_backend", "nccl"), rank=self._rank,
                                                 world_size=self._world_size)

    def _init_dataset_definition(self) -> None:
        self.dataset_definition = DatasetDefinition.from_yaml(Path(self.config["dataset_descriptor_file"]))
        self.config["class_names"] = self.dataset_definition.classes

    def _init_model(self) -> None:
        if self.dataset_definition is None:
            raise RuntimeError("dataset definition not initialized")
        if self.config["pretrained_path"] is None or self.config["pretrained_path"] == "none":
            net = YOGO(img_size=tuple(self.config["image_hw"]), anchor_w=self.config["anchor_w"], anchor_h=self.config["anchor_h"],
                       is_rgb=self.config["rgb"], num_classes=len(self.config["class_names"]),
                       model_func=get_model_func(self.config["model"]))
            self.global_step = 0
        else:
            net, net_cfg = YOGO.from_pth(self.config["pretrained_path"])
            if any(int(a) != int(b) for a, b in zip(net.img_size.cpu().tolist(), self.config["image_hw"])):
                raise RuntimeError("mismatch in pretrained network image resize shape and current resize shape: "
                                   f"pretrained network image_hw = {net.img_size}, requested image_hw = {self.config['image_hw']}")
            self.global_step = net_cfg["step"]
            self.config["normalize_images"] = bool(net.normalize_images)
            self.config["model"] = net.model_version
            net.train()
        self.net = net.to(self._torch_device())
        self.Sx, self.Sy = net.get_grid_size()

    def _torch_device(self):
        return torch.device(self.config.get("compute_device") or self.device)

    def _init_dataset(self) -> None:
        if self.Sx is None or self.Sy is None:
            raise RuntimeError("model not initialized")
        if self._loader_factory is not None:
            loaders = self._loader_factory(self.dataset_definition, self.config, self.Sx, self.Sy)
        else:
            from yogo_amd.yogo_dataloader import get_dataloader

            loaders = get_dataloader(self.dataset_definition, self.config["batch_size"], Sx=self.Sx, Sy=self.Sy,
                                     image_hw=tuple(self.config["image_hw"]), rgb=self.config["rgb"],
                                     normalize_images=self.config["normalize_images"],
                                     split_fraction_override=self.config["dataset_split_override"], device=self._torch_device())
        self.train_dataloader = loaders["train"]
        self.validate_dataloader = loaders.get("val", [])
        self.test_dataloader = loaders.get("test", [])
        if self._dataset_size(self.validate_dataloader) == 0:
            warnings.warn("no validation dataset found")
        if self._dataset_size(self.test_dataloader) == 0:
            warnings.warn("no test dataset found")

    @staticmethod
    def _dataset_size(dataloader) -> int:
        return len(dataloader.dataset) if hasattr(dataloader, "dataset") else len(dataloader)

    def _init_training_tools(self) -> None:
        # T_max = epochs * batches per epoch, eta_min = lr / decay_factor, scheduler stepped every iteration (yogo/train.py:213-223)
        total_steps = self.config["epochs"] * len(self.train_dataloader)
        self.backend = self._backend_factory(self.net, self.config, total_steps, self._torch_device())
        if self.global_step:
            self.backend.set_global_step(0)   # the reference restarts its schedule on --from-pretrained; only the counter carries on

    def _init_run_dir(self) -> None:
        if self._rank != 0:
            return
        base = Path(self.config.get("trained_models_dir") or (Path.cwd() / "trained_models"))
        name = self.config.get("name") or f"run_{torch.randint(100000000, size=(1,)).item():08}"
        self.model_save_dir = base / name
        self.model_save_dir.mkdir(exist_ok=True, parents=True)
        self.log = RunLog(self.model_save_dir, self.config, use_wandb=bool(self.config.get("wandb_project")))
        self.log.log({"Sx": self.Sx, "Sy": self.Sy, "training set size": self._dataset_size(self.train_dataloader),
                      "validation set size": self._dataset_size(self.validate_dataloader),
                      "testing set size": self._dataset_size(self.test_dataloader)}, step=self.global_step)

    # ---- checkpoint: the reference's dict (yogo/train.py:267-293) ---------------------------------------------------------------
    def checkpoint(self, filename: Union[str, Path], model_name: str, **kwargs) -> None:
        torch.save(
            {
                "epoch": self.epoch,
                "step": self.global_step,
                "normalize_images": self.config["normalize_images"],
                "classes": self.config["class_names"],
                "model_name": model_name,
                "model_state_dict": deepcopy({k: v.detach().cpu() for k, v in self.net.state_dict().items()}),
                "optimizer_state_dict": deepcopy(self.backend.optimizer_state_dict()),
                "model_version": self.net.model_version,
                **kwargs,
            },
            str(filename),
        )

    # ---- the loop (yogo/train.py:295-372) -----------------------------------------------------------------------------------------
    def train(self) -> None:
        if not self._initialized:
            raise RuntimeError("trainer not initialized")
        if self._world_size > 1:
            torch.distributed.barrier()
        for epoch in range(self.config["epochs"]):
            self.epoch = epoch
            sampler = getattr(self.train_dataloader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            self.net.train()
            for imgs, labels in self.train_dataloader:
                lr = self.backend.current_lr()
                comps = self.backend.train_step(imgs, labels)
                self.global_step += 1
                if self._rank == 0 and self.log is not None:
                    self.log.log({"train loss": comps["loss"], "epoch": epoch, "LR": lr,
                                  **{k: v for k, v in comps.items() if k != "loss"}}, step=self.global_step)
            if epoch % 4 == 0:
                self._validate()
        if self._rank == 0 and self.model_save_dir is not None:
            best = self.model_save_dir / "best.pth"
            if best.exists():
                ckpt = torch.load(best, map_location="cpu", weights_only=False)
                self.net.load_state_dict(ckpt["model_state_dict"])
            else:
                warnings.warn(f"no best model found at {best} for testing...")
        if self._world_size > 1 and hasattr(self.backend, "opt") and hasattr(self.backend.opt, "broadcast_parameters"):
            self.backend.opt.broadcast_parameters()   # every rank tests rank 0's best checkpoint, not its own last weights
        if hasattr(self.backend, "sync_eval_state"):
            self.backend.sync_eval_state()
        test_metrics = self.test(self.test_dataloader, self._torch_device(), self.config, self.net, rank=self._rank, backend=self.backend)
        if self._rank == 0:
            if test_metrics is not None:
                self._log_test_metrics(*test_metrics)
            else:
                warnings.warn("no test metrics found - most likely test_dataloader is empty")
            if self.log is not None:
                self.log.finish()
        if self._world_size > 1:
            torch.distributed.destroy_process_group()

    @torch.no_grad()
    def _validate(self) -> None:
        if self._dataset_size(self.validate_dataloader) == 0:
            return
        net_state = self.net.training
        self.net.eval()
        if hasattr(self.backend, "sync_eval_state"):
            self.backend.sync_eval_state()   # every rank validates with rank 0's running statistics, as under DDP
        # a zero tensor on the device, as the reference starts (yogo/train.py:385): a rank whose shard of the validation set is
        # empty still takes part in the all-reduce below
        val_loss = torch.zeros(1, dtype=torch.float32, device=self._torch_device())
        for imgs, labels in self.validate_dataloader:
            _, loss = self.backend.eval_batch(imgs, labels)
            val_loss = val_loss + loss.reshape(-1)[:1].to(val_loss.dtype)
        if self._world_size > 1:
            torch.distributed.all_reduce(val_loss, op=torch.distributed.ReduceOp.SUM)
            val_loss = val_loss / self._world_size
        self.net.train(net_state)
        if self._rank != 0:
            return
        mean_val_loss = float(val_loss) / len(self.validate_dataloader)
        self.log.log({"val loss": mean_val_loss}, step=self.global_step)
        name = self.config.get("name") or "recent_run"
        if mean_val_loss < self.min_val_loss:
            self.min_val_loss = mean_val_loss
            self.log.log({"best_val_loss": mean_val_loss}, step=self.global_step)
            self.checkpoint(self.model_save_dir / "best.pth", model_name=f"{name}_best" if not self.config.get("name") else name)
        else:
            self.checkpoint(self.model_save_dir / "latest.pth", model_name=f"{name}_latest" if not self.config.get("name") else name)

    @staticmethod
    def _check_keys(config: dict) -> None:
        for k in ("class_names", "no_obj_weight", "iou_weight", "label_smoothing", "half"):
            if k not in config:
                raise ValueError(f"config is missing {k}")

    @staticmethod
    @torch.no_grad()
    def test(test_dataloader, device, config: WandbConfig, net: torch.nn.Module, rank: int = 0, include_mAP: bool = True,
             include_background: bool = False, backend=None) -> Optional[Tuple[Any, ...]]:
        """loss + Metrics over the test split (yogo/train.py:446-528); the network keeps ``inference=False`` like the reference"""
        if Trainer._dataset_size(test_dataloader) == 0:
            return None
        net_state = net.training
        net.eval()
        Trainer._check_keys(config)
        metrics = Metrics(classes=config["class_names"], device=str(device), sync_on_compute=False, include_mAP=include_mAP,
                          include_background=include_background)
        if backend is None:
            loss_fn = YOGOLoss(no_obj_weight=config["no_obj_weight"], iou_weight=config["iou_weight"],
                               label_smoothing=config["label_smoothing"]).to(device)

            def eval_batch(imgs, labels):
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(config["half"])):
                    out = net(imgs)
                loss, _ = loss_fn(out, labels)
                return out, loss.detach()
        else:
            eval_batch = backend.eval_batch
        test_loss = 0.0
        for imgs, labels in test_dataloader:
            outputs, loss = eval_batch(imgs, labels)
            test_loss += float(loss)
            metrics.update(outputs.detach(), labels.detach())
        mean_loss = test_loss / len(test_dataloader)
        (mAP, confusion, accuracy, roc, precision, recall, calibration_error, missed, extra, total) = metrics.compute()
        net.train(net_state)
        if rank != 0:
            return None
        return (mean_loss, mAP, metrics.get_wandb_confusion_matrix(confusion), accuracy, roc, precision, recall, calibration_error,
                missed, extra, total, config["class_names"])

    def _log_test_metrics(self, mean_loss, mAP, confusion_rows, accuracy, roc, precision, recall, calibration_error, missed, extra, total,
                          class_names) -> None:
        rec = {"test loss": mean_loss, "test mAP": float(mAP["map"]), "test calibration error": float(calibration_error),
               "test total true objects": int(total)}
        for i, c in enumerate(class_names):
            rec[f"test accuracy {c}"] = float(accuracy[i])
            rec[f"test precision {c}"] = float(precision[i])
            rec[f"test recall {c}"] = float(recall[i])
            rec[f"test missed {c}"] = int(missed[i])
            rec[f"test extra {c}"] = int(extra[i])
        if self.log is not None:
            self.log.log(rec, step=self.global_step)
        if self.model_save_dir is not None:
            with open(self.model_save_dir / "test_metrics.json", "w") as f:
                json.dump({**rec, "confusion": confusion_rows}, f, indent=1)


def build_config(args) -> dict:
    """the flat config dict of yogo/train.py:612-643 (it doubles as the wandb config)"""
    return {
        "learning_rate": args.learning_rate,
        "decay_factor": args.lr_decay_factor,
        "weight_decay": args.weight_decay,
        "label_smoothing": args.label_smoothing,
        "iou_weight": args.iou_weight,
        "no_obj_weight": args.no_obj_weight,
        "classify_weight": args.classify_weight,   # parsed and stored but never handed to the loss -- as in the reference
        "tcp_store_port": str(get_free_port()),
        "master_port": str(get_free_port()),
        "epochs": args.epochs,
        "batch_size": args.batch_size,
        "device": str(args.device),
        "anchor_w": df.ANCHOR_W,
        "anchor_h": df.ANCHOR_H,
        "model": args.model,
        "half": args.half,
        "rgb": args.rgb_images,
        "image_hw": tuple(args.image_hw),
        "pretrained_path": args.from_pretrained,
        "normalize_images": args.normalize_images,
        "dataset_split_override": args.dataset_split_override,
        "dataset_descriptor_file": args.dataset_descriptor_file,
        "slurm-job-id": os.getenv("SLURM_JOB_ID", default=None),
        "torch-version": torch.__version__,
        "python-version": sys.version,
        "name": args.name,
        "note": args.note,
        "tags": args.tags,
        "wandb_entity": args.wandb_entity,
        "wandb_project": args.wandb_project,
    }


def visible_gpu_count() -> int:
    """GPUs this process may use, counted WITHOUT creating a HIP context in the caller: the parent of a multi-GPU job must stay
    off the GPU (its children are fresh interpreters, one per device).  ``torch.cuda.device_count()`` asks the driver for the
    count only (no context, no allocation); honouring HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES is the driver's job."""
    return int(torch.cuda.device_count())


def do_training(args) -> None:
    """parse args, then one training process per visible GPU (yogo/train.py:606-656).

    Two launches are supported:
      * ``yogo train ...`` on its own: the parent counts the devices (no HIP context) and ``mp.spawn``s one fresh interpreter per
        GPU, as the reference does (yogo/train.py:654-656);
      * under a launcher -- ``python -m torch.distributed.run --nproc-per-node N -m yogo_amd train ...`` --
        RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment and this process IS one rank."""
    config = build_config(args)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        if os.environ.get("MASTER_PORT"):
            config["master_port"] = os.environ["MASTER_PORT"]
        config["compute_device"] = f"cuda:{local}"
        Trainer.train_from_ddp(rank, world, config)
        return
    world_size = visible_gpu_count()
    if world_size == 0:
        raise RuntimeError("at least 1 gpu is required for training; the hot path is HIP-only (no CPU compute path)")
    if world_size == 1:
        Trainer.train_from_ddp(0, 1, config)
    else:
        import torch.multiprocessing as mp

        mp.spawn(Trainer.train_from_ddp, args=(world_size, config), nprocs=world_size, join=True)
