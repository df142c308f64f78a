"""`yogo` console entry point: `python -m yogo_amd train|test|infer ...` (yogo/__main__.py:8-39)."""
import sys

import torch

from yogo_amd.utils.argparsers import global_parser


def main(argv=None) -> None:
    # (the `yogo` console script of pyproject.toml lands here: worker processes must be fresh interpreters, never forks of a
    #  parent that may have touched the GPU)
    if torch.multiprocessing.get_start_method(allow_none=True) != "spawn":
        torch.multiprocessing.set_start_method("spawn", force=True)
    p = global_parser()
    args = p.parse_args(argv)
    if args.task == "train":
        from yogo_amd.trainer import do_training

        do_training(args)
    elif args.task == "test":
        from yogo_amd.utils.test_model import do_model_test

        do_model_test(args)
    elif args.task == "export":
        print("yogo_amd: `export` (ONNX / OpenVINO for another deployment target) is not part of this MI355X build; "
              "checkpoints written here load in the reference (same state_dict keys), export them there")
        sys.exit(1)
    elif args.task == "infer":
        from yogo_amd.infer import do_infer

        do_infer(args)
    else:
        p.print_help()


if __name__ == "__main__":
    torch.multiprocessing.set_start_method("spawn", force=True)
    main()
