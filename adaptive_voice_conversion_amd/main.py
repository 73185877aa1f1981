"""Training CLI with the reference's flags (main.py:7-33; train.sh).

    python -m adaptive_voice_conversion_amd.main -c config.yaml -d <data_dir> -train_set train \\
        -train_index_file train_samples_128.json -store_model_path <path> -iters 200000

Multi-GPU: ``torchrun --nproc-per-node N -m adaptive_voice_conversion_amd.main ...`` (one process per
GPU; gradients are all-reduced over RCCL inside ``Solver.ae_step``).  ``--device_feed`` keeps the
corpus resident in HBM instead of the 4-worker DataLoader.
"""
import os
from argparse import ArgumentParser

import torch

from .config import DEFAULT_YAML, load_config
from .solver import Solver


def main():
    parser = ArgumentParser()
    parser.add_argument("-config", "-c", default=DEFAULT_YAML)
    parser.add_argument("-data_dir", "-d", default=None)
    parser.add_argument("-train_set", default="train")
    parser.add_argument("-train_index_file", default="train_samples_64.json")
    parser.add_argument("-logdir", default="log/")
    parser.add_argument("--load_model", action="store_true")
    parser.add_argument("--load_opt", action="store_true")
    parser.add_argument("-store_model_path", default="model")
    parser.add_argument("-load_model_path", default="model")
    parser.add_argument("-summary_steps", default=100, type=int)
    parser.add_argument("-save_steps", default=5000, type=int)
    parser.add_argument("-tag", "-t", default="init")
    parser.add_argument("-iters", default=0, type=int)
    parser.add_argument("--device_feed", action="store_true", help="gather segments from an HBM-resident corpus")
    args = parser.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    config = load_config(args.config)
    solver = Solver(config=config, args=args)
    if args.iters > 0:
        solver.train(n_iterations=args.iters)


if __name__ == "__main__":
    main()
