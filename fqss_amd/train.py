#!/usr/bin/env python3
"""`train.py -env <asteroid|speechbrain|tasnet|htdemucs> -y cfg.yaml` (reference: train.py:10-53).
This build serves the asteroid env (ConvTasNet, DPTNet) and the speechbrain env (Sepformer); tasnet / htdemucs are later rows
of SURVEY.md §8."""
import argparse

import torch


def argument_handler():
    p = argparse.ArgumentParser()
    p.add_argument("--env_name", "-env", type=str, required=True, help="Training environment name: asteroid/tasnet/speechbrain/htdemucs")
    p.add_argument("--yml_path", "-y", type=str, required=True, help="YML configuration file")
    p.add_argument("--use_cpu", action="store_true", help="Use cpu")
    p.add_argument("--local_rank", type=int, default=0, help="Rank ID")
    p.add_argument("--distributed_launch", action="store_true", help="Multi-GPU training")
    return p.parse_args()


def train():
    args = argument_handler()
    device = "cpu" if args.use_cpu or not torch.cuda.is_available() else "cuda"
    if args.env_name == "asteroid":
        from .train_env.asteroid_librimix import asteroid_librimix_trainer
        asteroid_librimix_trainer.train(args.yml_path, device)
    elif args.env_name == "speechbrain":
        from .train_env.speechbrain_librimix import speechbrain_librimix_trainer
        speechbrain_librimix_trainer.train(args.yml_path, args.local_rank, args.distributed_launch, device)
    elif args.env_name in ("tasnet", "htdemucs"):
        raise NotImplementedError(f"env {args.env_name}: SURVEY.md §8 row a15 (later rounds)")
    else:
        assert False, "Training environment {} is not supported!".format(args.env_name)
    print("Training is done!")


if __name__ == "__main__":
    train()
