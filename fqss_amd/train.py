#!/usr/bin/env python3
"""`train.py -env <asteroid|speechbrain|tasnet|htdemucs> -y cfg.yaml` (reference: train.py:10-53).
This build serves the asteroid env (ConvTasNet, DPTNet), the speechbrain env (Sepformer) and the htdemucs env (HTDemucs); the tasnet
env is outside SURVEY.md §8."""
import argparse
import os
import sys

from .launch import already_launched, spawn_ranks, visible_gpus


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
    n_gpus = 0 if args.use_cpu else visible_gpus()          # device_count(): no HIP context in this process yet
    device = "cuda" if n_gpus > 0 else "cpu"
    # The reference trains on every GPU of the node from this one command (pl.Trainer(strategy="ddp", devices="auto"),
    # asteroid_librimix_trainer.py:125-135; speechbrain's --distributed_launch, speechbrain_librimix_trainer.py:592; the tasnet env's own
    # Popen per GPU, tasnet_musdbhq_trainer.py:17-30): more than one visible GPU (or --distributed_launch) and no launcher around us ->
    # start one rank per GPU BEFORE this process touches a GPU and wait for them (launch.py).  FQSS_NPROC overrides the rank count
    # (1 = stay single-process).
    if device == "cuda" and not already_launched():
        want = int(os.environ.get("FQSS_NPROC", n_gpus if (n_gpus > 1 or args.distributed_launch) else 1))
        if want > 1:
            rc = spawn_ranks(want, [sys.executable, "-m", "fqss_amd.train"] + sys.argv[1:])
            if rc:
                sys.exit(rc)
            print("Training is done!")
            return
    if args.env_name == "asteroid":
        from .train_env.asteroid_librimix import asteroid_librimix_trainer
        asteroid_librimix_trainer.train(args.yml_path, device)
    elif args.env_name == "speechbrain":
        from .train_env.speechbrain_librimix import speechbrain_librimix_trainer
        speechbrain_librimix_trainer.train(args.yml_path, args.local_rank, args.distributed_launch, device)
    elif args.env_name == "htdemucs":
        from .train_env.htdemucs_musdbhq import train as htdemucs_musdbhq_trainer
        # the reference hands the device to its hydra entry point as an override (train.py:44-46); -y selects the YAML here
        sys.argv[1:] = ["+device=" + device, "+yml_path=" + args.yml_path]
        htdemucs_musdbhq_trainer.main()
    elif args.env_name == "tasnet":
        raise NotImplementedError("env tasnet (ConvTasNetMusic on MUSDB) is outside SURVEY.md §8")
    else:
        assert False, "Training environment {} is not supported!".format(args.env_name)
    print("Training is done!")


if __name__ == "__main__":
    train()
