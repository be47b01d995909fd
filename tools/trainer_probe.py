"""Why did the trainer's captured step replay at 15.7 ms when bench.py's replays at 13.3?  Same process: (1) the asteroid trainer on a
generated LibriMix tree, (2) its stepper on synthetic device-resident batches, (3) on two batches of the tree, (4) a fresh bench-like step."""
import os
import sys
import tempfile
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqss_amd.data import synth_batch  # noqa: E402
from fqss_amd.runtime import KDTrainStep  # noqa: E402
from fqss_amd.smoke import build_pair  # noqa: E402
from fqss_amd.train_env.asteroid_librimix import asteroid_librimix_trainer as T  # noqa: E402
from tests.helpers_librimix import make_librimix_tree  # noqa: E402


def timed(step, X, n=40):
    for it in range(6):
        step(*X[it & 1], x_next=X[(it + 1) & 1][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(n):
        step(*X[it & 1], x_next=X[(it + 1) & 1][0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


amp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
with tempfile.TemporaryDirectory() as tmp:
    tree = make_librimix_tree(tmp, n_train=160, n_dev=8, seconds=(4.1, 4.6))
    conf = yaml.safe_load(open(os.path.join(ROOT, "configs", "convtasnet_2spks_8k.yaml")))
    conf["work_dir"] = os.path.join(tmp, "run")
    conf["dataset_cfg"].update(train_dir=tree["train_dir"], valid_dir=tree["valid_dir"], segment=4)
    conf["training_cfg"].update(epochs=4, batch_size=8)
    yml = os.path.join(tmp, "cfg.yaml")
    open(yml, "w").write(yaml.safe_dump(conf))
    hist = T.train(yml, "cuda")
    print([round(h["train_ms_per_step"], 2) for h in hist])
    step = T.LAST_SYSTEM.stepper
    X = [synth_batch(8, 32000, seed=s, device="cuda") for s in (1, 2)]
    print(f"trainer's stepper, synthetic batches: {timed(step, X):.2f} ms")
    Xs = [(amp * 5 * x, amp * 5 * t) for x, t in X]
    print(f"trainer's stepper, synthetic batches x5: {timed(step, Xs):.2f} ms")
    ds = T.prepare_datasets(conf["dataset_cfg"], conf["training_cfg"], torch.device("cuda", 0))[0]
    Xt = [ds.batch(list(range(8 * i, 8 * i + 8))) for i in range(2)]
    print(f"trainer's stepper, two batches of the tree: {timed(step, Xt):.2f} ms; |x| mean {Xt[0][0].abs().mean().item():.3f} vs synthetic {X[0][0].abs().mean().item():.3f}")
    model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
    fresh = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, teacher_ahead=True)
    fresh(*X[0])
    with torch.no_grad():
        for _ in range(49):
            model(X[0][0])
    fresh(*X[0])
    fresh.capture(*X[0])
    print(f"fresh bench-like step, synthetic: {timed(fresh, X):.2f} ms; on the tree's batches: {timed(fresh, Xt):.2f} ms")
    print(f"trainer's stepper again, synthetic: {timed(step, X):.2f} ms")
