"""GPU: the `train.py -env asteroid` plugin surface end to end on synthetic mixtures."""
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def test_asteroid_env_trains_and_exports(tmp_path):
    assert torch.cuda.is_available()
    from fqss_amd.train_env.asteroid_librimix import asteroid_librimix_trainer as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = yaml.safe_load(open(os.path.join(root, "configs", "convtasnet_2spks_8k_synthetic.yaml")))
    conf["work_dir"] = str(tmp_path / "run")
    conf["dataset_cfg"].update(segment=0.5, steps_per_epoch=4, val_steps=2)
    conf["training_cfg"].update(epochs=2, batch_size=2)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(conf))
    hist = T.train(str(yml), "cuda")
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["loss"])) for h in hist)
    assert hist[-1]["loss"] < hist[0]["loss"] + 5.0
    sd = torch.load(os.path.join(conf["work_dir"], "best_model.pth"))
    assert len(sd) == 948 and "masker.TCN.0.shared_block.0.activation_fake_quantize.min_range" in sd
    assert os.path.exists(os.path.join(conf["work_dir"], "conf.yml"))
    with pytest.raises(RuntimeError):
        T.train(str(yml), "cpu")
