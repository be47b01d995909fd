"""create_pretrained_model -> (student, teacher) (reference: train_env/train_utils.py:8-27)."""
import copy

import torch

from ..quantization.qat.models.load_model import create_model, quantize_model


def create_pretrained_model(model_cfg, use_weights=True):
    model = create_model(model_cfg)
    path = model_cfg.get("model_path")
    if use_weights and path is not None:
        sd = (torch.hub.load_state_dict_from_url(path, map_location="cpu", check_hash=True)
              if path.startswith("https") else torch.load(path, map_location="cpu", weights_only=False))
        try:
            model.load_state_dict(sd.get("state", sd), strict=True)
        except Exception:
            try:
                model.load_pretrain(path)
            except Exception:
                print("Warning: No pretraind weights were loaded!")
    fmodel = copy.deepcopy(model)          # the float TEACHER is frozen before the student is quantized
    model = quantize_model(model, model_cfg["quantization"])
    return model, fmodel


def augmentation_2mix(signal1, signal2, augmentation_cfg):
    """train_utils.py:30-39: one uniform SNR draw (host RNG, like the reference) -> generate_2mix_snr on the device"""
    import numpy as np
    from ..process import generate_2mix_snr
    if augmentation_cfg.get("distribution") != "uniform":
        raise AssertionError("Augmentation is not supoorted!")
    snr = np.random.uniform(low=augmentation_cfg.get("param0"), high=augmentation_cfg.get("param1"))
    return generate_2mix_snr(signal1, signal2, snr)


def augmentation_3mix(signal1, signal2, signal3, augmentation_cfg):
    """train_utils.py:42-52"""
    import numpy as np
    from ..process import generate_3mix_snr
    if augmentation_cfg.get("distribution") != "uniform":
        raise AssertionError("Augmentation is not supoorted!")
    snr1_23 = np.random.uniform(low=augmentation_cfg.get("param0"), high=augmentation_cfg.get("param1"))
    snr2_3 = np.random.uniform(low=augmentation_cfg.get("param0"), high=augmentation_cfg.get("param1"))
    return generate_3mix_snr(signal1, signal2, signal3, snr1_23, snr2_3)
