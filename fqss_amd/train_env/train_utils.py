"""create_pretrained_model -> (student, teacher) (reference: train_env/train_utils.py:8-27)."""
import copy

import torch

from ..quantization.qat.models.load_model import create_model, quantize_model


def create_pretrained_model(model_cfg, use_weights=True):
    model = create_model(model_cfg)
    path = model_cfg.get("model_path")
    if use_weights and path is not None:
        sd = (torch.hub.load_state_dict_from_url(path, map_location="cpu", check_hash=True)
              if path.startswith("https") else torch.load(path, map_location="cpu"))
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
