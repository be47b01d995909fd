#!/usr/bin/env python3
"""Golden vectors of the data-side augmentation from the REAL reference's process.generate_2mix_snr /
generate_3mix_snr / generate_mix_noise (process.py:57-103).  Usage: python tools/make_goldens_data.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402
import process as RP  # noqa: E402


def main():
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    d = {}
    T = 4000
    s = [MG.keyed_randn(f"data.s{i}", (T,), sc) for i, sc in enumerate((0.05, 0.12, 0.3, 0.9))]
    d["s"] = np.stack([MG.npy(t) for t in s])
    cases = [(0, 1, -3.0), (0, 1, 4.0), (1, 0, 0.5), (2, 3, 2.0), (3, 2, -5.0), (0, 3, 5.0)]
    d["cases"] = np.array(cases, dtype=np.float32)
    d["mix2"] = np.stack([MG.npy(RP.generate_2mix_snr(s[int(i)].clone(), s[int(j)].clone(), snr)) for i, j, snr in cases])
    d["mix2_noclip"] = np.stack([MG.npy(RP.generate_2mix_snr(s[int(i)].clone(), s[int(j)].clone(), snr, clip=False)) for i, j, snr in cases])
    d["noise"] = np.stack([MG.npy(RP.generate_mix_noise(s[int(i)].clone(), s[int(j)].clone(), abs(snr) + 6.0)) for i, j, snr in cases])
    d["mix3"] = MG.npy(RP.generate_3mix_snr(s[0].clone(), s[1].clone(), s[2].clone(), 1.5, -2.0))
    d["zero"] = MG.npy(RP.generate_2mix_snr(torch.zeros(T), s[1].clone(), 3.0))
    np.savez_compressed(os.path.join(out, "data_aug.npz"), **d)
    print("data_aug:", len(d), "arrays")


if __name__ == "__main__":
    main()
