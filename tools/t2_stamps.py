"""Cycle stamps of k_tgemm2's barriers (diagnostic library built with -DFQSS_T2_STAMP, selected through FQSS_LIB): per barrier and wave of
workgroup 0 the time the wave was done with its work, the time its own s_waitcnt returned and the time the barrier released it.
    FQSS_LIB=$PWD/fqss_amd/csrc/variants/libfqss_stamp.so python tools/t2_stamps.py [t1|t3]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fqss_amd import _lib
from fqss_amd import roofline_cases as RC


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "t3"
    # t3 / t1: the teacher GEMMs (libfqss built with -DFQSS_T2_STAMP); dx / dx2: the student's dgrad 128->512 / res|skip pair on the
    # ring (csrc/qgemm_ring.hip built with -DFQSS_R_STAMP)
    key, nth, sym = {"t3": ("k_tgemm2<1>", 0, "fqss_debug_t2_stamps"), "t1": ("k_tgemm2<0>", 0, "fqss_debug_t2_stamps"),
                     "dx": ("k_qgemm<1>", 0, "fqss_debug_r_stamps"), "dx2": ("k_qgemm<1>", 1, "fqss_debug_r_stamps")}[which]
    case = [c for c in RC.build(torch.device("cuda", 0)) if c["kernel"] == key][nth]
    for i in range(6):
        case["fn"](i)
    torch.cuda.synchronize()
    buf = np.zeros((8, 160, 3), dtype=np.uint64)
    rc = getattr(_lib.load(), sym)(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    t = buf.astype(np.int64)
    nb = int((t[0, :, 2] > 0).sum())
    t0 = t[:, 0, 0].min()
    role = ["C0", "C1", "C2", "C3", "A4", "A5", "W6", "W7"]      # compute x 4, activation / gradient x 2, weight x 2
    print(f"{which}: {nb} barriers; columns per wave: work-done / wait-done relative to the barrier's release (cycles, negative = earlier)")
    print("bar  release(+cyc)  period | " + "  ".join(f"{r:>13s}" for r in role))
    prev = None
    for k in range(nb):
        rel = t[:, k, 2].max()
        row = "  ".join(f"{int(t[w, k, 0] - rel):6d}/{int(t[w, k, 1] - rel):6d}" for w in range(8))
        print(f"{k:3d}  {int(rel - t0):10d}  {int(rel - prev) if prev is not None else 0:6d} | {row}")
        prev = rel
    print("last-arriving wave per barrier (by wait-done):", [role[int(np.argmax(t[:, k, 1]))] for k in range(nb)])


if __name__ == "__main__":
    main()
