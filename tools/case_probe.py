"""A/B of library variants on the isolated launches of fqss_amd/roofline_cases.py (cfg-2 shapes, hipGraph of 24 launches, HIP events).
    python tools/case_probe.py <kernel-substring>[,<substring>...] [lib.so ...]     (no lib = the product library)
Every library runs in a child process of its own (FQSS_LIB is read at import), the set is visited twice in alternating order; with
`--digest` a child also prints a checksum of every output tensor of one launch per case (bit-identity across variants)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(sel):
    import torch
    sys.path.insert(0, ROOT)
    from fqss_amd import roofline_cases as RC
    dev = torch.device("cuda", 0)
    cases = [c for c in RC.build(dev) if any(s in c["kernel"] for s in sel)]
    for c in cases:
        out = c["fn"](0)
        torch.cuda.synchronize()
        outs = out if isinstance(out, (tuple, list)) else (out,)
        h = hashlib.sha1()
        for t in outs:
            if torch.is_tensor(t):
                h.update(t.detach().contiguous().cpu().numpy().tobytes())
        ms = min(RC.time_case(c) for _ in range(3))
        print(f"{c['kernel']:32s} {ms * 1e3:8.2f} us   outputs {h.hexdigest()[:12]}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2].split(","))
        sys.exit(0)
    sel, libs = sys.argv[1], sys.argv[2:] or [""]
    for rnd in range(2):
        for lib in (libs if rnd == 0 else libs[::-1]):
            env = dict(os.environ)
            if lib:
                env["FQSS_LIB"] = os.path.abspath(lib)
            print(f"== {lib or 'product'} (round {rnd})", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sel], env=env, check=False)
