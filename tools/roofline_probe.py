"""Launches the roofline cases (fqss_amd/roofline_cases.py) a fixed number of times each, for the rocprofv3 passes:

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rl_stats -- python3 tools/roofline_probe.py
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/rl_fetch -- python3 tools/roofline_probe.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/rl_write -- python3 tools/roofline_probe.py
  python3 tools/roofline_probe.py --reduce <fetch counter csv> <write counter csv> gpurun_out/rl_manifest.json <out.json>

--reduce folds the two counter collections into HBM bytes per launch, per case and per kernel, reported RAW in both readings of the
gfx950 FETCH_SIZE counter (MI355X_MICROARCH.md "HBM": it tallies a 128-B read request at 64 B, so wide streaming reads need x2; 64-B
requests do not): traffic_x2 = 2 FETCH_SIZE + WRITE_SIZE (the guide's rule, what bench.py reports as `traffic`) and traffic_x1 =
FETCH_SIZE + WRITE_SIZE (its lower bound).  No per-kernel fitting."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ITERS = 6


def run(which="cfg2"):
    import torch
    from fqss_amd import roofline_cases as RC
    dev = torch.device("cuda", 0)
    cases = RC.build(dev) if which == "cfg2" else RC.build_other(dev)      # --set other: the roofline kernels of cfg 3 / 4 / 5
    torch.cuda.synchronize()
    manifest = []
    for c in cases:
        for i in range(ITERS):
            c["fn"](i)
        torch.cuda.synchronize()
        manifest.append(dict(kernel=c["kernel"], label=c["label"], iters=ITERS, rd=c["rd"], wr=c["wr"], launches=c["launches"], group=c["group"]))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/rl_manifest.json", "w") as f:
        json.dump(manifest, f, indent=1)


def _members(kernel):
    """kernel SYMBOLS (the name up to its template arguments) of a case's launches, in launch order.  The manifest names carry the
    template arguments that matter for reading the table ("k_qgemm<1>"); the profiler prints the full instantiation
    ("fqss::k_qgemm<1, 3>(...)"), so the match is on the symbol, bounded on both sides ("k_qwgrad" must not match "k_qwgrad2")"""
    if kernel.startswith("k_gnq_bwd_rows+apply"):
        return ["k_gnq_bwd_rows", "k_gnq_bwd_apply"]
    if kernel == "k_qgemm<3>":
        return ["k_qgemm"]
    if kernel == "k_dwq_bwd<3, GA, GB>":
        return ["k_dwq_bwd"]
    return [kernel.split("<")[0]]


def _is(symbol, name):
    i = name.find("fqss::" + symbol)
    return i >= 0 and name[i + 6 + len(symbol):i + 7 + len(symbol)] in ("<", "(", "")


def _sequence(path, counter):
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter and "fqss::" in row["Kernel_Name"]:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    return rows


def _assign(rows, manifest):
    """consume the dispatch sequence case by case (the probe launches the cases in manifest order): KiB per invocation.  A case whose
    kernels are not found where the manifest says they were launched is an ERROR (round 2 lost 11 of 15 kernels silently here)"""
    out, pos = [], 0
    for m in manifest:
        mem = _members(m["kernel"])
        vals = []
        for it in range(m["iters"]):
            tot = 0.0
            for sym in mem:
                q = pos
                while q < len(rows) and not _is(sym, rows[q][1]):
                    q += 1
                if q == len(rows):
                    raise SystemExit(f"roofline_probe --reduce: no dispatch of fqss::{sym} for case {m['label']!r} (launch {it} of {m['iters']}); "
                                     f"next dispatches: {[r[1][:60] for r in rows[pos:pos + 4]]}")
                tot += rows[q][2]
                pos = q + 1
            vals.append(tot)
        out.append(sum(vals) / len(vals))
    return out


def reduce(fetch_csv, write_csv, manifest_json, out_json):
    manifest = json.load(open(manifest_json))
    fetch = _assign(_sequence(fetch_csv, "FETCH_SIZE"), manifest)
    write = _assign(_sequence(write_csv, "WRITE_SIZE"), manifest)
    cases, per = [], {}
    for m, f, w in zip(manifest, fetch, write):
        if f is None or w is None:
            continue
        fb, wb, alg = f * 1024.0, w * 1024.0, m["rd"] + m["wr"]
        cases.append(dict(kernel=m["kernel"], label=m["label"], launches_per_step=m["launches"], group_of_launches=m["group"],
                          algorithmic_read_bytes=m["rd"], algorithmic_write_bytes=m["wr"], FETCH_SIZE_KiB=round(f, 1), WRITE_SIZE_KiB=round(w, 1),
                          traffic_x2=round(2 * fb + wb), traffic_x1=round(fb + wb), x2_over_algorithmic=round((2 * fb + wb) / alg, 3),
                          x1_over_algorithmic=round((fb + wb) / alg, 3)))
        g = per.setdefault(m["kernel"], [0.0, 0.0, 0])
        g[0] += m["launches"] * (2 * fb + wb)
        g[1] += m["launches"] * (fb + wb)
        g[2] += m["launches"]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/roofline_probe.py on MI355X; raw counters, "
                     "x2 = 2 FETCH_SIZE + WRITE_SIZE (gfx950: 128-B read requests tallied at 64 B), x1 = FETCH_SIZE + WRITE_SIZE",
           "per_launch_bytes": {k: {"x2": round(v[0] / v[2]), "x1": round(v[1] / v[2])} for k, v in per.items()}, "cases": cases}
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    for c in cases:
        print("%-100s alg %6.1f MB  x2 %6.1f MB (%.2f)  x1 %6.1f MB (%.2f)" % (c["label"][:100], (c["algorithmic_read_bytes"] + c["algorithmic_write_bytes"]) / 1e6,
                                                                              c["traffic_x2"] / 1e6, c["x2_over_algorithmic"], c["traffic_x1"] / 1e6, c["x1_over_algorithmic"]))


if __name__ == "__main__":
    if len(sys.argv) >= 6 and sys.argv[1] == "--reduce":
        reduce(*sys.argv[2:6])
    elif len(sys.argv) >= 3 and sys.argv[1] == "--set":
        run(sys.argv[2])
    else:
        run()
