"""Launches the roofline cases (fqss_amd/roofline_cases.py, plus its single-pass calibration shapes and a 65.5 MB
device copy) a fixed number of times each, for the rocprofv3 passes:

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rl_stats -- python3 tools/roofline_probe.py
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/rl_fetch -- python3 tools/roofline_probe.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/rl_write -- python3 tools/roofline_probe.py
  python3 tools/roofline_probe.py --reduce <fetch counter csv> <write counter csv> gpurun_out/rl_manifest.json <out.json>

--reduce folds the two counter collections into HBM bytes per launch, per case and per kernel.  gfx950 byte
scale (MI355X_MICROARCH.md "HBM"): FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE tallies a 128-B read request at 64 B,
so wide streaming reads need x2 while 64-B requests do not.  The scale of each kernel's access pattern is therefore
CALIBRATED on a shape of that kernel where every byte is requested exactly once (known byte count), and that factor
is applied to its other shapes; kernels without a calibration shape use the factor that makes their cheapest
plausible reading (factor in {1, 2}) consistent -- both raw and scaled numbers are written out."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ITERS = 10
UNCALIBRATED_SCALE = {"k_qgemm<0>": 1.0}
KERNELS = ("k_tgemm", "k_qwgrad", "k_qgemm<1>", "k_qgemm<0>", "k_dwq_bwd", "k_ewq_bwd", "k_actq_bwd",
           "__amd_rocclr_copyBuffer")


def run():
    import torch
    from fqss_amd import roofline_cases as RC
    dev = torch.device("cuda", 0)
    cases = RC.build(dev, calib=True)
    a = torch.empty(8 * 512 * 4000, device=dev).normal_()
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    manifest = []
    for c in cases:
        it = max(1, ITERS * c["launches"] // 24)   # launch counts in the step's proportions
        for _ in range(it):
            c["fn"]()
        torch.cuda.synchronize()
        manifest.append(dict(kernel=c["kernel"], label=c["label"], iters=it, rd=c["rd"], wr=c["wr"], calib=c["calib"],
                             launches=c["launches"]))
    for _ in range(ITERS):
        b.copy_(a)
    torch.cuda.synchronize()
    manifest.append(dict(kernel="__amd_rocclr_copyBuffer", label="calibration: 65.5 MB device copy", iters=ITERS, rd=4.0 * a.numel(),
                         wr=4.0 * a.numel(), calib=True, launches=0, min_grid=1 << 16))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/rl_manifest.json", "w") as f:
        json.dump(manifest, f, indent=1)


def _sequence(path, counter):
    """dispatch-ordered [(kernel key, value, grid)] of the kernels of interest"""
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            for key in KERNELS:
                if key in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), key, float(row["Counter_Value"]), int(row["Grid_Size"])))
                    break
    rows.sort()
    return rows


def _assign(rows, manifest):
    """consume the dispatch sequence case by case (the probe launches the cases in manifest order)"""
    out, pos = [], 0
    for m in manifest:
        vals = []
        if m["kernel"] not in KERNELS:      # multi-kernel cases (timed in bench.py only) have no single PMC row
            out.append(None)
            continue
        while pos < len(rows) and len(vals) < m["iters"]:
            _, key, v, grid = rows[pos]
            pos += 1
            if key == m["kernel"] and grid >= m.get("min_grid", 0):
                vals.append(v)
        out.append(sum(vals) / max(1, len(vals)) if vals else None)
    return out


def reduce(fetch_csv, write_csv, manifest_json, out_json):
    manifest = json.load(open(manifest_json))
    fetch = _assign(_sequence(fetch_csv, "FETCH_SIZE"), manifest)
    write = _assign(_sequence(write_csv, "WRITE_SIZE"), manifest)
    KIB = 1024.0
    # per-kernel read scale from its single-pass calibration shape (true bytes / raw FETCH bytes)
    scale = {}
    for m, f in zip(manifest, fetch):
        if m["calib"] and f:
            scale[m["kernel"]] = m["rd"] / (f * KIB)
    cases = []
    for m, f, w in zip(manifest, fetch, write):
        if f is None or w is None:
            continue
        sc = scale.get(m["kernel"])
        # uncalibrated kernels: fp32 rows streamed 16 B/lane in >= 128-B runs are the guide's x2 case; the u8-code
        # rows of the forward q-GEMM are requested in 64-B runs (one request = 64 B, x1)
        used = sc if sc is not None else UNCALIBRATED_SCALE.get(m["kernel"], 2.0)
        cases.append(dict(kernel=m["kernel"], label=m["label"], launches_per_step=m["launches"], calibration_shape=m["calib"],
                          algorithmic_read_bytes=m["rd"], algorithmic_write_bytes=m["wr"], FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w,
                          read_scale=round(used, 3), read_scale_source="calibrated on this kernel's single-pass shape" if sc is not None
                          else ("x%g (request width of the access pattern, MI355X_MICROARCH.md HBM section)" % used),
                          read_bytes=round(f * KIB * used), write_bytes=round(w * KIB),
                          traffic_over_algorithmic=round((f * KIB * used + w * KIB) / (m["rd"] + m["wr"]), 3)))
    per_kernel = {}
    for c in cases:
        if c["calibration_shape"] or not c["launches_per_step"]:
            continue
        g = per_kernel.setdefault(c["kernel"], [0.0, 0])
        g[0] += c["launches_per_step"] * (c["read_bytes"] + c["write_bytes"])
        g[1] += c["launches_per_step"]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/roofline_probe.py on MI355X",
           "read_scale_per_kernel": {k: round(v, 3) for k, v in scale.items()},
           "per_launch_bytes": {k: round(v[0] / v[1]) for k, v in per_kernel.items()}, "cases": cases}
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("read_scale_per_kernel", "per_launch_bytes")}, indent=1))
    for c in cases:
        print("%-90s read %7.1f MB (alg %6.1f)  write %6.1f MB (alg %6.1f)  x%.2f" % (
            c["label"][:90], c["read_bytes"] / 1e6, c["algorithmic_read_bytes"] / 1e6, c["write_bytes"] / 1e6,
            c["algorithmic_write_bytes"] / 1e6, c["traffic_over_algorithmic"]))


GROUPS = {"k_gnq_bwd_rows+coef+apply": ("k_gnq_bwd_rows", "k_gnq_bwd_coef", "k_gnq_bwd_apply")}
# (label, algorithmic read / write bytes per element of the [8, 512, 3999] tensor, launches per step) in the probe's launch order
GROUP_CASES = {"k_gnq_bwd_rows+coef+apply": (
    ("gLN+fq backward (2 passes) with the producer conv's STE/PReLU/range/bias backward fused, C=512", 14.0, 4.0, 24),
    ("gLN+fq backward (2 passes, plain), C=512", 10.0, 4.0, 24))}


def reduce_groups(fetch_csv, write_csv, out_json, elems=8 * 512 * 3999, iters=ITERS):
    """multi-kernel roofline cases: the PMC rows of the member kernels are summed per invocation (the probe launches each case
    `iters` times back to back).  Read scale x2: the members stream fp32 rows 16 B per lane (128-B requests, tallied at 64 B on
    gfx950, MI355X_MICROARCH.md HBM section); the u8 code rows they also read are 1/9 .. 1/13 of the bytes, so x2 overstates the
    traffic by at most that share -- the conservative reading."""
    def rows(path, counter):
        out = []
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter:
                    val = float(r.get("Counter_Value_KiB", r.get("Counter_Value")))
                    out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], val))
        return sorted(out)
    doc = json.load(open(out_json))
    for group, members in GROUPS.items():
        per_case = []
        for path, counter in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
            seq = [(k, v) for _, k, v in rows(path, counter) if any(m in k for m in members)]
            inv, cur = [], 0.0
            for k, v in seq:
                cur += v
                if members[-1] in k:          # the last member closes an invocation
                    inv.append(cur)
                    cur = 0.0
            per_case.append([sum(inv[i * iters:(i + 1) * iters]) / iters for i in range(len(GROUP_CASES[group]))])
        tot = n = 0.0
        doc["cases"] = [c for c in doc["cases"] if c["kernel"] != group]
        for (label, rd, wr, launches), f, w in zip(GROUP_CASES[group], *per_case):
            rb, wb = round(f * 1024.0 * 2.0), round(w * 1024.0)
            doc["cases"].append(dict(kernel=group, label=label, launches_per_step=launches, calibration_shape=False,
                                     algorithmic_read_bytes=rd * elems, algorithmic_write_bytes=wr * elems, FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w,
                                     read_scale=2.0, read_scale_source="x2 (fp32 rows streamed in 128-B requests; sum over the member kernels)",
                                     read_bytes=rb, write_bytes=wb, traffic_over_algorithmic=round((rb + wb) / ((rd + wr) * elems), 3)))
            tot += launches * (rb + wb)
            n += launches
            print("%-100s read %7.1f MB (alg %6.1f)  write %6.1f MB (alg %6.1f)" % (label[:100], rb / 1e6, rd * elems / 1e6, wb / 1e6, wr * elems / 1e6))
        doc["per_launch_bytes"][group] = round(tot / n)
    with open(out_json, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--reduce-groups":
        reduce_groups(*sys.argv[2:5])
    elif len(sys.argv) >= 6 and sys.argv[1] == "--reduce":
        reduce(*sys.argv[2:6])
    else:
        run()
