#!/usr/bin/env python3
"""Turn rocprofv3 outputs under gpurun_out/ into the small, tracked summaries under profiles/.

  python tools/summarise_profiles.py stats  gpurun_out/prof_X  profiles/NAME.csv   # --kernel-trace --stats
  python tools/summarise_profiles.py pmc    gpurun_out/pmc_fetch gpurun_out/pmc_write [more pmc dirs] \
         --rows 16777216 --m 64 --out profiles/traffic_latest.json --csv profiles/NAME_pmc.csv

  python tools/summarise_profiles.py pmcall gpurun_out/pmc_A gpurun_out/pmc_B ... profiles/NAME.csv  # every kernel,
         one line per (kernel, grid size): mean counter values per dispatch, HBM bytes corrected as below

HBM traffic follows the gfx950 rules for rocprofv3: FETCH_SIZE and WRITE_SIZE are reported in KiB by separate
passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane streaming stores.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys


def stats(src, dst, timed_steps=20):
    f = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(dst, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Name(truncated to 100 chars)", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"]])
    # the kernel-trace of the same run: per-dispatch durations of the sampler kernel, so that the warm-up launches
    # (the clocks settle over the first ~10) can be told from the timed ones
    tr = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
    if tr:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr[0]))
             if "vmf_rsample_fwd" in r["Kernel_Name"]]
        if d:
            timed = d[-timed_steps:] if timed_steps and len(d) > timed_steps else d
            with open(dst, "a", newline="") as out:
                w = csv.writer(out)
                w.writerow([])
                w.writerow(["vmf_rsample_fwd per-dispatch ns (launch order)"] + d)
                w.writerow([f"mean of the last {len(timed)} dispatches (the timed steps) ns", sum(timed) / len(timed)])
            print("timed-step mean ms", sum(timed) / len(timed) / 1e6)
    print("wrote", dst)


def pmc(dirs, rows, m, out_json, out_csv, log_prob=True, kernel="vmf_rsample_fwd"):
    agg = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in agg.items()}
    with open(out_csv, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["counter", "dispatches", "mean_per_dispatch"])
        for k in sorted(mean):
            w.writerow([k, len(agg[k]), mean[k]])
    res = {"rows": rows, "m": m, "log_prob": bool(log_prob), "kernel": kernel}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        fetch_b = mean["FETCH_SIZE"] * 1024 * 2  # gfx950: doubled
        write_b = mean["WRITE_SIZE"] * 1024
        res.update({"fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
                    "algorithmic_bytes_per_launch": rows * (2 * m + 1 + (1 if log_prob else 0)) * 4})
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def pmcall(dirs, out_csv):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if name.startswith("void at::") or "elementwise_kernel" in name or "distribution" in name:
                    continue  # torch's own fills and random inputs of the bench script
                agg[(name[:90], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in agg.values() for c in v})
    with open(out_csv, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["kernel(truncated)", "grid_threads", "dispatches"] + names + ["hbm_read_bytes(FETCH_SIZE*1024*2)",
                                                                               "hbm_write_bytes(WRITE_SIZE*1024)"])
        for (k, g), v in sorted(agg.items()):
            mean = {c: sum(x) / len(x) for c, x in v.items()}
            w.writerow([k, g, max(len(x) for x in v.values())] + [f"{mean[c]:.6g}" if c in mean else "" for c in names] +
                       [f"{mean['FETCH_SIZE'] * 2048:.6g}" if "FETCH_SIZE" in mean else "",
                        f"{mean['WRITE_SIZE'] * 1024:.6g}" if "WRITE_SIZE" in mean else ""])
    print("wrote", out_csv)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmcall":
        pmcall(sys.argv[2:-1], sys.argv[-1])
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("mode")
        ap.add_argument("dirs", nargs="+")
        ap.add_argument("--rows", type=int, default=1 << 24)
        ap.add_argument("--m", type=int, default=64)
        ap.add_argument("--out", default="profiles/traffic_latest.json")
        ap.add_argument("--csv", default="profiles/pmc_latest.csv")
        ap.add_argument("--no-log-prob", dest="log_prob", action="store_false", default=True,
                        help="the passes ran bench.py --no-log-prob (516 B/row instead of 520)")
        a = ap.parse_args()
        pmc(a.dirs, a.rows, a.m, a.out, a.csv, a.log_prob)
