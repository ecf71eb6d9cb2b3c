#!/usr/bin/env python3
"""Digests gpurun_out/prof_<tag>/ (written by tools/profile.sh) into profiles/<tag>_*.csv|json:
the rocprofv3 --stats kernel summary as is, and per-kernel averages of every PMC counter."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """mcamd::kernel<template args> without the parameter list (kernel names in rocprofv3 CSVs are demangled)."""
    import re
    m = re.search(r"mcamd::(\w+(?:<[^()]*?>)?)\(", name)
    return m.group(1) if m else name[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # gpurun merges every call's output into the same scratch directory: a directory profiled twice holds two runs.
    # Only the newest file of each pass is digested (one run of the command per pass).
    newest = lambda pattern: sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1:]
    for f in newest(os.path.join(src, "trace", "**", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    agg = defaultdict(lambda: defaultdict(list))
    passes = sorted(d for d in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d))
    for f in [x for d in passes for x in newest(os.path.join(d, "**", "*_counter_collection.csv"))]:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())}
           for k, cs in agg.items() if "kernel" in k}
    # the library the counters were taken from (tools/profile.sh writes its mcamd_build_id beside the passes): bench.py
    # quotes valu_busy from this digest only for that library
    try:
        out["build_id"] = open(os.path.join(src, "build_id.txt")).read().strip()
    except OSError:
        pass
    with open(os.path.join(dst, f"{tag}_pmc_per_kernel.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
