#!/usr/bin/env python3
"""Condense rocprofv3 output directories (under gpurun_out/) into the small,
tracked summaries kept in profiles/.

usage: tools/prof_summary.py <tag> --kt DIR [--fetch DIR] [--write DIR] [--bench-log FILE] [--cmd "..."]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim
rows) and profiles/<tag>_summary.md (per-kernel average duration, PMC HBM traffic
corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 64 B per
128-B request on wide coalesced reads, so it is doubled; WRITE_SIZE is exact;
both are in KiB).
"""
import argparse
import collections
import csv
import glob
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    m = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return m[0] if m else None


def short(name):
    name = name.replace("wr::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def pmc(d, counter):
    f = find(d, "*_counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--kt", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench-log")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    ks = find(a.kt, "*_kernel_stats.csv")
    shutil.copy(ks, os.path.join(out_dir, f"{a.tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    fetch = pmc(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = pmc(a.write, "WRITE_SIZE") if a.write else {}
    lines = [f"# rocprofv3 summary `{a.tag}`", ""]
    if a.cmd:
        lines += [f"Command: `{a.cmd}`", ""]
    lines += ["| kernel | calls | avg ms | min ms | max ms | % | FETCH_SIZE KiB/launch | HBM read GB (x2 gfx950 corr.) | WRITE_SIZE KiB/launch | HBM write GB |",
              "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        k = short(r["Name"])
        f, w = fetch.get(k), write.get(k)
        lines.append("| {} | {} | {:.4f} | {:.4f} | {:.4f} | {} | {} | {} | {} | {} |".format(
            k, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
            r["Percentage"],
            f"{f:.0f}" if f is not None else "-", f"{f * 1024 * 2 / 1e9:.3f}" if f is not None else "-",
            f"{w:.0f}" if w is not None else "-", f"{w * 1024 / 1e9:.3f}" if w is not None else "-"))
    if a.bench_log and os.path.exists(a.bench_log):
        js = [l for l in open(a.bench_log) if l.startswith("{")]
        if js:
            lines += ["", "bench.py line of the profiled (kernel-trace) run:", "", "```json", js[-1].strip(), "```"]
    open(os.path.join(out_dir, f"{a.tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    # machine-readable PMC traffic per launch (bytes), picked up by bench.py's roofline.traffic
    import json
    traffic = {}
    for r in rows:
        k = short(r["Name"])
        if k in fetch and k in write:
            traffic[k.split("<")[0]] = {"hbm_read_bytes": fetch[k] * 1024 * 2, "hbm_write_bytes": write[k] * 1024,
                                        "hbm_bytes": fetch[k] * 1024 * 2 + write[k] * 1024,
                                        "source": f"profiles/{a.tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                                  "passes; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md)"}
    if traffic:                                      # only a run with both PMC passes may replace the committed constants
        json.dump({"tag": a.tag, "kernels": traffic}, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
