#!/usr/bin/env python3
"""Copy the judged summaries of a rocprofv3 run from gpurun_out/ (scratch) into profiles/.

    python tools/summarize_profile.py r1b    # reads gpurun_out/prof_r1b_{trace,fetch,write}

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim) and
profiles/<tag>_hbm_traffic.md (per-kernel FETCH_SIZE / WRITE_SIZE from the two --pmc passes,
with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md section HBM).
"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    lines = [f"# HBM traffic per launch ({tag})", "",
             "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, counter unit KiB.",
             "gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of a wide "
             "coalesced streaming read, so the read column is also shown doubled; WRITE_SIZE is "
             "exact for 16-B-per-lane streaming stores.", "",
             "| kernel | launches | FETCH_SIZE KiB (raw) | read MB (x2 corrected) | WRITE_SIZE KiB | write MB |",
             "|---|---|---|---|---|---|"]
    fetch = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_fetch", "*", "*_counter_collection.csv"))
    write = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_write", "*", "*_counter_collection.csv"))
    f = per_kernel(fetch[0], "FETCH_SIZE") if fetch else {}
    w = per_kernel(write[0], "WRITE_SIZE") if write else {}
    summary = {}
    for k in sorted(set(f) | set(w)):
        if "anonymous namespace" not in k:
            continue
        fv = sorted(f.get(k, [0.0]))
        wv = sorted(w.get(k, [0.0]))
        fm, wm = fv[len(fv) // 2], wv[len(wv) // 2]
        name = k.split("(anonymous namespace)::")[1].split("(")[0]
        lines.append(f"| {name} | {len(fv)} | {fm:.1f} | {2 * fm * 1024 / 1e6:.2f} | {wm:.1f} | "
                     f"{wm * 1024 / 1e6:.2f} |")
        summary[name] = {"fetch_kib_raw": fm, "write_kib": wm,
                         "hbm_bytes_per_launch": int((2 * fm + wm) * 1024),
                         "launch": "B=4096, 16x16 grid, 10 agents"}
    lines.append("")
    lines.append("Values are the median over the launches of one bench.py run (B = 4096 boards/envs "
                 "of 16x16/10 per launch); connector_kernel mixes 1 reset + N step launches.")
    open(os.path.join(out, f"{tag}_hbm_traffic.md"), "w").write("\n".join(lines) + "\n")
    import json
    json.dump({"tag": tag, "kernels": summary},
              open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
