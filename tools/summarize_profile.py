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
    issue = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_issue", "*", "*_counter_collection.csv"))
    if issue:
        write_issue(tag, issue[0], out)


def write_issue(tag, path, out):
    """SQ instruction / cycle counters per kernel (one more --pmc pass): how busy the vector
    issue ports are.  SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* and SQ_WAIT_* count quad-cycles
    (MI355X_MICROARCH.md, constants table)."""
    names = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
             "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY")
    per = {n: per_kernel(path, n) for n in names}
    lines = [f"# Issue counters per launch ({tag})", "",
             "rocprofv3 --pmc " + " ".join(names) + " (one pass; medians over the launches of one",
             "bench.py run, B = 4096, 16x16/10).  Cycle counters are quad-cycles.", "",
             "| kernel | waves | VALU | SALU | LDS | instr / wave | wave quad-cycles | active any | "
             "waiting (s_waitcnt) | VALU quad-cycles / wave quad-cycles |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(per["SQ_WAVES"]):
        if "anonymous namespace" not in k:
            continue
        med = {n: sorted(per[n].get(k, [0.0]))[len(per[n].get(k, [0.0])) // 2] for n in names}
        if med["SQ_WAVES"] == 0:
            continue
        name = k.split("(anonymous namespace)::")[1].split("(")[0]
        total = med["SQ_INSTS_VALU"] + med["SQ_INSTS_SALU"] + med["SQ_INSTS_LDS"]
        wc = max(med["SQ_WAVE_CYCLES"], 1.0)
        lines.append(f"| {name} | {med['SQ_WAVES']:.0f} | {med['SQ_INSTS_VALU']:.0f} | "
                     f"{med['SQ_INSTS_SALU']:.0f} | {med['SQ_INSTS_LDS']:.0f} | "
                     f"{total / med['SQ_WAVES']:.0f} | {wc:.0f} | "
                     f"{med['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {med['SQ_WAIT_ANY'] / wc:.2f} | "
                     f"{med['SQ_ACTIVE_INST_VALU'] / wc:.2f} |")
    lines.append("")
    open(os.path.join(out, f"{tag}_issue.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
