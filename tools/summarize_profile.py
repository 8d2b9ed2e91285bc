#!/usr/bin/env python3
"""Copy the judged summaries of rocprofv3 runs from gpurun_out/ (scratch) into profiles/.

    python tools/summarize_profile.py r2 [suffix ...]

For every suffix s (default: "", i.e. the plain bench.py run) it reads
gpurun_out/prof_<tag><s>_{trace,fetch,write,issue} and writes
  profiles/<tag><s>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag><s>_hbm_traffic.md/json  FETCH_SIZE / WRITE_SIZE per kernel (two --pmc passes)
  profiles/<tag><s>_issue.md/json      SQ instruction / cycle counters (one more --pmc pass)
Suffixes name the single-population runs of bench.py: "_step" (--mode step-replay), "_stream"
(--mode stream).

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the
bytes of a WIDE coalesced streaming read (16 B per lane); it is doubled only for the kernels
whose reads are such streams (WIDE_READS); other access widths are uncalibrated and shown raw.
WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernels that read their state with global_load_dwordx4 streams (grids, 16 B per lane)
WIDE_READS = ("connector_step_kernel", "connector_step_policy_kernel", "connector_step_deep_kernel",
              "connector_step_deep_policy_kernel", "connector_observe_kernel",
              "connector_rollout_kernel", "connector_step_gen_kernel", "connector_rollout_gen_kernel")


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def short(k):
    return k.split("(anonymous namespace)::")[1].split("(")[0]


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def one(tag, out):
    def src(kind, pat):  # rocprofv3 writes into <dir>/ or <dir>/<host>/ depending on its version
        base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}")
        return glob.glob(os.path.join(base, pat)) + glob.glob(os.path.join(base, "*", pat))

    stats = src("trace", "*_kernel_stats.csv")
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    fetch, write = src("fetch", "*_counter_collection.csv"), src("write", "*_counter_collection.csv")
    if fetch or write:
        f = per_kernel(fetch[0], "FETCH_SIZE") if fetch else {}
        w = per_kernel(write[0], "WRITE_SIZE") if write else {}
        lines = [f"# HBM traffic per launch ({tag})", "",
                 "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, counter unit "
                 "KiB, medians over the launches of one bench.py run.",
                 "read MB = FETCH_SIZE x 2 for kernels whose reads are 16-B-per-lane streams "
                 "(gfx950 tallies 128-B requests at 64 B), raw for the others (uncalibrated "
                 "access width); WRITE_SIZE is exact for 16-B-per-lane streaming stores.", "",
                 "| kernel | launches | FETCH_SIZE KiB (raw) | read MB | x2 applied | WRITE_SIZE KiB | write MB |",
                 "|---|---|---|---|---|---|---|"]
        summary = {}
        for k in sorted(set(f) | set(w)):
            if "anonymous namespace" not in k:
                continue
            fm, wm = median(f.get(k, [0.0])), median(w.get(k, [0.0]))
            name = short(k)
            wide = name.split("<")[0] in WIDE_READS
            rd = (2 if wide else 1) * fm * 1024
            lines.append(f"| {name} | {len(f.get(k, w.get(k)))} | {fm:.1f} | {rd / 1e6:.2f} | "
                         f"{'yes' if wide else 'no'} | {wm:.1f} | {wm * 1024 / 1e6:.2f} |")
            summary[name] = {"fetch_kib_raw": fm, "write_kib": wm, "fetch_x2": wide,
                             "hbm_bytes_per_launch": int(rd + wm * 1024)}
        open(os.path.join(out, f"{tag}_hbm_traffic.md"), "w").write("\n".join(lines) + "\n")
        json.dump({"tag": tag, "kernels": summary},
                  open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)
        print("\n".join(lines))
    issue = src("issue", "*_counter_collection.csv")
    if issue:
        write_issue(tag, issue[0], out)


def write_issue(tag, path, out):
    """SQ instruction / cycle counters per kernel: how busy the vector issue ports are.
    SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* and SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)."""
    names = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
             "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY")
    per = {n: per_kernel(path, n) for n in names}
    lines = [f"# Issue counters per launch ({tag})", "",
             "rocprofv3 --pmc " + " ".join(names) + " (one pass; medians over the launches of one",
             "bench.py run).  Cycle counters are quad-cycles.", "",
             "| kernel | waves | VALU | SALU | LDS | instr / wave | wave quad-cycles | active any | "
             "waiting (s_waitcnt) | VALU quad-cycles / wave quad-cycles |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    summary = {}
    for k in sorted(per["SQ_WAVES"]):
        if "anonymous namespace" not in k:
            continue
        med = {n: median(per[n].get(k, [0.0])) for n in names}
        if med["SQ_WAVES"] == 0:
            continue
        name = short(k)
        total = med["SQ_INSTS_VALU"] + med["SQ_INSTS_SALU"] + med["SQ_INSTS_LDS"]
        wc = max(med["SQ_WAVE_CYCLES"], 1.0)
        lines.append(f"| {name} | {med['SQ_WAVES']:.0f} | {med['SQ_INSTS_VALU']:.0f} | "
                     f"{med['SQ_INSTS_SALU']:.0f} | {med['SQ_INSTS_LDS']:.0f} | "
                     f"{total / med['SQ_WAVES']:.0f} | {wc:.0f} | "
                     f"{med['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {med['SQ_WAIT_ANY'] / wc:.2f} | "
                     f"{med['SQ_ACTIVE_INST_VALU'] / wc:.2f} |")
        summary[name] = {"waves": med["SQ_WAVES"], "valu_insts_per_launch": med["SQ_INSTS_VALU"],
                         "salu_insts_per_launch": med["SQ_INSTS_SALU"],
                         "lds_insts_per_launch": med["SQ_INSTS_LDS"],
                         "wave_quad_cycles": med["SQ_WAVE_CYCLES"]}
    lines.append("")
    open(os.path.join(out, f"{tag}_issue.md"), "w").write("\n".join(lines) + "\n")
    json.dump({"tag": tag, "kernels": summary},
              open(os.path.join(out, f"{tag}_issue.json"), "w"), indent=1)
    print("\n".join(lines))


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    for suffix in (sys.argv[2:] or [""]):
        one(tag + suffix, out)


if __name__ == "__main__":
    main()
