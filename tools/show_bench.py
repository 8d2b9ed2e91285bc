#!/usr/bin/env python3
"""Developer helper: the figures of one bench.py line (a JSON file) at a glance."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"boards/s {d['value'] / 1e6:.1f} M  ms/step {d['ms_per_step'] * 1e3:.2f} us  events "
      f"{d['roofline']['kernel_avg_ms'] * 1e3:.2f} us  serial frac {d['roofline']['serial']['frac']:.3f}")
if "c4" in d:
    print(f"c4 {d['c4']['value'] / 1e6:.1f} M ({d['c4']['ms_per_step'] * 1e3:.1f} us)  strong "
          f"{d['strong']['value'] / 1e6:.1f} M ({d['strong']['ms_per_step'] * 1e3:.2f} us)")
e = d.get("env_step")
if e:
    print("env", round(e["value"] / 1e6, 1), e["value_variant"],
          {k: round(v / 1e6, 1) for k, v in e["variants"].items()})
    print("  us/step", {k: round(v * 1e3, 2) for k, v in e["ms_per_step"].items()})
    r = e["roofline"]
    print(f"  step kernel {r['kernel_avg_ms'] * 1e3:.2f} us frac {r['frac']:.3f} with refills "
          f"{r['span_with_refills_ms'] * 1e3:.2f} us")
    print("  api", {k: round(v / 1e6, 1) for k, v in e["wrapper_api"].items() if k != "note"})
    for k, v in e["by_generation_type"].items():
        if isinstance(v, dict):
            st = v["stash"]
            print(f"  {k:24s} {v['us_per_step']:6.2f} us  wall {v.get('host_wall_us_per_step', 0):6.2f}  "
                  f"tail {v.get('refills_in_flight_at_the_end_ms', 0):.2f} ms  "
                  f"{'no stash' if isinstance(st, str) else (st['hits'], st['misses'])}")
    ps = e.get("pure_step_stream")
    if ps:
        print(f"  stream frac {ps['frac']:.3f}")
ro = d.get("rollout")
if ro:
    rec = ro["recorded"]
    print(f"rollout {ro['value'] / 1e9:.2f} G ({ro['launch_avg_ms'] * 1e3:.0f} us) fused resets "
          f"{ro['fused_resets']['value'] / 1e9:.2f} G  recorded {rec['value'] / 1e6:.0f} M "
          f"({rec['launch_avg_ms'] * 1e3:.0f} us, {rec['roofline']['frac']:.3f})")
c = d.get("cpu_baseline")
if c:
    print(f"cpu {c['value'] / 1e3:.1f} k boards/s on {c['cores']} threads")
if "philox_stream" in d:
    print(f"philox {d['philox_stream']['value_per_gpu'] / 1e6:.1f} M")
