"""The float32 rounding audit of SequentialRandomWalk's integer Gumbel-top-1 rule over the
shapes and keys of the large-sample GPU sweep (tests/test_gpu_parity.py::SEQ_SWEEP at
RBG_SWEEP_SCALE, default 64 = the 3.6 M boards DESIGN.md quotes).  CPU only.

  python tools/seq_rw_rounding_audit.py [--scale 64] [--threads 8]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SEQ_SWEEP = [(16, 10, 14), (10, 5, 13), (8, 8, 13), (6, 5, 13), (10, 9, 12), (16, 14, 12),
             (5, 6, 12), (7, 7, 12), (12, 12, 11), (24, 20, 10), (9, 4, 12), (40, 30, 7)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=64)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="", help="write the rows (with board indices) to this JSON file")
    args = ap.parse_args()
    from oracle import oracle as O

    total = {"boards": 0, "draws": 0, "differs": 0, "within_4ulp": 0, "min_ulps": 2 ** 31 - 1}
    rows = []
    for m, N, log2b in SEQ_SWEEP:
        B = (1 << log2b) * args.scale
        parts = []
        for lo in range(0, B, 32768):  # the sweep's keys, chunk by chunk
            keys = O.split(O.prng_key(2000 + m * 64 + N + lo), min(32768, B - lo))
            parts.append(O.seq_rw_rounding_audit(keys, m, N, threads=args.threads))
        a = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        row = {"m": m, "N": N, "boards": B, "draws": int(a["draws"].astype(np.int64).sum()),
               "differs": int(a["differs"].sum()), "within_4ulp": int(a["within_4ulp"].sum()),
               "min_ulps": int(a["min_ulps"].min()),
               "boards_with_a_sensitive_draw": np.nonzero(a["within_4ulp"] + a["differs"])[0].tolist()}
        rows.append(row)
        print(json.dumps({k: v for k, v in row.items() if k != "boards_with_a_sensitive_draw"}
                         | {"n_boards_with_a_sensitive_draw": len(row["boards_with_a_sensitive_draw"])}),
              flush=True)
        for k in ("boards", "draws", "differs", "within_4ulp"):
            total[k] += row[k]
        total["min_ulps"] = min(total["min_ulps"], row["min_ulps"])
    total["boards_with_a_sensitive_draw"] = sum(len(r["boards_with_a_sensitive_draw"]) for r in rows)
    print(json.dumps({"total": total}))
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"keys": "split(PRNGKey(2000 + m * 64 + N + lo), min(32768, B - lo)) for lo in "
                               "range(0, B, 32768) -- tests/test_gpu_parity.py::"
                               "test_many_sequential_boards_match_oracle at RBG_SWEEP_SCALE=%d; board "
                               "indices are positions in that concatenation" % args.scale,
                       "libm": "glibc logf", "rows": rows, "total": total}, f)


if __name__ == "__main__":
    main()
