"""Is the tile-configuration rule of vcx_gemm_f16 (csrc/gemm.hip: 256-row tiles from 384 tiles on, tail rows to the small
configuration) the best choice for every problem of the forward?  Times each problem of profiles/gemm_shapes.json (the recorded
descriptors: no model build) under the product dispatch and under each forced tile configuration (knob GEMM_CFG = 0 .. 3:
128x128, 128x160, 256x256, 256x320 - a forced configuration also means the tiled engine where the product takes the
weight-stationary kernel), interleaved, min over rounds.

    python tools/gemm_cfg_scan.py [--rounds 3] [--iters 5] [--min-total-ms 0.3]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from viewcrafter_amd import ops  # noqa: E402
from gemm_shapes import FIELDS, _time, make_problem  # noqa: E402

CFG_NAME = {-1: "product", 0: "128x128", 1: "128x160", 2: "256x256", 3: "256x320"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=os.path.join(ROOT, "profiles", "gemm_shapes.json"))
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--min-total-ms", type=float, default=0.3, help="skip problems whose count x ms is below this")
    args = ap.parse_args()
    rows = json.load(open(args.shapes))["rows"]
    print(f"# {len(rows)} problems from {args.shapes}; ms per call, min over {args.rounds} interleaved rounds of {args.iters} launches")
    print(f"{'cnt':>4} {'M':>8} {'N':>6} {'K':>6} {'mode':>4} {'flags':>5} " + " ".join(f"{CFG_NAME[c]:>8}" for c in (-1, 0, 1, 2, 3)) + "   best     gain x cnt")
    tot_prod = tot_best = 0.0
    for r in rows:
        if r["total_ms"] < args.min_total_ms or r["mode"] == 2:
            continue
        key = tuple(r[f] for f in FIELDS)
        run = make_problem(key)
        geglu = bool(r["flags"] & 16)
        cfgs = [-1, 0, 2] if geglu else [-1, 0, 1, 2, 3]
        best = {c: float("inf") for c in cfgs}
        for _ in range(args.rounds):
            for c in cfgs:
                prev = ops.tune_set("GEMM_CFG", c)
                try:
                    best[c] = min(best[c], _time(run, args.iters))
                except Exception as e:      # a configuration this epilogue is not instantiated for
                    best[c] = float("nan")
                    print(f"#   {r['M']}x{r['N']}x{r['K']} flags {r['flags']} cfg {c}: {type(e).__name__}", file=sys.stderr)
                finally:
                    ops.tune_set("GEMM_CFG", prev)
        del run
        torch.cuda.empty_cache()
        ok = {c: v for c, v in best.items() if v == v}
        cb = min(ok, key=ok.get)
        gain = (best[-1] - ok[cb]) * r["count"]
        tot_prod += best[-1] * r["count"]
        tot_best += ok[cb] * r["count"]
        cells = " ".join(f"{best.get(c, float('nan')):8.3f}" if c in best else f"{'-':>8}" for c in (-1, 0, 1, 2, 3))
        mark = "  <--" if cb != -1 and best[-1] > 1.03 * ok[cb] else ""
        print(f"{r['count']:4d} {r['M']:8d} {r['N']:6d} {r['K']:6d} {r['mode']:4d} {r['flags']:5d} {cells}   {CFG_NAME[cb]:>8} {gain:7.3f}{mark}")
    print(f"# product dispatch {tot_prod:.2f} ms, per-problem best {tot_best:.2f} ms: {tot_prod - tot_best:.2f} ms per forward pair to gain from a perfect rule")


if __name__ == "__main__":
    main()
