"""Times every problem of profiles/gemm_shapes.json that carries a residual (flag 8) on the library given as argument (default: the product
build) - a library-level A/B is two invocations on the same box, interleaved by the caller (tools/_abl/libvcx_<name>.so from
tools/build_abl.sh or a build of another commit's sources).

    python tools/res_ab.py [library.so] [--all]      # --all: every problem, not only the residual ones
"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from viewcrafter_amd import _lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    _lib.LIB_PATH = os.path.abspath(args[0])
from gemm_shapes import FIELDS, _time, make_problem
rows = json.load(open(os.path.join(ROOT, "profiles", "gemm_shapes.json")))["rows"]
tot = 0.0
out = []
for r in rows:
    if r["mode"] == 2 or (not (r["flags"] & 8) and "--all" not in sys.argv):
        continue
    run = make_problem(tuple(r[f] for f in FIELDS))
    ms = min(_time(run, 5) for _ in range(3))
    tot += ms * r["count"]
    out.append(f"{r['count']:3d} {r['M']:7d} {r['N']:5d} {r['K']:6d} {r['mode']} {r['flags']:4d} {ms:7.3f}")
    del run
    torch.cuda.empty_cache()
print("library", _lib.LIB_PATH)
print("\n".join(out))
print(f"total {tot:.3f} ms per forward pair")
