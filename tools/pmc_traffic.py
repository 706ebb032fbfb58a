"""HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950
(checked in round 1 on the 460800x320x320+residual GEMM: 2 x 288 MB raw = 576 MB vs 590 MB algorithmic).
    python tools/pmc_traffic.py fetch.db write.db [--json out.json --algo-bytes-per-step B --gemm-calls-per-step N --commit C]
The number of DDIM steps in each pass is read from the trace itself (one ddim_update_kernel dispatch per step)."""
import argparse, collections, hashlib, json, os, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("fetch_db"); ap.add_argument("write_db")
ap.add_argument("--json"); ap.add_argument("--algo-bytes-per-step", type=float, default=None)
ap.add_argument("--gemm-calls-per-step", type=float, default=460.0); ap.add_argument("--commit", default=None)
a = ap.parse_args()


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tables if t.startswith(p)][0]
    q = f"""select s.kernel_name, count(distinct d.id), sum(e.value) from {T('rocpd_pmc_event')} e
            join {T('rocpd_info_pmc')} p on e.pmc_id = p.id join {T('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
            join {T('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id where p.name = '{counter}' group by s.kernel_name"""
    return {k: (n, v) for k, n, v in db.execute(q)}


fam = lambda k: ("gemm" if "gemm_" in k else "flash_attn" if ("flash_d64" in k or "flash2_d64" in k or "xattn_resident" in k) else "temporal_attn" if "tattn" in k else
                 "groupnorm" if "gn_" in k else "layernorm" if "layernorm" in k else None)
f = per_kernel(a.fetch_db, "FETCH_SIZE")
w = per_kernel(a.write_db, "WRITE_SIZE")
steps = lambda d: max(1.0, float(sum(n for k, (n, _) in d.items() if "ddim_update" in k)))
sf, sw = steps(f), steps(w)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k, (n, v) in f.items():
    if fam(k): agg[fam(k)][0] += n / sf; agg[fam(k)][1] += 2.0 * v * 1024 / sf
for k, (n, v) in w.items():
    if fam(k): agg[fam(k)][2] += v * 1024 / sw
print(f"# DDIM steps in the FETCH pass: {sf:.0f}, in the WRITE pass: {sw:.0f}")
# per kernel (template instantiation) of the GEMM family: which variants carry the read traffic
import re
short = lambda k: re.sub(r"^.*?(gemm_\w+kernel)I(?:NS_)?\d*(?:TileCfg|PPCfg)?I?", r"\1<", k)[:90]
print(f"{'gemm kernel':92s} {'disp/step':>9s} {'read GB/step':>12s} {'write GB/step':>13s}")
for k in sorted((k for k in f if fam(k) == "gemm"), key=lambda k: -f[k][1]):
    print(f"{short(k):92s} {f[k][0]/sf:9.0f} {2.0*f[k][1]*1024/sf/1e9:12.1f} {w.get(k,(0,0))[1]*1024/sw/1e9:13.1f}")
print(f"{'family':14s} {'launches/step':>13s} {'read GB/step (2x FETCH)':>24s} {'write GB/step':>14s} {'bytes/launch':>14s}")
for k, (n, r, wr) in agg.items():
    print(f"{k:14s} {n:13.0f} {r/1e9:24.1f} {wr/1e9:14.1f} {(r+wr)/max(n,1):14.3e}")
if a.json:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import csrc_hash                    # ONE definition of "the GEMM sources" (bench.py compares against it)
    hgemm, ha = csrc_hash(), csrc_hash(("attention.hip", "attention_v2.hip"))
    n, r, wr = agg["gemm"]
    out = {"family": "gemm", "hbm_bytes_per_launch": (r + wr) / a.gemm_calls_per_step, "hbm_read_gb_per_step": r / 1e9,
           "hbm_write_gb_per_step": wr / 1e9, "kernel_dispatches_per_step": n, "launches_per_step": a.gemm_calls_per_step,
           "algorithmic_bytes_per_launch": (a.algo_bytes_per_step / a.gemm_calls_per_step) if a.algo_bytes_per_step else None,
           "note": "per vcx_gemm_f16 call; FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as reported; separate --pmc passes",
           "source": "tools/pmc_passes.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --steps 1 --warmup 0 ...)",
           "csrc_sha256": hgemm, "attention_sha256": ha, "commit": a.commit, "workload": "ViewCrafter_25_576x1024x25",
           "families": {k: {"launches_per_step": v[0], "read_gb_per_step": v[1] / 1e9, "write_gb_per_step": v[2] / 1e9,
                                "hbm_bytes_per_launch": (v[1] + v[2]) / max(v[0], 1)} for k, v in agg.items()}}
    json.dump(out, open(a.json, "w"), indent=1)
