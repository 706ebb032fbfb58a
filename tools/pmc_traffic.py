"""HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950
(checked here on the 460800x320x320+residual GEMM: 2 x 288 MB raw = 576 MB vs 590 MB algorithmic).
    python tools/pmc_traffic.py fetch.db write.db [steps_in_fetch_run] [steps_in_write_run]"""
import collections, sqlite3, sys

def per_kernel(path, counter):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tables if t.startswith(p)][0]
    q = f"""select s.kernel_name, count(distinct d.id), sum(e.value) from {T('rocpd_pmc_event')} e
            join {T('rocpd_info_pmc')} p on e.pmc_id = p.id join {T('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
            join {T('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id where p.name = '{counter}' group by s.kernel_name"""
    return {k: (n, v) for k, n, v in db.execute(q)}

fam = lambda k: ("gemm" if "gemm_" in k else "flash_attn" if "flash_d64" in k else "temporal_attn" if "tattn" in k else
                 "groupnorm" if "gn_" in k else "layernorm" if "layernorm" in k else None)
f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
sf = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
sw = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k, (n, v) in f.items():
    if fam(k): agg[fam(k)][0] += n / sf; agg[fam(k)][1] += 2.0 * v * 1024 / sf
for k, (n, v) in w.items():
    if fam(k): agg[fam(k)][2] += v * 1024 / sw
print(f"{'family':14s} {'launches/step':>13s} {'read GB/step (2x FETCH)':>24s} {'write GB/step':>14s} {'bytes/launch':>14s}")
for k, (n, r, wr) in agg.items():
    print(f"{k:14s} {n:13.0f} {r/1e9:24.1f} {wr/1e9:14.1f} {(r+wr)/max(n,1):14.3e}")
