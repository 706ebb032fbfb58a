"""Summarise rocprofv3 --pmc result db: per kernel, per counter: sum over dispatches (and per-dispatch mean)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tables if t.startswith(p)][0]
q = f"""select s.kernel_name, p.name, count(*), sum(e.value) from {T('rocpd_pmc_event')} e
        join {T('rocpd_info_pmc')} p on e.pmc_id = p.id
        join {T('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
        join {T('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id group by s.kernel_name, p.name"""
rows = collections.defaultdict(dict)
for k, c, n, v in db.execute(q):
    rows[k][c] = (n, v)
for k, d in rows.items():
    if "vcx" not in k and "GLOBAL__N_1" not in k:
        continue
    print(k[:100])
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:32s} n={n:4d} sum={v:16.0f} per_dispatch={v/max(n,1):14.1f}")
