"""Matrix-pipe occupancy of every libvcx kernel over one DDIM step, from one rocprofv3 --pmc pass (tools/pmc_mfma_step.sh):
occupancy in cycles = SQ_VALU_MFMA_BUSY_CYCLES / (32 SQ_BUSY_CYCLES) (the normalisation of profiles/r02_experiments.md section 1 and r05_experiments.md
section 14), waves parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES.      python tools/pmc_mfma_step.py <results.db>"""
import collections
import sqlite3
import sys

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
out = []
for k, d in rows.items():
    if "GLOBAL__N_1" not in k or "at6native" in k or "SQ_BUSY_CYCLES" not in d:
        continue
    n, busy = d["SQ_BUSY_CYCLES"]
    mfma = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]
    wave, wait = d.get("SQ_WAVE_CYCLES", (0, 0.0))[1], d.get("SQ_WAIT_ANY", (0, 0.0))[1]
    out.append((busy, n, mfma / (32.0 * busy) if busy else 0.0, wait / wave if wave else 0.0, k))
tot = sum(o[0] for o in out)
print(f"{'dispatches':>10s} {'busy cycles %':>13s} {'matrix pipe':>11s} {'waves parked':>12s}  kernel")
for busy, n, occ, parked, k in sorted(out, reverse=True):
    name = k.replace("_ZN12_GLOBAL__N_1", "").replace("EvN7vcxgemm8GemmArgsEjj.kd", "").replace("EvN7vcxgemm8GemmArgsEj.kd", "")[:90]
    print(f"{n:10d} {100 * busy / tot:13.2f} {occ:11.3f} {parked:12.3f}  {name}")
mf = sum(o[2] * o[0] for o in out) / tot
print(f"whole step, weighted by busy cycles: matrix pipe {mf:.3f}")
