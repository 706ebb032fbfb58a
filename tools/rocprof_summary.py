"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) per kernel: launches, total ms, average us.

    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start),
                               max(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size)
                               from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""))
    total = sum(r[2] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, {total/1e6:.2f} ms of kernel time")
    print(f"{'calls':>7} {'total_ms':>10} {'%':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>6}  kernel")
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds in rows:
        print(f"{n:7d} {tot/1e6:10.3f} {100*tot/total:6.2f} {avg/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:6d}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
