"""Per-kernel summary of a rocprofv3 --kernel-trace results database (rocpd sqlite): count, mean / min duration, total.
Usage: python scratch/trace_summary.py gpurun_out/<dir>/<name>_results.db [top]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = c.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by 5 desc").fetchall()
span = c.execute("select min(start), max(end) from kernels").fetchone()
print("span %.2f ms, kernel time %.2f ms" % ((span[1] - span[0]) / 1e6, sum(r[4] for r in rows) / 1e6))
for r in rows[:top]:
    print("%-64s n=%6d avg=%9.2f us min=%8.2f total=%9.2f ms" % (r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e6))
