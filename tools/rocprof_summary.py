"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as a CSV:
per-kernel calls / total / average duration.  Usage: rocprof_summary.py <results.db> <out.csv>"""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for n, k, s, a, mn, mx in rows:
            f.write('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.2f\n' % (n.replace('"', "'"), k, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    print("wrote", out, "total kernel ms", tot / 1e6)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
