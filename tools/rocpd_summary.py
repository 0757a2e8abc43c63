"""Turn a rocprofv3 (ROCm 7.2, rocpd SQLite output) result into the per-kernel summary kept under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db profiles/round1_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 3)])
    print("wrote %s (%d kernels, %.3f ms total)" % (out, len(rows), total / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
