"""Kernel timeline of ONE pass from a rocprofv3 --kernel-trace result (rocpd SQLite): every kernel of the last `n` launches'
window with start offset, duration, stream / queue, plus the union busy time, the idle gaps and the overlap.
    python tools/timeline_r4.py <results.db> <out.txt> [passes_in_trace]
"""
import sqlite3
import sys


def main(db, out, passes=10):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    if not cols:      # a view
        cols = [d[0] for d in c.execute("select * from kernels limit 1").description]
    pick = lambda *names: next((n for n in names if n in cols), None)
    cs, ce, cn = pick("start", "start_timestamp"), pick("end", "end_timestamp"), pick("name", "kernel_name")
    cq = pick("stream_id", "queue_id", "stream", "queue")
    rows = list(c.execute("select %s, %s, %s, %s from kernels order by %s" % (cn, cs, ce, cq or "0", cs)))
    n = len(rows) // passes
    rows = rows[-n:]                      # the last pass
    t0 = rows[0][1]
    busy, cur_s, cur_e, gaps = 0, rows[0][1], rows[0][2], []
    for nm, s, e, q in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = max(r[2] for r in rows) - t0
    tot = sum(r[2] - r[1] for r in rows)
    with open(out, "w") as f:
        f.write("# columns: %s ; queue column: %s\n" % (",".join(cols), cq))
        f.write("# last pass: %d kernels, span %.3f ms, union busy %.3f ms, idle %.3f ms in %d gaps, sum of durations %.3f ms (overlap %.3f ms)\n"
                % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), tot / 1e6, (tot - busy) / 1e6))
        big = sorted(gaps, reverse=True)[:10]
        f.write("# largest gaps (us @ offset us): %s\n" % ", ".join("%.1f@%.0f" % (g / 1e3, o / 1e3) for g, o in big))
        for nm, s, e, q in rows:
            short = nm.replace("void pg::", "").replace("pg::", "")
            short = short.split("(")[0][:70]
            f.write("%9.1f %8.1f q%-3s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, q, short))
    print(open(out).read().splitlines()[1])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10)
