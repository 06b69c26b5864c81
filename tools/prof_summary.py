#!/usr/bin/env python3
"""Per-kernel summary (calls, avg/total duration, share) of a rocprofv3 rocpd SQLite trace -> text for profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = (f"select s.{name_col}, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 4 desc")
    rows = list(c.execute(q))
    total = sum(r[3] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {db}", f"# columns: {cols}",
             f"{'kernel':<70} {'calls':>8} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'pct':>6}"]
    for n, cnt, avg, tot, mn, mx in rows:
        short = n.split("(")[0][-70:]
        lines.append(f"{short:<70} {cnt:>8} {avg/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {tot/1e6:>10.3f} {100*tot/total:>6.2f}")
    lines.append(f"{'TOTAL':<70} {sum(r[1] for r in rows):>8} {'':>10} {'':>10} {'':>10} {total/1e6:>10.3f} {100:>6.1f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
