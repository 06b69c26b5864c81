#!/usr/bin/env python3
"""Prints the kernel timeline (start offset, duration, queue) of a slice of a rocprofv3 rocpd trace."""
import sqlite3, sys
db = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 20000; n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(c.execute(f"select d.start, d.end, d.queue_id, d.stream_id, s.{name_col} from {disp} d join {sym} s on d.kernel_id=s.id order by d.start limit {n} offset {skip}"))
t0 = rows[0][0]
prev_end = t0
for st, en, q, sid, nm in rows:
    short = nm.split("(")[0].replace("_ZN4ftcf", "")[:38]
    print(f"+{(st-t0)/1e3:9.2f}us dur {(en-st)/1e3:7.2f}us gap {(st-prev_end)/1e3:7.2f} q{q} s{sid} {short}")
    prev_end = max(prev_end, en)
