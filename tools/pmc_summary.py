#!/usr/bin/env python3
"""Per-kernel mean of the PMC counters in a rocprofv3 rocpd database (one row per dispatch and counter)."""
import sqlite3, sys, collections
db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
disp, sym, pmc, info = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
print([r[1] for r in c.execute(f"pragma table_info({pmc})")])
print([r[1] for r in c.execute(f"pragma table_info({info})")])
scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
q = (f"select s.{name_col}, i.name, count(*), avg(e.value), avg(d.end-d.start) from {pmc} e join {disp} d on e.event_id = d.event_id "
     f"join {sym} s on d.kernel_id = s.id join {info} i on e.pmc_id = i.id group by s.{name_col}, i.name order by 1,2")
rows = list(c.execute(q))
per = collections.defaultdict(dict)
for k, n, cnt, v, dur in rows:
    per[k][n] = (cnt, v, dur)
for k, d in per.items():
    if "ftcf" not in k:
        continue
    short = k.split("(")[0].replace("_ZN4ftcf", "")[:40]
    dur = list(d.values())[0][2] / 1e3
    print(f"{short:<42} calls {list(d.values())[0][0]:>6} dur_us {dur:8.2f}  " + "  ".join(f"{n}={v[1]:.4g}" for n, v in sorted(d.items())))
