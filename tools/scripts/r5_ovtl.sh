#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_ovtl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FTCF_FAKE_AR_US=20 FTCF_DECODE_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $O/tr -o r -- python $R/bench.py --fake-tp 8 --batch 32 --prompt-len 256 --output-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-pmc > /dev/null 2> $O/tr.err
python - $(find $O/tr -name "*results.db" | head -1) > $O/timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
nc = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(c.execute(f"select s.{nc}, d.stream_id, d.queue_id, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
# the last token's decode: find the last 400 dispatches, print 90 of them from a spin kernel on
rows = rows[-700:-300]
i0 = next(i for i, r in enumerate(rows) if "spin" in r[0])
t0 = rows[i0][3]
for n, st, q, a, b in rows[i0:i0 + 90]:
    short = n.split("(")[0].replace("ftcf::", "")[:44]
    print(f"{(a - t0) / 1e3:9.2f} {(b - t0) / 1e3:9.2f} us  stream {st} queue {q}  {short}")
PY
head -95 $O/timeline.txt
find $O -name "*.db" -delete
