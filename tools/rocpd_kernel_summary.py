#!/usr/bin/env python3
"""Per-kernel call count / mean / max duration from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -o x` writes
x_results.db when no --output-format is given).  usage: rocpd_kernel_summary.py x_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start), max(d.end-d.start) from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for r in cur.execute(q): print("%-60s %4d %10.1f us %10.1f" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3))
