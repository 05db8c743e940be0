#!/usr/bin/env python3
"""per-kernel statistics (count, mean / min us) from a rocprofv3 rocpd database or directory of them"""
import glob, os, sqlite3, sys
paths = []
for a in sys.argv[1:]:
    paths += glob.glob(os.path.join(a, "**", "*.db"), recursive=True) if os.path.isdir(a) else [a]
for db in paths:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 5 desc"
    print("# %s" % db)
    print("%-110s %6s %10s %10s %12s" % ("kernel", "calls", "mean_us", "min_us", "total_us"))
    for r in c.execute(q):
        print("%-110s %6d %10.1f %10.1f %12.1f" % (r[0][:110], r[1], r[2], r[3], r[4]))
