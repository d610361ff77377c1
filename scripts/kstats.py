"""Per-kernel summary (calls, avg us, total us, share) of a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (kd, ks)).fetchall()
tot = sum(r[3] for r in rows)
for name, n, avg, sm in rows:
    print("%7d %9.2f us %10.1f us %5.1f%%  %s" % (n, avg / 1e3, sm / 1e3, 100 * sm / tot, name[:110]))
