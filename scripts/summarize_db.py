"""Per-kernel summary of a rocprofv3 results database (kernel trace): calls, average / total duration."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), avg(d.end - d.start), sum(d.end - d.start) from {kd} d "
                        f"join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 4 desc"))
tot = sum(r[3] for r in rows)
print("| kernel | calls | avg us | total ms | share |\n|---|---|---|---|---|")
for name, n, avg, s in rows[:16]:
    print(f"| `{name[:90]}` | {n} | {avg / 1e3:.1f} | {s / 1e6:.2f} | {100 * s / tot:.1f} % |")
