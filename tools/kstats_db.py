"""per-kernel statistics from a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME writes DIR/NAME_results.db):
   python tools/kstats_db.py DIR_OR_DB [rows]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
db = path if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) from {kd} d "
     f"join {ks} s on d.kernel_id = s.id group by s.kernel_name order by sum(d.end-d.start) desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 14}")
print(f"{'kernel':88s} {'calls':>7s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}")
for r in c.execute(q):
    print(f"{r[0][:88]:88s} {r[1]:7d} {r[2] / 1e3:9.2f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f}")
