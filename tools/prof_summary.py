"""Condense a rocprofv3 --kernel-trace --stats CSV into a short table (name, calls, avg us, total ms, %)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'%':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = r["Name"]
    n = n.replace("void ", "").split("(")[0][:70]
    print(f"{n:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
