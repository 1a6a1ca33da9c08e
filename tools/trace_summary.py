"""Summarise a rocprofv3 --kernel-trace CSV: last steps' kernel timeline (start, duration, gap)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f'{r["Kernel_Name"][:60]:60s} dur {(e - s) / 1e3:9.1f} us  gap {gap:9.1f} us  grid {r["Grid_Size_X"]}')
    prev = e
