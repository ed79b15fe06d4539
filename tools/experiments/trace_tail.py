"""tools/experiments/trace_tail.py <rocprofv3 output dir> [n] -- the last n kernels of a rocprofv3 --kernel-trace run as a timeline:
start (us from the first shown), duration, the gap in front, name, grid. GPU box (or wherever the trace lies)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0, prev = int(rows[0]["Start_Timestamp"]), None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"][:72]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {name}  grid {r.get('Grid_Size', r.get('Grid_Size_X'))} wg {r.get('Workgroup_Size', r.get('Workgroup_Size_X'))}")
    prev = e
