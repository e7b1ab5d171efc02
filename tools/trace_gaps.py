"""Timeline of ONE fit from a rocprofv3 --kernel-trace CSV: every kernel between the last two launches of the fit's first
kernel (default: fit_probe_kernel), with start offset, duration and the idle gap in front of it.
Usage: python tools/trace_gaps.py <dir-with-*_kernel_trace.csv> [marker-substring]"""
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "fit_probe_kernel"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if marker in r[2]]
if len(idx) < 2:
    sys.exit(f"marker {marker!r} seen {len(idx)} times")
a, b = idx[-2], idx[-1]
t0 = rows[a][0]
busy = 0.0
prev_end = t0
print(f"# one fit: kernels {a}..{b - 1} of the trace; times in microseconds")
print(f"{'start':>10} {'dur':>10} {'gap':>8}  kernel")
gaps = 0.0
for s, e, name in rows[a:b]:
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    busy += (e - s) / 1e3
    short = name.split("(")[0][-70:]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:10.1f} {gap:8.1f}  {short}")
    prev_end = max(prev_end, e)
print(f"# span {(rows[b][0] - t0) / 1e3:.1f} us, kernels busy {busy:.1f} us, idle {gaps:.1f} us (+ {(rows[b][0] - prev_end) / 1e3:.1f} us before the next fit)")
