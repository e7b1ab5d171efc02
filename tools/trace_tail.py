"""Timeline of the LAST `ms` milliseconds of a rocprofv3 --kernel-trace CSV (kernels shorter than `min_us` and gaps shorter
than `min_us` are folded into one '... n small kernels' line).  Usage: python tools/trace_tail.py <dir> <ms> [min_us]"""
import csv, glob, sys
d, ms = sys.argv[1], float(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= end - ms * 1e6]
t0 = rows[0][0]
prev_end = t0
small_n, small_t = 0, 0.0
busy = idle = 0.0
def flush():
    global small_n, small_t
    if small_n:
        print(f"{'':>10} {small_t:10.1f} {'':>8}  ... {small_n} small kernels")
    small_n, small_t = 0, 0.0
print(f"{'start':>10} {'dur':>10} {'gap':>8}  kernel")
for s, e, name in rows:
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    busy += dur
    idle += max(gap, 0.0)
    if dur < min_us and gap < min_us:
        small_n += 1; small_t += dur
    else:
        flush()
        print(f"{(s - t0) / 1e3:10.1f} {dur:10.1f} {gap:8.1f}  {name.split('(')[0][-80:]}")
    prev_end = max(prev_end, e)
flush()
print(f"# window {(end - t0) / 1e3:.1f} us, kernels busy {busy:.1f} us (overlapping streams add up), idle {idle:.1f} us")
