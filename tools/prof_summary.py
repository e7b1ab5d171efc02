"""Summarise rocprofv3 output (CSV kernel trace [+ counter collection]) into a small text table.
Usage: python tools/prof_summary.py <dir-with-*_kernel_trace.csv> [> profiles/xxx.txt]"""
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
agg = collections.OrderedDict()
for f in kt:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0, r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size_X"), r.get("Workgroup_Size_X")])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = sum(a[1] for a in agg.values())
print(f"# rocprofv3 --kernel-trace summary of {d}  (durations in microseconds)")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>11} {'min_us':>11} {'max_us':>11} {'pct':>6} {'vgpr':>5} {'lds':>6}  kernel")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    short = name if len(name) < 110 else name[:107] + "..."
    print(f"{a[0]:6d} {a[1]:12.1f} {a[1]/a[0]:11.2f} {a[2]:11.2f} {a[3]:11.2f} {100*a[1]/tot:6.2f} {a[4]:>5} {a[5]:>6}  {short}")
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if cc:
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in cc:
        for r in csv.DictReader(open(f)):
            pm[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n# PMC counters (mean per dispatch)")
    for name, cs in pm.items():
        if "eofx" not in name:
            continue
        print(name[:100])
        for c, v in cs.items():
            print(f"    {c:28s} {sum(v)/len(v):18.1f}  (n={len(v)})")
