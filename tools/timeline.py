"""Gap analysis of a rocprofv3 --kernel-trace csv: per kernel name, mean duration and mean idle time between the previous kernel's end
and this kernel's start (all queues merged, sorted by start).  Usage: python tools/timeline.py <dir with *_kernel_trace.csv> [skip_first_n]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Queue_Id", "")) for r in rows))
ev = ev[skip:]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
prev_end = None
for s, e, name, q in ev:
    if prev_end is not None:
        gap[name].append(s - prev_end)
    dur[name].append(e - s)
    prev_end = max(prev_end or 0, e)
span = ev[-1][1] - ev[0][0]
print(f"{len(ev)} kernels over {span/1e3:.1f} us")
for name in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap[name]
    print(f"{name:50s} n={len(dur[name]):6d} dur mean {sum(dur[name])/len(dur[name])/1e3:8.2f} us   gap-before mean {sum(g)/max(len(g),1)/1e3:8.2f} us  (min {min(g)/1e3 if g else 0:.2f}, max {max(g)/1e3 if g else 0:.2f})")
busy = sum(sum(v) for v in dur.values())
print(f"busy {busy/1e3:.1f} us = {100*busy/span:.1f} % of the span (overlap between queues counted twice)")
