"""Development: per-launch durations and the gaps between consecutive dispatches of a pipelined rollout, from a rocprofv3 --kernel-trace csv.
usage: gap_trace.py <dir with *_kernel_trace.csv> [first kernel name fragment of a step, default k_step]"""
import csv, glob, sys, collections
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_step"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void tg::", "").replace("tg::", "").split("(")[0][:40]) for r in rows]
ev = ev[len(ev) // 2:]                      # the second half of the run: steady state
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for k in range(1, len(ev)):
    s, e, name = ev[k]
    dur[name].append(e - s)
    gap[ev[k - 1][2] + " -> " + name].append(s - ev[k - 1][1])
steps = [ev[k][0] for k in range(len(ev)) if first in ev[k][2] and "bank" not in ev[k][2]]
per = [(b - a) for a, b in zip(steps, steps[1:])]
per.sort()
print("step-to-step (start of %s to the next): median %.2f us over %d steps" % (first, per[len(per) // 2] / 1e3, len(per)))
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("  %-42s n=%5d  median %7.2f us  mean %7.2f us" % (name, len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
print("gaps (end of one dispatch to the start of the next; negative = overlap):")
for name, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:12]:
    v.sort()
    print("  %-86s n=%5d  median %6.2f us" % (name, len(v), v[len(v) // 2] / 1e3))
