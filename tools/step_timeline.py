"""Timeline of one training step from a rocprofv3 kernel trace (csv): span, forward / backward / optimizer phases, time with one
or two kernels on the chip, per-kernel totals.  usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
step = rows[a + 1:b + 1]
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
t0 = S(step[0]); t1 = max(E(r) for r in step)
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:44]
print("step: %d kernels, span %.3f ms, sum of kernel time %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(E(r) - S(r) for r in step) / 1e6))
def first(key): return next((r for r in step if key in r["Kernel_Name"]), None)
mse, ssq, adam = first("masked_mse"), first("sumsq"), first("clip_adam")
if mse and ssq and adam:
    print("  forward (to the loss) %.3f ms | backward (to the norm) %.3f ms | norm + clip + Adam %.3f ms" %
          ((S(mse) - t0) / 1e6, (S(ssq) - S(mse)) / 1e6, (E(adam) - S(ssq)) / 1e6))
ev = sorted([(S(r), 1) for r in step] + [(E(r), -1) for r in step])
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last; last = t; depth += d
print("  chip idle %.3f ms, one kernel %.3f ms, two %.3f ms, three or more %.3f ms" % tuple(hist[k] / 1e6 for k in range(4)))
# what ran alone vs beside another kernel, per name
tot = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    tot[short(r["Kernel_Name"])][0] += 1; tot[short(r["Kernel_Name"])][1] += (E(r) - S(r)) / 1e3
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:32]:
    print("  %-46s x%3d %8.1f us  (avg %5.1f)" % (k, n, us, us / n))
