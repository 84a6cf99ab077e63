"""Kernel overlap in a rocprofv3 --kernel-trace directory: how much of the traced interval has 0, 1, 2, ... k_nn_accum
launches in flight at once, and the busy fraction (union of kernel intervals / span)."""
import csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
nn = [(a, b) for a, b, n in rows if "k_nn_accum" in n]
nn.sort()
# the timed phase of tools/concurrent_pairs.py is the last K x 20 x 50 launches (give that number as the 2nd argument)
if len(sys.argv) > 2:
    nn = nn[-int(sys.argv[2]):]
t0, t1 = nn[0][0], max(b for _, b in nn)
ev = sorted([(a, 1) for a, _ in nn] + [(b, -1) for _, b in nn])
hist, cur, last = {}, 0, t0
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d; last = t
span = t1 - t0
print("k_nn_accum launches analysed: %d over %.2f ms" % (len(nn), span / 1e6))
for k in sorted(hist):
    print("  %d in flight: %5.1f %% of the time" % (k, 100.0 * hist[k] / span))
print("  sum of launch durations / span = %.2f (average launches in flight)" % (sum(b - a for a, b in nn) / span))
