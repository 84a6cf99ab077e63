"""Print the kernels and copies of the LAST set_pair of tools/pair_prof.py in time order, with gaps (us).
usage: pair_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv, glob, os, re, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"[<(].*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:48], "q%s" % r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
ev.sort()
# the last alignment = after the last-but-50th k_nn_accum launch block: find the start of the last run of build kernels
nn = [i for i, e in enumerate(ev) if "k_nn_accum" in e[2]]
last_loop_start = nn[-50] if len(nn) >= 50 else nn[0]
prev_loop_end = nn[-51] if len(nn) >= 51 else 0
seg = ev[prev_loop_end + 2:last_loop_start + 1]
t0 = seg[0][0]
prev_end = t0
for s, e, name, q in seg:
    print("%9.1f  +%7.1f us  gap %7.1f  %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name, q))
    prev_end = max(prev_end, e)
print("total %.1f us" % ((seg[-1][0] - t0) / 1e3))
