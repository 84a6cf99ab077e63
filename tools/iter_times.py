"""Per-launch durations of k_nn_accum / k_reduce_solve in launch order from a rocprofv3 kernel trace CSV."""
import csv, glob, os, sys
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    nn = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_nn_accum" in r["Kernel_Name"]]
    sv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_reduce_solve" in r["Kernel_Name"]]
    gaps = []
    kk = [r for r in rows if "k_nn_accum" in r["Kernel_Name"] or "k_reduce_solve" in r["Kernel_Name"]]
    for a, b in zip(kk, kk[1:]):
        gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    last = nn[-50:]
    print("nn last-run per-iteration us:", " ".join(f"{x:.0f}" for x in last))
    # k_nn_accum<COUNT, MODE, LISTS, HALO>: search / search + write lists / list launch
    kind = lambda nm: "L" if ", 0, 2," in nm or ", 0, 2>" in nm else ("W" if ", 0, 1," in nm or ", 0, 1>" in nm else "S")
    kinds = [kind(r["Kernel_Name"]) for r in rows if "k_nn_accum" in r["Kernel_Name"]][-50:]
    print("launch kinds (S search, W search + write lists, L list launch):", "".join(kinds))
    for kd in "SWL":
        v = [x for x, y in zip(last, kinds) if y == kd]
        if v:
            print(f"  {kd}: {len(v)} launches, mean {sum(v) / len(v):.1f} us")
    print("nn mean %.1f  solve mean %.1f (n=%d)  median gap %.2f us" % (sum(last) / len(last), sum(sv[-50:]) / 50, len(sv), sorted(gaps[-99:])[49]))
