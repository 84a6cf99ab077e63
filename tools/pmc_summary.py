"""Average per-dispatch PMC values per kernel from rocprofv3 --pmc CSV output (all passes under a dir)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # keep dispatch order; k_nn_accum: only the LAST 50 dispatches (the measured icp run)
    per_disp = collections.OrderedDict()
    for r in rows:
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        per_disp.setdefault(key, {})[r["Counter_Name"]] = per_disp.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    by_kernel = collections.defaultdict(list)
    for (did, k), vals in per_disp.items():
        by_kernel[k].append(vals)
    for k, lst in by_kernel.items():
        short = None
        if "k_nn_accum<" in k:
            short = k[k.index("k_nn_accum<"):k.index(">", k.index("k_nn_accum<")) + 1]      # one entry per instantiation (search / write lists / read lists)
        elif "k_reduce19" in k or "k_solve" in k:
            short = "k_reduce19" if "k_reduce19" in k else "k_solve"
        if not short:
            continue
        lst = lst[-50:]
        for vals in lst:
            for c, v in vals.items():
                agg[short][c].append(v)
lines = []
for k in agg:
    lines.append(f"== {k} (avg per dispatch over the last {max(len(v) for v in agg[k].values())} dispatches)")
    for c in sorted(agg[k]):
        v = agg[k][c]
        early, late = v[:10], v[-10:]
        lines.append(f"  {c:45s} {sum(v)/len(v):18.1f}   (min {min(v):.0f} max {max(v):.0f}; first 10: {sum(early)/len(early):.0f}, last 10: {sum(late)/len(late):.0f})")
txt = "\n".join(lines)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
