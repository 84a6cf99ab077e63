"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a small text table."""
import csv, glob, os, sys
d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
lines = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    lines.append(f"# {os.path.relpath(f, d)}")
    for i, row in enumerate(csv.reader(open(f))):
        lines.append(" | ".join(c[:70] for c in row))
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        k = r["Kernel_Name"][:60]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        by.setdefault(k, []).append((dur, r))
    lines.append(f"# {os.path.relpath(f, d)}: per-kernel launches / avg us / min / max / VGPR / LDS / grid / wg")
    for k, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        durs = [x[0] for x in v]
        r = v[0][1]
        lines.append(f"{k:60s} n={len(durs):5d} avg={sum(durs)/len(durs):9.2f} min={min(durs):9.2f} max={max(durs):9.2f} "
                     f"vgpr={r.get('VGPR_Count','?')} sgpr={r.get('SGPR_Count','?')} lds={r.get('LDS_Block_Size','?')} grid={r.get('Grid_Size_X', r.get('Grid_Size','?'))} wg={r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))}")
txt = "\n".join(lines)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
