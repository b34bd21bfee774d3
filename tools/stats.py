"""Condense a rocprofv3 --stats kernel_stats.csv into per-step milliseconds,
with the two MFMA kernel families aggregated over their instantiations (the
numbers bench.py's `roofline` object must agree with)."""
import csv, glob, re, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/*/*kernel_stats.csv")[0] if not d.endswith('.csv') else d
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("launch set: %s  (divided by %g recorded train() steps: warm-up + timed + the eager instrumented pass)" % (f.split('/')[-1], steps))
print("total kernel ms/step %.2f" % (tot / 1e6 / steps))
# swconv: the tile kernels (swconv_kernel), the software-pipelined tiles
# (swconv_swp_kernel) and the split-K finishing launches ride in one family, as
# bench.py times them; wgrad: wgrad_kernel + wgrad_multi_kernel + the reduce
for fam, pats in (("swconv", ("swconv_kernel", "swconv_swp_kernel")),
                  # (cg_wgrad's kernels; not dense_wgrad_kernel / dense1_wgrad_kernel)
                  ("wgrad", ("::wgrad_",))):
    sel = [r for r in rows if any(p in r["Name"] for p in pats)]
    calls = sum(int(r["Calls"]) for r in sel)
    ns = sum(float(r["TotalDurationNs"]) for r in sel)
    print("FAMILY %-14s %6d calls (%.1f per step) %8.1f us avg %6.2f ms/step %5.1f%%" % (
        fam, calls, calls / steps, ns / calls / 1e3, ns / 1e6 / steps, 100 * ns / tot))
for r in rows[:34]:
    n = r["Name"]
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    n = m.group(1) if m else n[:52]
    print("%-40s %6s calls %8.1f us avg %6.2f ms/step %5.1f%%" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"])))
