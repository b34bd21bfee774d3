import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", tot / 1e6 / steps)
for r in rows[:34]:
    n = r["Name"]
    n = n.split("::")[-1] if "::" in n else n
    n = n.replace("((anonymous namespace)", "(")[:52]
    print("%-52s %5s %8.1f us  %6.2f ms/step %5.1f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"])))
