import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print("%9.2f ms %6.2f%% calls=%6s avg=%9.2fus  %s" % (float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), r["Calls"],
                                                         float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total %.2f ms" % (tot / 1e6))
