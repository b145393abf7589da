import csv, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    import re
    if re.search(pat, r["Name"]):
        print("   %-48s calls=%5s avg_us=%10.1f" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3))
