import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "blend_step" in n or "tile_order" in n or "tile_sort" in n:
        print("   %-40s calls %s avg %.1f us" % (n.split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
