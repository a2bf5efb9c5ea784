"""rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter) -> one row per kernel and counter: dispatches, mean value per
dispatch.  The raw file of a 12-block encoder run has ~10^4 rows; this is what profiles/ keeps.  Usage: summarize_pmc.py FILE > out.csv"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::|amuse::|void ", "", r["Kernel_Name"]).split("(")[0]
    a = acc[(name, r["Counter_Name"])]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for (k, c), (n, s) in sorted(acc.items()):
    w.writerow([k, c, n, f"{s / n:.6g}"])
