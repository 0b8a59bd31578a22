"""Average counter value per launch and kernel from rocprofv3 --pmc csv files (one or more passes).
usage: pmc_table.py <title> <kernel-name substring> <counter_collection.csv> [more csv ...]"""
import csv, sys
from collections import defaultdict
title, sub = sys.argv[1], sys.argv[2]
vals = defaultdict(list)
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = max((len(v) for v in vals.values()), default=0)
print("%s (%d launches; rocprofv3 --pmc, %d pass(es)):" % (title, n, len(sys.argv) - 3))
for k in sorted(vals):
    print("  %-32s %.4g" % (k, sum(vals[k]) / len(vals[k])))
