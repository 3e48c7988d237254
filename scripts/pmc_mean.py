"""Mean counter values per launch and kernel from rocprofv3 --pmc CSVs (one or more passes):
    python scripts/pmc_mean.py out.csv pass1_counter_collection.csv [pass2_counter_collection.csv ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:80]


acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for d in acc.values() for c in d})
with open(sys.argv[1], "w") as f:
    f.write("kernel,launches," + ",".join(counters) + "\n")
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        f.write('"%s",%d,%s\n' % (k, n, ",".join("%d" % round(sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else "" for c in counters)))
print(open(sys.argv[1]).read())
