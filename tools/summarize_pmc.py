"""Average a rocprofv3 --pmc counter per kernel from *_counter_collection.csv (values are per dispatch)."""
import csv, sys, collections, re
path, = sys.argv[1:2]
acc = collections.defaultdict(lambda: [0.0, 0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        k = (name, r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("kernel,counter,dispatches,avg_value")
for (n, c), (s, k) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{n},{c},{k},{s / k:.3f}")
