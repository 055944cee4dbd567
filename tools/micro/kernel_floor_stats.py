"""Per-kernel averages of tools/micro/kernel_floor.hip under rocprofv3 (usage on the GPU box: see tools/micro/kernel_floor.sh)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:50]:50s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:7.2f} min_us {float(r['MinNs']) / 1e3:7.2f}")
