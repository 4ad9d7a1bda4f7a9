"""Print the per-kernel rows of a rocprofv3 --kernel-trace --stats csv directory."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e6:9.4f} ms  {r['Percentage']:>6s} %")
