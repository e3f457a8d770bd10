"""Sum rocprofv3 PMC csv counters per kernel name prefix (development aid).  usage: pmc_show.py <dir> <kernel substr>"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot):
    print(f"{k:32s} {tot[k]/max(n[k],1):16.0f}  per launch ({n[k]} launches)")
