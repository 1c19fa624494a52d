"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches)."""
import csv, collections, glob, sys
path = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
for k, v in sorted(agg.items()):
    if k.startswith('__amd'): continue
    print(k, 'dispatches', max(n[(k, c)] for c in v), {a: f"{b:.4g}" for a, b in sorted(v.items())})
