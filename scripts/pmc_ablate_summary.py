"""One line of scripts/pmc_ablate.sh: FETCH_SIZE of the shade kernels (bytes = KiB x 1024; 64 per L2-miss read request) per vertex of their front-end, for one RTX_DBG setting.
The measurement process renders the frame twice (exp_ablate.py child): counters are summed over both, vertices are per frame."""
import csv, glob, re, sys, collections
d, log, dbg = sys.argv[1:4]
path = glob.glob(d + '/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != 'FETCH_SIZE': continue
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rtx::', '')
    agg[k] += float(r['Counter_Value']) * 1024.0
txt = open(log).read()
m = re.search(r"vertices \[(\d+), (\d+), (\d+), (\d+)\]", txt)
verts = dict(zip(("lambert_const", "lambert", "two_lobe", "generic"), map(int, m.groups()))) if m else {}
fe = collections.defaultdict(float)
for k, v in agg.items():
    if not k.startswith('k_shade<'): continue
    mode = k[len('k_shade<')]
    fe[{'1': 'lambert_const', '3': 'lambert', '5': 'two_lobe', '6': 'two_lobe', '0': 'generic'}[mode]] += v
frames = 2.0
out = [f"dbg={dbg:>4s}"]
for name in ("lambert", "two_lobe", "generic", "lambert_const"):
    if verts.get(name):
        b = fe[name] / frames / verts[name]
        out.append(f"{name} {b:7.1f} B = {b / 64.0:5.2f} requests / vertex")
tr = sum(v for k, v in agg.items() if k.startswith('k_trace')) / frames
out.append(f"| trace kernels {tr / 1e9:7.2f} GB, k_resolve {sum(v for k, v in agg.items() if k.startswith('k_resolve')) / frames / 1e9:6.2f} GB")
print("  ".join(out), flush=True)
