"""Post-process rocprofv3 output (run on the GPU box by scripts/profile_round.sh) into profiles/.

Inputs under gpurun_out/<tag>/: stats/ (--kernel-trace --stats), fetch/ (--pmc FETCH_SIZE), write/ (--pmc WRITE_SIZE).
Outputs: profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 summary), profiles/<tag>_pmc_hbm.csv (per-kernel HBM
bytes per launch with the MI355X guide's gfx950 correction: FETCH_SIZE is in KiB and reads of a 16-B-per-lane
stream are reported at half their size), profiles/pmc_trace_closest.json (what bench.py reports as roofline.traffic).
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
base = os.path.join('gpurun_out', tag)
os.makedirs('profiles', exist_ok=True)
st = glob.glob(os.path.join(base, 'stats', '*', '*_kernel_stats.csv'))
if st:
    shutil.copy(st[0], os.path.join('profiles', f'{tag}_kernel_stats.csv'))

def per_kernel(sub, counter):
    files = glob.glob(os.path.join(base, sub, '*', '*_counter_collection.csv'))
    agg = collections.defaultdict(float); n = collections.Counter()
    if not files:
        return agg, n
    for r in csv.DictReader(open(files[0])):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        agg[k] += float(r['Counter_Value']); n[k] += 1
    return agg, n

fetch, nf = per_kernel('fetch', 'FETCH_SIZE')
write, nw = per_kernel('write', 'WRITE_SIZE')
rows = []
for k in sorted(set(fetch) | set(write)):
    if k.startswith('__amd'):
        continue
    launches = max(nf.get(k, 0), nw.get(k, 0), 1)
    rd = fetch.get(k, 0.0) * 1024.0 * 2.0 / launches   # KiB -> B, x2: gfx950 reports wide streaming reads at half size
    wr = write.get(k, 0.0) * 1024.0 / launches
    rows.append((k.split('(')[0].replace('void ', ''), launches, rd, wr, rd + wr))
with open(os.path.join('profiles', f'{tag}_pmc_hbm.csv'), 'w') as f:
    f.write('kernel,launches,read_bytes_per_launch(x2_corrected),write_bytes_per_launch,hbm_bytes_per_launch\n')
    for r in rows:
        f.write(f'"{r[0]}",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.0f}\n')
tc = [r for r in rows if r[0].startswith('rtx::k_trace<false')]
if tc:
    tot_l = sum(r[1] for r in tc)
    val = sum(r[4] * r[1] for r in tc) / tot_l
    json.dump({'kernel': 'k_trace<closest>', 'hbm_bytes_per_launch': round(val), 'launches_profiled': tot_l,
               'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB->B; reads x2 (gfx950 16-B/lane correction, MI355X guide)',
               'source': f'profiles/{tag}_pmc_hbm.csv'}, open(os.path.join('profiles', 'pmc_trace_closest.json'), 'w'), indent=1)
print(open(os.path.join('profiles', f'{tag}_pmc_hbm.csv')).read())
