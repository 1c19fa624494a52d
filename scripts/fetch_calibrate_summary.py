"""Joins the known byte counts of scripts/micro/fetch_calibrate (known.json, in launch order) with the per-dispatch counter rows of its --pmc passes; writes
<dir>/calibration.json = per pattern: what FETCH_SIZE / WRITE_SIZE reported (KiB -> bytes) against the bytes requested, the distinct 64-byte sectors and 128-byte lines."""
import csv, glob, json, sys, collections
d = sys.argv[1]
known = json.load(open(d + '/known.json'))['runs']
per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [per dispatch, in order]
for f in sorted(glob.glob(d + '/p*/*/*_counter_collection.csv')):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        per[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = []
seen = collections.Counter()
for run in known:
    k = run['kernel']
    # warm-up launches of the same kernel names (flushes, table warmers) are NOT in known.json: pick dispatches by the order the program reports them
    out.append(dict(run))
    out[-1]['counters'] = {}
# order of dispatches per kernel name, as the program issues them (see main()): cal_stream<4> = report, then 2 flushes before each of 6 gather groups; cal_gather<1, 150> = 3 reported + 4 warmers
def pick(kernel, state):
    if kernel == 'cal_stream<4>': return 0
    if kernel.startswith('cal_gather'): return {'cold': 0, 'warm': 1, 'warm2': 2}[state]
    if kernel.startswith('cal_mix'): return {'rep0': 0, 'rep1': 1}[state]
    return 0
for o in out:
    i = pick(o['kernel'], o['state'])
    for c, vals in per.get(o['kernel'], {}).items():
        if i < len(vals): o['counters'][c] = vals[i]
    cs = o['counters']
    if 'FETCH_SIZE' in cs:
        o['fetch_bytes'] = cs['FETCH_SIZE'] * 1024.0
        for key in ('bytes_requested', 'bytes_sectors64', 'bytes_lines128'):
            if o[key] > 0: o['fetch_over_' + key[6:]] = round(o['fetch_bytes'] / o[key], 4)
    if 'WRITE_SIZE' in cs and o['bytes_stored'] > 0:
        o['write_bytes'] = cs['WRITE_SIZE'] * 1024.0; o['write_over_stored'] = round(o['write_bytes'] / o['bytes_stored'], 4)
    if 'TCC_HIT_sum' in cs: o['l2_hit_rate'] = round(cs['TCC_HIT_sum'] / max(cs['TCC_HIT_sum'] + cs.get('TCC_MISS_sum', 0.0), 1.0), 4)
json.dump({'runs': out}, open(d + '/calibration.json', 'w'), indent=1)
for o in out:
    print(f"{o['kernel']:24s} {o['state']:6s} {o['ms']:9.3f} ms  req {o['bytes_requested']/1e6:9.1f} MB  sect64 {o['bytes_sectors64']/1e6:9.1f}  FETCH {o.get('fetch_bytes', float('nan'))/1e6:9.1f} MB"
          f"  /req {o.get('fetch_over_requested', '-')}  /sect64 {o.get('fetch_over_sectors64', '-')}  /line128 {o.get('fetch_over_lines128', '-')}  WRITE/stored {o.get('write_over_stored', '-')}  L2 hit {o.get('l2_hit_rate', '-')}"
          f"  rdreq {o['counters'].get('TCC_EA0_RDREQ_sum', '-')} rdreq32 {o['counters'].get('TCC_EA0_RDREQ_32B_sum', '-')}")
