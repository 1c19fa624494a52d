"""Post-process rocprofv3 output (run on the GPU box by scripts/profile_round.sh) into profiles/.

Inputs under gpurun_out/<tag>_<scene>/: stats/ (--kernel-trace --stats), fetch/ (--pmc FETCH_SIZE), write/ (--pmc WRITE_SIZE).
Outputs: profiles/<tag>_<scene>_kernel_stats.csv (verbatim rocprofv3 summary), profiles/<tag>_<scene>_pmc_hbm.csv (per-kernel HBM
bytes per launch with the MI355X guide's gfx950 correction: FETCH_SIZE is in KiB and reads of a 16-B-per-lane stream are
reported at half their size), profiles/pmc_<scene>.json (what bench.py reports as roofline.traffic: HBM bytes per launch of
the path-continuation closest-hit kernel and of the shade kernel, at the pass size of the full-spp run).
"""
import collections, csv, glob, json, os, shutil, sys

tag, scene = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'cornell')
base = os.path.join('gpurun_out', f'{tag}_{scene}')
os.makedirs('profiles', exist_ok=True)
st = glob.glob(os.path.join(base, 'stats', '*', '*_kernel_stats.csv'))
if st:
    shutil.copy(max(st, key=os.path.getmtime), os.path.join('profiles', f'{tag}_{scene}_kernel_stats.csv'))  # the latest run

def per_kernel(sub, counter):
    files = glob.glob(os.path.join(base, sub, '*', '*_counter_collection.csv'))
    agg = collections.defaultdict(float); n = collections.Counter()
    if not files:
        return agg, n
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
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
    rows.append((k.split('(')[0].replace('void ', ''), launches, rd, wr, rd + wr, rd / 2.0 + wr))
with open(os.path.join('profiles', f'{tag}_{scene}_pmc_hbm.csv'), 'w') as f:
    f.write('kernel,launches,read_bytes_per_launch(x2_corrected),write_bytes_per_launch,hbm_bytes_per_launch,hbm_bytes_per_launch_raw_reads\n')
    for r in rows:
        f.write(f'"{r[0]}",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.0f},{r[5]:.0f}\n')

def pick(prefixes, exclude=(), per_stage=False):
    sel = [r for r in rows if any(r[0].startswith(p) for p in prefixes) and not any(e in r[0] for e in exclude)]
    if not sel:
        return None
    # several variants of one stage (k_shade<3> + k_shade<0> on a class-split queue, k_trace variants) run once per bounce each:
    # bytes per launch = all their bytes / the launches of the most frequent variant
    tot_l = max(r[1] for r in sel) if per_stage else sum(r[1] for r in sel)
    lanes = None
    if valu:
        tc = sum(valu[k][0] for k in valu if any(k.split('(')[0].replace('void ', '').startswith(p) for p in prefixes) and not any(e in k for e in exclude))
        ni = sum(valu[k][1] for k in valu if any(k.split('(')[0].replace('void ', '').startswith(p) for p in prefixes) and not any(e in k for e in exclude))
        lanes = round(tc / ni, 1) if ni else None
    # the x2 of the guide holds for wide coalesced 16-B-per-lane streams; kernels that gather lie between the raw and the doubled figure: both are kept
    return {'kernels': sorted({r[0] for r in sel}), 'hbm_bytes_per_launch': round(sum(r[4] * r[1] for r in sel) / tot_l),
            'hbm_bytes_per_launch_raw': round(sum(r[5] * r[1] for r in sel) / tot_l), 'launches_profiled': tot_l, 'lanes_per_valu': lanes}

# lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU), when the SQ pass was collected
valu = {}
tcv, _ = per_kernel('sq', 'SQ_THREAD_CYCLES_VALU')
niv, _ = per_kernel('sq', 'SQ_INSTS_VALU')
for k in set(tcv) & set(niv):
    valu[k] = (tcv[k], niv[k])

def source_sha():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join('rustracer_amd', 'csrc')
    host_only = ('rtx_pbrt.inl', 'rtx_images.inl', 'rtx_spectrum_tables.inl')  # (bench.py source_sha: what librtx_hip.so is built from)
    for f in sorted(os.listdir(d)):
        if f.endswith(('.h', '.hip', '.inl')) and f not in host_only:
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]

# closest-hit kernels of the timed frames: k_trace<false, false, ...> (LDS scenes), k_trace_pair<false, ...> or k_trace_top<false, ...> (HBM scenes)
import datetime, subprocess
try:
    commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
out = {'scene': scene, 'collected': f'{tag}, {datetime.date.today().isoformat()}' + (f', tree {commit}' if commit else ''),
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB->B. hbm_bytes_per_launch_raw = writes + reads as counted (64 B per L2-miss read request), hbm_bytes_per_launch = writes + 2 x reads (rounds 1 - 5). Calibrated in round 6 (profiles/r06_fetch_calibration.txt): a coalesced stream is fetched in 128-byte requests (counter = half its bytes), a gather that misses L2 is one request per 64-byte sector (counter = what it moved) - bench.py therefore reports raw + half of the STREAMED read bytes of the stage as `traffic`',
       'source': f'profiles/{tag}_{scene}_pmc_hbm.csv', 'kernel_source_sha': source_sha(),
       'trace_closest': pick(['rtx::k_trace<false, false', 'rtx::k_trace_pair<false', 'rtx::k_trace_top<false']),
       'shade': pick(['rtx::k_shade'], per_stage=True)}

def per_sample(prefixes):
    """All launches of a kernel group (path and MIS rays alike: the counters are per kernel name) as HBM bytes per camera sample of the profiled run."""
    sel = [r for r in rows if any(r[0].startswith(p) for p in prefixes)]
    if not sel or not samples:
        return None
    return {'kernels': sorted({r[0] for r in sel}), 'hbm_bytes_per_camera_sample': sum(r[4] * r[1] for r in sel) / samples,
            'hbm_bytes_per_camera_sample_raw': sum(r[5] * r[1] for r in sel) / samples, 'camera_samples_profiled': samples}

samples = None
try:  # the PMC runs write their full measurement next to their counters (bench.py --detail)
    samples = json.load(open(os.path.join(base, 'fetch_detail.json')))['camera_samples_per_step']
except Exception:
    pass
out['trace_closest_all'] = per_sample(['rtx::k_trace<false, false', 'rtx::k_trace_pair<false', 'rtx::k_trace_top<false'])  # k_trace_big is the visit-counting frame's kernel: not a timed launch
out['trace_any_all'] = per_sample(['rtx::k_trace<true, false', 'rtx::k_trace_pair<true', 'rtx::k_trace_top<true', 'rtx::k_trace_quad'])  # (<ANY, COUNT = false, ...>: the counting frame's kernels are not timed launches)

# Per shade front-end: HBM bytes per VERTEX of the class the form serves ("kernel: k_shade" hid which form wastes the bytes). Vertex counts by
# front-end class come from the PMC run's own detail file; the plain and BOUNCED / LEAN / QLIGHTS forms of one front-end share its count.
verts = {}
try:
    _fd = json.load(open(os.path.join(base, 'fetch_detail.json')))
    # the detail file counts the vertices of ONE frame; a PMC run renders steps + warmup timed frames and the counting frame, all through the same shade kernels
    _frames = int(_fd.get('steps', 1)) + int(_fd.get('warmup', 0)) + 1
    verts = {k: v * _frames for k, v in (_fd.get('vertices_by_shade_front_end') or {}).items()}
except Exception:
    pass
CLASS_OF = {'1': 'lambert_const', '3': 'lambert', '5': 'two_lobe', '6': 'two_lobe', '0': 'generic'}
forms = collections.defaultdict(lambda: [0.0, 0.0, 0, []])
for r in rows:
    if r[0].startswith('rtx::k_shade<'):
        cls = CLASS_OF.get(r[0][len('rtx::k_shade<')], None)
        if cls:
            f = forms[cls]; f[0] += r[4] * r[1]; f[1] += r[5] * r[1]; f[2] += r[1]; f[3].append(r[0])
out['shade_by_front_end'] = {cls: {'kernels': sorted(f[3]), 'launches': f[2], 'hbm_bytes': round(f[0]), 'hbm_bytes_raw_reads': round(f[1]), 'vertices': verts.get(cls),
                                   'hbm_bytes_per_vertex': round(f[0] / verts[cls], 1) if verts.get(cls) else None,
                                   'hbm_bytes_per_vertex_raw_reads': round(f[1] / verts[cls], 1) if verts.get(cls) else None} for cls, f in forms.items()}
# How busy the SIMDs' vector ALUs are in the two heavy stages: SQ_INSTS_VALU (wave instructions, 4 cycles each on a 16-lane SIMD) of the SQ run / (1024 SIMDs x the kernels' time x
# sclk); the time of the SQ run's dispatches is estimated from the traced run's average per dispatch (same kernels, same launch sizes), sclk from the detail file's reading after the frames
def valu_busy(prefixes):
    try:
        sclk = (_fd.get('gpu_clocks') or {}).get('sclk_MHz', {}).get('after') or 2300
        stats_files = glob.glob(os.path.join(base, 'stats', '*', '*_kernel_stats.csv'))
        avg = {}
        for r in csv.DictReader(open(max(stats_files, key=os.path.getmtime))):
            avg[r['Name'].split('(')[0].replace('void ', '')] = float(r['AverageNs'])
        insts = cyc = 0.0
        for k, (tc, ni) in valu.items():
            kk = k.split('(')[0].replace('void ', '')
            if any(kk.startswith(p_) for p_ in prefixes) and kk in avg:
                nd = [n_ for (kn, c), n_ in sq_disp.items() if kn == k and c == 'SQ_INSTS_VALU']
                insts += ni; cyc += avg[kk] * 1e-9 * (nd[0] if nd else 0) * sclk * 1e6 * 1024.0
        return round(insts * 4.0 / cyc, 3) if cyc else None
    except Exception:
        return None
_, sq_disp_n = per_kernel('sq', 'SQ_INSTS_VALU')
sq_disp = {(k, 'SQ_INSTS_VALU'): v for k, v in sq_disp_n.items()}
for grp, pre in (('trace_closest', ['rtx::k_trace<false, false', 'rtx::k_trace_pair<false', 'rtx::k_trace_top<false']), ('shade', ['rtx::k_shade'])):
    if out.get(grp):
        out[grp]['valu_busy'] = valu_busy(pre)
json.dump(out, open(os.path.join('profiles', f'pmc_{scene}.json'), 'w'), indent=1)

# The SQ pass, summarised per kernel next to the traffic files and under the same kernel_source_sha (the r04 SQ summaries predated the final kernels)
sq_files = glob.glob(os.path.join(base, 'sq', '*', '*_counter_collection.csv'))
if sq_files:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
    for r in csv.DictReader(open(max(sq_files, key=os.path.getmtime))):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith('__amd'):
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])] += 1
    unit_of = {'rtx::k_shade<1': ('lambert_const', 'vertex'), 'rtx::k_shade<3': ('lambert', 'vertex'), 'rtx::k_shade<5': ('two_lobe', 'vertex'), 'rtx::k_shade<6': ('two_lobe', 'vertex'),
               'rtx::k_shade<0': ('generic', 'vertex')}
    with open(os.path.join('profiles', f'{tag}_{scene}_pmc_sq.txt'), 'w') as f:
        f.write(f'# rocprofv3 --pmc SQ_* (one pass, scripts/profile_round.sh {tag} {scene}); kernel_source_sha {source_sha()}; sums over all dispatches of the run\n')
        f.write('# lanes = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (active lanes per VALU instruction, of 64); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (x waves per SIMD = share of cycles a SIMD issues);\n')
        f.write('# valu_per_unit = SQ_INSTS_VALU (wave instructions) x 64 / units = VALU lane-slots per vertex (shade kernels; units from the run\'s vertices_by_shade_front_end)\n')
        for k, v in sorted(agg.items()):
            iv, tc, wc = v.get('SQ_INSTS_VALU', 0.0), v.get('SQ_THREAD_CYCLES_VALU', 0.0), v.get('SQ_WAVE_CYCLES', 0.0)
            extra = ''
            for pre, (cls, unit) in unit_of.items():
                if k.startswith(pre) and verts.get(cls):
                    extra = f' valu_per_{unit}={iv * 64.0 / verts[cls]:.0f}'
            f.write(f"{k} dispatches={max(disp[(k, c)] for c in v)} lanes={tc / iv if iv else 0:.1f} wait={v.get('SQ_WAIT_ANY', 0.0) / wc if wc else 0:.3f} "
                    f"issue={v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc if wc else 0:.3f}{extra} " + ' '.join(f'{a}={b:.4g}' for a, b in sorted(v.items())) + '\n')

print(open(os.path.join('profiles', f'{tag}_{scene}_pmc_hbm.csv')).read())
