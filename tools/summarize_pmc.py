#!/usr/bin/env python
"""Summarise gpurun_out/pmc_<tag>_{1..4} (tools/gpu_pmc.sh) into profiles/:
  profiles/<round>_pmc_summary.csv   per kernel symbol x grid: mean counters
  profiles/pmc_traffic.json          HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KB -> B)
FETCH_SIZE on gfx950 under-counts wide (16 B/lane) streaming reads by 2x
(MI355X_MICROARCH.md section HBM); the conv staging mixes 4 B/lane input loads with
16 B/lane weight loads, so the raw value is reported and flagged `uncorrected`."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01b'
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r01'
outdir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'profiles')     # the GPU box writes to gpurun_out/
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3, 4, 5, 6, 7, 8):     # 5-6: the 2xBI config, 7-8: the training step (tools/gpu_pmc.sh)
    path = os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}_{i}', 'pmc_counter_collection.csv')
    if not os.path.isfile(path):
        continue
    for r in csv.DictReader(open(path)):
        if 'tg::' not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Grid_Size'] + ('' if i <= 4 else (' [2xBI]' if i <= 6 else ' [train128]')))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for d in agg.values() for c in d})
out = os.path.join(outdir, f'{rnd}_pmc_summary.csv')
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid', 'dispatches'] + names)
    for (k, g), d in sorted(agg.items()):
        n = max(len(v) for v in d.values())
        w.writerow([k, g, n] + [f'{sum(d[c]) / len(d[c]):.1f}' if c in d else '' for c in names])
traffic = {}
for (k, g), d in agg.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        # weight by dispatch count: per-symbol mean over all launches of a frame; the launches of
        # the other workloads (2xBI clip, training step) are kept apart under a tagged key
        tagk = k + (' ' + g[g.index('['):] if '[' in g else '')
        t = traffic.setdefault(tagk, [0.0, 0])
        n = len(d['FETCH_SIZE'])
        t[0] += (sum(d['FETCH_SIZE']) / n + sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])) * 1024 * n
        t[1] += n
table = {k: v[0] / v[1] for k, v in traffic.items()}
# stamp: sha1 of the kernel sources these counters were collected on (bench.py prints traffic_stale on mismatch)
sys.path.insert(0, ROOT)
import hashlib
csrc = os.path.join(ROOT, 'tecogan-pytorch_amd', 'csrc')
table['_sources'] = {f: hashlib.sha1(open(os.path.join(csrc, f), 'rb').read()).hexdigest()[:16]
                     for f in sorted(os.listdir(csrc)) if f.endswith(('.hip', '.h'))}
json.dump(table, open(os.path.join(outdir, 'pmc_traffic.json'), 'w'), indent=1)
print('wrote', out, 'and profiles/pmc_traffic.json')
