#!/usr/bin/env python3
"""profiles/hbm_counters.json from a rocprofv3 PMC summary (tools/pmc_summary.py output of the FETCH_SIZE / WRITE_SIZE passes of tools/final_profile.sh):
what bench.py reads for `roofline.traffic`.  Usage: tools/make_hbm_counters.py gpurun_out/TAG_pmc_summary.json TAG "note" > profiles/hbm_counters.json"""
import json, re, sys
src, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ''
d = json.load(open(src))
kern = {}
for name, c in d.items():
    if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
        continue
    short = re.sub(r'^void ', '', name)
    short = re.sub(r'^mi::', '', short)
    short = re.split(r'[<(]', short)[0].strip()
    k = kern.setdefault(short, {'FETCH_SIZE_KB': 0.0, 'WRITE_SIZE_KB': 0.0})
    k['FETCH_SIZE_KB'] = max(k['FETCH_SIZE_KB'], c['FETCH_SIZE'])          # instantiations of one kernel: the one that did the work
    k['WRITE_SIZE_KB'] = max(k['WRITE_SIZE_KB'], c['WRITE_SIZE'])
out = {'config': {'images_per_gpu': 32, 'width': 1920, 'height': 1080, 'speed': 4, 'quality': 80.0, 'bit_depth': 10},
       'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around `python bench.py --steps 1 --warmup 1 --pipeline 1 --no-pcie-loop`; mean per launch; tools/final_profile.sh %s%s' % (tag, ('; ' + note) if note else ''),
       'units': "KB as reported by rocprofv3; bench.py reports traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM')",
       'kernels': kern}
print(json.dumps(out, indent=1, sort_keys=True))
