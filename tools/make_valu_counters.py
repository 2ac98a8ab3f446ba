#!/usr/bin/env python3
"""profiles/valu_counters.json from a rocprofv3 PMC summary (tools/pmc_summary.py over the SQ_* pass of tools/final_profile.sh) and the kernel-trace
statistics of the same command: what bench.py reads for `roofline_valu` -- the figure that actually bounds the tile search (VALU issue), next to the HBM
roofline north_star asks for.  Usage: tools/make_valu_counters.py gpurun_out/TAG_pmc_summary.json gpurun_out/TAG_kernel_stats_pipeline1.txt TAG > profiles/valu_counters.json"""
import json, re, sys
src, stats, tag = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(src))
avg_ms = {}
for ln in open(stats):                             # tools/rocpd_summary.py rows: name (70 columns), calls, total_ms, avg_ms, ...
    m = re.match(r'(.{70})\s+(\d+)\s+([\d.]+)\s+([\d.]+)', ln)
    if m:
        short = re.split(r'[<(]', re.sub(r'^(void |mi::)+', '', m.group(1).strip()))[0].strip()
        avg_ms[short] = max(avg_ms.get(short, 0.0), float(m.group(4)))
kern = {}
for name, c in d.items():
    if 'SQ_ACTIVE_INST_VALU' not in c:
        continue
    short = re.split(r'[<(]', re.sub(r'^(void |mi::)+', '', name))[0].strip()
    if short in kern and kern[short]['SQ_ACTIVE_INST_VALU'] >= c['SQ_ACTIVE_INST_VALU']:
        continue                                   # instantiations of one kernel: the one that did the work
    kern[short] = {k: c[k] for k in ('SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES',
                                         'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY') if k in c}
    if short in avg_ms:
        kern[short]['avg_launch_ms'] = avg_ms[short]
out = {'config': {'images_per_gpu': 32, 'width': 1920, 'height': 1080, 'speed': 4, 'quality': 80.0, 'bit_depth': 10},
       'source': 'rocprofv3 --pmc SQ_* (own pass, --kernel-trace only) around `python bench.py --steps 1 --warmup 1 --pipeline 1 --no-pcie-loop`; mean per launch; avg_launch_ms from the kernel trace of the same command; tools/final_profile.sh ' + tag,
       'units': 'SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_*: quad-cycles (4 clocks) summed over the wavefronts; SQ_THREAD_CYCLES_VALU: lane x quad-cycles; SQ_INSTS_*: wave-instructions',
       'kernels': kern}
print(json.dumps(out, indent=1, sort_keys=True))
