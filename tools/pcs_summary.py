#!/usr/bin/env python3
"""Summarise a rocprofv3 PC-sampling run (csv): samples per kernel, and for the tile search per source line / per function.
Usage: tools/pcs_summary.py DIR [kernel substring]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else 'tile_search_kernel'
csv.field_size_limit(1 << 30)
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
disp = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        disp[r.get('Dispatch_Id')] = r.get('Kernel_Name', '')
files = [f for f in glob.glob(os.path.join(d, '**', '*pc_sampling*.csv'), recursive=True)]
print('files:', [os.path.basename(f) for f in files])
per_kernel = collections.Counter(); by_line = collections.Counter(); by_inst = collections.Counter(); by_op = collections.Counter()
extra = collections.defaultdict(collections.Counter)
n = 0; cols = None
for f in files:
    rd = csv.DictReader(open(f))
    cols = rd.fieldnames
    for r in rd:
        n += 1
        kn = disp.get(r.get('Dispatch_Id'), '?')
        per_kernel[re.sub(r'\(.*', '', kn)[:90]] += 1
        if want not in kn:
            continue
        ins = r.get('Instruction', ''); cm = r.get('Instruction_Comment', '')
        by_line[cm] += 1; by_inst[(cm, ins)] += 1; by_op[ins.split(' ')[0]] += 1
        for k in ('Wave_Issued_Instruction', 'Instruction_Type', 'Stall_Reason', 'Wave_Count'):
            if k in r: extra[k][r[k]] += 1
print('columns:', cols); print('samples:', n)
for k, v in per_kernel.most_common(12): print('%9d %5.1f%%  %s' % (v, 100.0 * v / max(n, 1), k))
tot = sum(by_line.values())
print('\n== %s: %d samples; by source line' % (want, tot))
for k, v in by_line.most_common(150): print('%8d %5.2f%%  %s' % (v, 100.0 * v / max(tot, 1), k))
print('\n== by opcode')
for k, v in by_op.most_common(40): print('%8d %5.2f%%  %s' % (v, 100.0 * v / max(tot, 1), k))
for k, c in extra.items():
    print('\n== ' + k)
    for kk, v in c.most_common(20): print('%8d %5.2f%%  %s' % (v, 100.0 * v / max(tot, 1), kk))
print('\n== by file')
byf = collections.Counter()
for k, v in by_line.items(): byf[k.rsplit(':', 1)[0].split('/')[-1]] += v
for k, v in byf.most_common(20): print('%8d %5.2f%%  %s' % (v, 100.0 * v / max(tot, 1), k))
