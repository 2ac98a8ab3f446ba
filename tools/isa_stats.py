#!/usr/bin/env python3
"""Static instruction counts per device function of the gfx950 ISA (hipcc -S --cuda-device-only): total / VALU / SALU / LDS / memory.
Usage: tools/isa_stats.py file.s [top N]   (the first thing to look at when a kernel slows down for no visible reason)"""
import re, subprocess, sys
fn = None; counts = {}
for ln in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):\s*(;.*)?$', ln)
    if m:
        fn = m.group(1); counts[fn] = dict(v=0, s=0, ds=0, mem=0, tot=0, pk=0); continue
    if fn is None:
        continue
    t = ln.strip()
    if not t or t[0] in ';.':
        if t.startswith('.Lfunc_end'):
            fn = None
        continue
    op = t.split()[0]; c = counts[fn]
    if op.startswith('v_'):
        c['v'] += 1
        if op.startswith(('v_pk_', 'v_dot', 'v_sad')): c['pk'] += 1
    elif op.startswith('s_'): c['s'] += 1
    elif op.startswith('ds_'): c['ds'] += 1
    elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): c['mem'] += 1
    c['tot'] += 1
names = list(counts)
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print('%7s %7s %6s %6s %5s %5s  function' % ('total', 'VALU', 'SALU', 'LDS', 'mem', 'pk'))
for n, d in sorted(zip(names, dem), key=lambda x: -counts[x[0]]['tot'])[:top]:
    c = counts[n]
    d = re.sub(r'\b(mi::|\(anonymous namespace\)::)', '', d)
    print('%7d %7d %6d %6d %5d %5d  %s' % (c['tot'], c['v'], c['s'], c['ds'], c['mem'], c['pk'], d[:170]))
