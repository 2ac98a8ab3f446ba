#!/usr/bin/env python3
"""Per-function register / scratch / LDS figures of the gfx950 code object, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(kernels and the non-inlined device functions they call).  Usage: tools/kernel_resources.py [extra hipcc flags] > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ['hipcc', '--offload-arch=gfx950', '-O2', '-mllvm', '-sink-insts-to-avoid-spills', '-mllvm', '-inline-threshold=1000', '-mllvm', '-disable-machine-licm', '-mllvm', '-phi-node-folding-threshold=1', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Wno-unused-variable', '-Rpass-analysis=kernel-resource-usage',
       '-o', '/tmp/libmi_resources_probe.so', os.path.join(root, 'cavif_rs_amd', 'csrc', 'mi_avif.hip'), '-lz'] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in err.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass-analysis', ln)
    if not m:
        continue
    k, _, v = m.group(1).strip().partition(':')
    if k == 'Function Name':
        cur = {'name': v.strip()}; rows.append(cur)
    elif cur is not None:
        cur[k.strip()] = v.strip()
def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n
print('%-110s %5s %5s %8s %8s %8s %4s %7s' % ('function', 'VGPR', 'SGPR', 'scratchB', 'vgprSpill', 'sgprSpill', 'occ', 'LDS'))
for r in rows:
    name = re.sub(r'\b(mi::|\(anonymous namespace\)::)', '', demangle(r['name']))
    name = re.sub(r'\(.*', '', name)
    print('%-110s %5s %5s %8s %8s %8s %4s %7s' % (name[:110], r.get('VGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('VGPRs Spill'), r.get('SGPRs Spill'),
                                              r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
