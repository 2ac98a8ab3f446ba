#!/usr/bin/env python3
"""Emit the AV1 constant tables (default CDFs in a flat per-tile layout, quantizer lookups,
scan orders) as a C header, once for the oracle and once for the HIP product.

Input: tools/av1_default_cdfs.json (made by tools/extract_av1_tables.py) plus the handful of
small spec tables written out below.  CDFs are stored in *inverse* form (32768 - cumulative),
N entries (last one 0) followed by the adaptation counter, i.e. N+1 uint16 per CDF.
"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = json.load(open(os.path.join(ROOT, 'tools', 'av1_default_cdfs.json')))

# small CDFs (AV1 spec "Default CDF tables"), cumulative form
SMALL = {
    'skip': [[31671], [16515], [4576]],
    'cfl_sign': [[1418, 2123, 13340, 18405, 26972, 28343, 32294]],
    'switchable_restore': [[9413, 22581]],
    'use_wiener': [[11570]],
    'use_sgrproj': [[16855]],
    # Default_Segment_Id_Cdf (spatial prediction contexts 0..2, 8 symbols); found verbatim in libaom's .rodata (tools/extract_av1_tables.py LIB, offset 0x4842f0)
    'segment_id': [[5622, 7893, 16093, 18233, 27809, 28373, 32533], [14274, 18230, 22557, 24935, 29980, 30851, 32344], [27527, 28487, 28723, 28890, 32397, 32647, 32679]],
}

def rows_of(name):
    return T[name]['rows']

# ---- flat layout: list of (NAME, rows(list of cumulative lists), stride) ; q-dependent ones take q index
LAYOUT = []
def add(name, rows, nsym_max):
    LAYOUT.append((name, rows, nsym_max + 1))

def build(q):
    L = []
    def add(name, rows, nsym_max): L.append((name, rows, nsym_max + 1))
    add('KF_Y', rows_of('kf_y_mode'), 13)
    add('ANGLE', rows_of('angle_delta'), 7)
    add('UV_NOCFL', rows_of('uv_mode_nocfl'), 13)
    add('UV_CFL', rows_of('uv_mode_cfl'), 14)
    add('PARTITION', rows_of('partition'), 10)
    add('SKIP', SMALL['skip'], 2)
    add('INTRA_TX1', rows_of('intra_tx_set1'), 7)
    add('INTRA_TX2', rows_of('intra_tx_set2'), 5)
    add('CFL_SIGN', SMALL['cfl_sign'], 8)
    add('CFL_ALPHA', rows_of('cfl_alpha'), 16)
    add('TX_SIZE', rows_of('tx_size'), 3)
    add('SW_RESTORE', SMALL['switchable_restore'], 3)
    add('USE_WIENER', SMALL['use_wiener'], 2)
    add('USE_SGRPROJ', SMALL['use_sgrproj'], 2)
    add('SEG_ID', SMALL['segment_id'], 8)
    def qslice(name, per_q):
        r = rows_of(name); n = len(r) // 4
        return r[q * n:(q + 1) * n]
    add('TXB_SKIP', qslice('txb_skip', 0), 2)          # [5][13]
    add('EOB_EXTRA', qslice('eob_extra', 0), 2)        # [5][2][9]
    add('DC_SIGN', qslice('dc_sign', 0), 2)            # [2][3]
    add('COEFF_BR', qslice('coeff_br', 0), 4)          # [5][2][21]
    add('COEFF_BASE', qslice('coeff_base', 0), 4)      # [5][2][42]
    add('COEFF_BASE_EOB', qslice('coeff_base_eob', 0), 3)  # [5][2][4]
    for i, s in enumerate((16, 32, 64, 128, 256, 512, 1024)):
        add('EOB_PT_%d' % s, qslice('eob_pt_%d' % s, 0), 5 + i)   # [2][2]
    return L

def flat(L):
    out = []; offs = {}
    for name, rows, stride in L:
        offs[name] = (len(out), stride, len(rows))
        for r in rows:
            n = len(r) + 1                       # number of symbols
            if not any(r):                        # unused row
                out += [0] * stride; continue
            v = [32768 - x for x in r] + [0]
            v += [0] * (stride - len(v))          # padding + counter (=0)
            out += v
    return out, offs

# ---- scans (spec: Default_Scan_NxM, row-major position = row * w + col)
def default_scan(w, h):
    pos = []
    for d in range(w + h - 1):
        cells = [(r, d - r) for r in range(h) if 0 <= d - r < w]   # increasing row = going down-left
        if w == h:
            if d % 2 == 0: cells.reverse()     # even diagonals go up-right (start at left column)
        elif w > h:
            cells.reverse()                    # wide: up-right
        pos += [r * w + c for r, c in cells]
    return pos

def check_scans_against_libaom():
    """libaom >= 3.6 stores the transposed scan (position = col * h + row)."""
    lib = '/usr/local/lib/python3.10/dist-packages/pillow.libs/libavif-a883386a.so.16.4.1'
    if not os.path.exists(lib): return
    import numpy as np
    d = open(lib, 'rb').read()
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8)):
        s = default_scan(w, h)
        tr = [(p % w) * h + (p // w) for p in s]
        b = np.array(tr, dtype='<i2').tobytes()
        assert d.find(b) > 0, ('scan mismatch', w, h)
    print('scans verified against libaom rodata')

def carr(name, ctype, vals, per=16):
    s = f'static const {ctype} {name}[{len(vals)}] = {{\n'
    for i in range(0, len(vals), per):
        s += '  ' + ', '.join(str(v) for v in vals[i:i + per]) + ',\n'
    return s + '};\n'

def emit(path, guard, const_qual):
    flats = []; offs = None
    for q in range(4):
        f, offs = flat(build(q)); flats.append(f)
    total = len(flats[0])
    o = [f'/* GENERATED by tools/gen_tables.py -- AV1 spec constant tables. Do not edit. */',
         f'#ifndef {guard}', f'#define {guard}', '#include <stdint.h>', '']
    o.append('/* Flat per-tile CDF context: every CDF is N inverse-cumulative uint16 (last = 0) + 1 counter. */')
    for name, (off, stride, nrows) in offs.items():
        o.append(f'#define CDF_{name} {off}')
        o.append(f'#define CDF_{name}_STRIDE {stride}')
    o.append(f'#define CDF_TOTAL {total}')
    o.append('')
    allv = [v for f in flats for v in f]
    o.append(carr('av1_default_cdfs' if not const_qual else 'av1_default_cdfs', 'uint16_t', allv).replace(
        f'[{len(allv)}]', f'[4 * CDF_TOTAL]'))
    for k in ('dc_q8', 'ac_q8', 'dc_q10', 'ac_q10'):
        o.append(carr('av1_' + k, 'int16_t', T[k]['rows']))
        if const_qual:                            # the segment kernel looks the per-segment steps up on the device
            o.append(carr('av1_' + k + '_dev', 'int16_t', T[k]['rows']).replace('static const', const_qual))
    for n in (4, 8, 16, 32):
        a = carr(f'av1_default_scan_{n}x{n}', 'uint16_t', default_scan(n, n))
        if const_qual:
            a = a.replace('static const', const_qual)
        o.append(a)
    o.append('#endif')
    open(path, 'w').write('\n'.join(o) + '\n')
    return total

if __name__ == '__main__':
    check_scans_against_libaom()
    t = emit(os.path.join(ROOT, 'oracle', 'av1_tables.h'), 'ORACLE_AV1_TABLES_H', '')
    emit(os.path.join(ROOT, 'cavif_rs_amd', 'csrc', 'av1_tables.h'), 'MI_AV1_TABLES_H', 'static __device__ const')
    print('CDF_TOTAL =', t, 'uint16 per q-context')
