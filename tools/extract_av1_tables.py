#!/usr/bin/env python3
"""Extract the AV1 *normative* default CDF tables (AV1 spec section "Default CDF tables")
from the libaom copy that ships inside Pillow's bundled libavif, and emit them as a C header.

Why: the AV1 specification's default CDFs (~15k numbers) are constants every conforming
encoder must start from; there is no network in the build container and no copy of the spec,
but libaom 3.13.2's .rodata holds them verbatim in AOM_CDFn() layout
(n-1 values stored as 32768-x, then a 0 terminator, then a 0 adaptation counter).
The offsets are located by searching for the (well known) first row of each table, the
array shapes are those of the AV1 spec, and every row is validated structurally.

Output values are in the *spec* convention: cumulative probabilities x (increasing, <32768),
n-1 per row; the oracle/product append 32768 and the counter themselves.

This script is a build-time data tool; the generated headers are committed.
"""
import sys, json, numpy as np

LIB = '/usr/local/lib/python3.10/dist-packages/pillow.libs/libavif-a883386a.so.16.4.1'

def load():
    return open(LIB, 'rb').read()

def find_unique(d, vals, lo=0x440000, hi=0x460000, first=False):
    b = np.array([32768 - x for x in vals], dtype='<u2').tobytes()
    i = d.find(b, lo)
    assert 0 <= i < hi, ('not found', vals)
    j = d.find(b, i + 1)
    assert first or j < 0 or j >= hi, ('ambiguous', vals, hex(i), hex(j))
    return i

def read_rows(d, off, nrows, nsym, slot=None):
    """nrows CDFs of nsym symbols stored in slots of `slot` u16 (default nsym+1)."""
    slot = slot or nsym + 1
    a = np.frombuffer(d, dtype='<u2', count=nrows * slot, offset=off).reshape(nrows, slot).astype(np.int64)
    out = []
    for r in a:
        if not r.any():           # unused (all-zero) row
            out.append([0] * (nsym - 1)); continue
        v = [32768 - int(x) for x in r[:nsym - 1]]
        assert r[nsym - 1] == 0 and all(x == 0 for x in r[nsym:]), (hex(off), r)
        assert all(0 < v[i] and (i == 0 or v[i] >= v[i - 1]) for i in range(len(v))) and v[-1] <= 32768, (hex(off), v)
        out.append(v)
    return out

def main():
    d = load()
    T = {}
    def tab(name, first, shape, nsym, slot=None, dup=False):
        off = find_unique(d, first, first=dup)
        n = int(np.prod(shape))
        rows = read_rows(d, off, n, nsym, slot)
        T[name] = dict(shape=list(shape), nsym=nsym, rows=rows)
        return off
    # ---- mode info (spec: Default_*_Cdf) ----
    tab('kf_y_mode', [15588, 17027, 19338, 20218, 20682], (5, 5), 13)
    tab('angle_delta', [2180, 5032, 7567, 22776, 26989, 30217], (8,), 7)
    # uv mode: [cfl_allowed][y_mode]; cfl-not-allowed rows have 13 symbols in 15-wide slots
    o = find_unique(d, [22631, 24152, 25378, 25661, 25986])
    T['uv_mode_nocfl'] = dict(shape=[13], nsym=13, rows=read_rows(d, o, 13, 13, 15))
    T['uv_mode_cfl'] = dict(shape=[13], nsym=14, rows=read_rows(d, o + 13 * 15 * 2, 13, 14, 15))
    # partition: 20 contexts in 11-wide slots: ctx 0..3 (8x8) 4 syms, 4..15 10 syms, 16..19 (128) 8 syms
    o = find_unique(d, [19132, 25510, 30392])
    rows = []
    for c in range(20):
        ns = 4 if c < 4 else (10 if c < 16 else 8)
        rows.append(read_rows(d, o + c * 11 * 2, 1, ns, 11)[0])
    T['partition'] = dict(shape=[20], nsym=-1, rows=rows)
    # intra tx type: [set 1: 7 syms][tx 4x4, 8x8][13 modes]; [set 2: 5 syms][4x4, 8x8, 16x16][13]
    o1 = find_unique(d, [1535, 8035, 9461, 12751, 23467, 27825])
    T['intra_tx_set1'] = dict(shape=[2, 13], nsym=7, rows=read_rows(d, o1, 26, 7, 17))
    o2 = o1 + 4 * 13 * 17 * 2
    T['intra_tx_set2'] = dict(shape=[3, 13], nsym=5, rows=read_rows(d, o2, 39, 5, 17))
    assert T['intra_tx_set2']['rows'][26][:4] == [1127, 12814, 22772, 27483]
    tab('cfl_alpha', [7637, 20719, 31401, 32481, 32657], (6,), 16)
    o = find_unique(d, [12272, 30172], first=True) - 3 * 4 * 2   # tx_size: [4 cats][3 ctx], 4-wide slots, cat 0 has 2 syms
    rows = []
    for c in range(12):
        rows.append(read_rows(d, o + c * 4 * 2, 1, 2 if c < 3 else 3, 4)[0])
    T['tx_size'] = dict(shape=[4, 3], nsym=-1, rows=rows)
    assert rows[0] == [19968] and rows[11] == [16803, 22759]
    # ---- coefficient CDFs: leading dim = 4 qindex categories ----
    tab('txb_skip', [31849, 32768, 32768, 5892], (4, 5, 13), 2)
    tab('eob_extra', [16961, 32768, 32768, 17223], (4, 5, 2, 9), 2)
    tab('dc_sign', [128 * 125, 32768, 32768, 128 * 102], (4, 2, 3), 2, dup=True)
    tab('coeff_br', [14298, 20718, 24174], (4, 5, 2, 21), 4)
    tab('coeff_base', [4034, 8930, 12727], (4, 5, 2, 42), 4)
    tab('coeff_base_eob', [17837, 29055], (4, 5, 2, 4), 3)
    tab('eob_pt_16', [840, 1039, 1980, 4895], (4, 2, 2), 5)
    tab('eob_pt_32', [400, 520, 977, 2102, 6542], (4, 2, 2), 6)
    tab('eob_pt_64', [329, 498, 1101, 1784, 3265, 7758], (4, 2, 2), 7)
    tab('eob_pt_128', [219, 482, 1140, 2091, 3680, 6028, 12586], (4, 2, 2), 8)
    tab('eob_pt_256', [310, 584, 1887, 3589, 6168, 8611, 11352, 15652], (4, 2, 2), 9)
    tab('eob_pt_512', [641, 983, 3707, 5430, 10234, 14958, 18788, 23412, 26061], (4, 2, 2), 10)
    tab('eob_pt_1024', [393, 421, 751, 1623, 3160, 6352, 13345, 18047, 22571, 25830], (4, 2, 2), 11)
    # ---- quantizer lookup (spec: Dc_Qlookup / Ac_Qlookup), int16[256], 8- and 10-bit ----
    def q16(first):
        b = np.array(first, dtype='<i2').tobytes()
        i = d.find(b, 0x440000); assert i > 0 and d.find(b, i + 1) < 0
        v = [int(x) for x in np.frombuffer(d, dtype='<i2', count=256, offset=i)]
        assert all(v[k] <= v[k + 1] for k in range(255))
        return v
    T['dc_q8'] = dict(shape=[256], nsym=0, rows=q16([4, 8, 8, 9, 10, 11, 12, 12, 13, 14, 15, 16]))
    T['ac_q8'] = dict(shape=[256], nsym=0, rows=q16([4, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18]))
    T['dc_q10'] = dict(shape=[256], nsym=0, rows=q16([4, 9, 10, 13, 15, 17, 20, 22, 25, 28, 31, 34]))
    T['ac_q10'] = dict(shape=[256], nsym=0, rows=q16([4, 9, 11, 13, 16, 18, 21, 24, 27, 30, 33, 37]))
    json.dump(T, open(sys.argv[1] if len(sys.argv) > 1 else 'tools/av1_default_cdfs.json', 'w'))
    for k, v in T.items():
        print(k, v['shape'], v['nsym'], len(v['rows']), v['rows'][0])

if __name__ == '__main__':
    main()
