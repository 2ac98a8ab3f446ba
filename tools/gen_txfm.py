#!/usr/bin/env python3
"""Generate straight-line C for the AV1 1-D integer transforms.

Inverse transforms follow the AV1 specification section 7.13.2 (butterfly network with
Round2(x*cos128 +/- y*sin128, 12) rotations; every rounding point is normative).  The forward
transforms are NOT normative (rav1e uses its own Daala-derived butterflies, absent from
/root/reference); here they are the exact transposed flow graph of the inverse network, so that
forward(N) = sqrt(N/2) * orthonormal DCT/DST with one Round2 per rotation output.

The same stage lists drive:  * a float self-check (python, run at generation time),
                             * oracle/txfm_gen.h   (plain C, scalar),
                             * cavif_rs_amd/csrc/txfm_gen.hip.h (device functions; one lane owns a row).
Each op is one of
   ('rot', p, q, (a,b,c,d))  : p' = R12(a*p + b*q), q' = R12(c*p + d*q)   (a..d are signed 12-bit constants)
   ('had', p, q, sp, sq)     : p' = p + q or -p + q ..., exact adds (see emit)
   ('perm', [src...])        : x'[i] = x[src[i]]
   ('neg', [idx...])         : sign flips
"""
import math, sys, os

COS = [int(round(math.cos(i * math.pi / 128) * 4096)) for i in range(65)]
SINPI = [0, 1321, 2482, 3344, 3803]   # round(sin(k*pi/9)*2*sqrt(2)/3*4096)

def brev(nbits, x):
    r = 0
    for i in range(nbits):
        r |= ((x >> i) & 1) << (nbits - 1 - i)
    return r

def idct_ops(N):
    n = N.bit_length() - 1
    ops = [('perm', [brev(n, p) for p in range(N)])]
    def rec(sz):
        if sz == 2:
            ops.append(('rot', 0, 1, (COS[32], COS[32], COS[32], -COS[32])))
            return
        M = sz // 2
        rec(M)
        # odd part, positions M..sz-1 (spec 7.13.2.3 steps; libaom idctN stage pattern)
        for j in range(M // 2):
            p, q = M + j, sz - 1 - j
            k = brev(sz.bit_length() - 1, p)
            t = 64 * k // sz
            ops.append(('rot', p, q, (COS[64 - t], -COS[t], COS[t], COS[64 - t])))
        g = 2
        while g <= M // 2:
            for v in range(M // g):
                for i in range(g // 2):
                    lo, hi = M + v * g + i, M + v * g + g - 1 - i
                    if v % 2 == 0: ops.append(('had', lo, hi, 1, 1, 1, -1))    # lo'=lo+hi, hi'=lo-hi
                    else:          ops.append(('had', lo, hi, -1, 1, 1, 1))   # lo'=-lo+hi, hi'=lo+hi
            G = max(1, M // (4 * g))
            gb = G.bit_length() - 1
            for j in range(M // 2):
                r = j % (2 * g)
                if g // 2 <= r < 3 * g // 2:
                    u = j // (2 * g)
                    t = 64 * g // M + (64 // G) * brev(gb, u)
                    p, q = M + j, sz - 1 - j
                    if r < g: ops.append(('rot', p, q, (-COS[t], COS[64 - t], COS[64 - t], COS[t])))
                    else:     ops.append(('rot', p, q, (-COS[64 - t], -COS[t], -COS[t], COS[64 - t])))
            g *= 2
        for i in range(M):
            ops.append(('had', i, sz - 1 - i, 1, 1, 1, -1))
    rec(N)
    return ops

def iadst4_ops():
    return [('adst4',)]

def iadst_ops(N):
    """Inverse ADST8 / ADST16 (spec 7.13.2.7 / 7.13.2.8; same network as libaom av1_iadst8/16)."""
    C = COS
    def P(p, q, t):   # p' = c[t] p + c[64-t] q ; q' = c[64-t] p - c[t] q
        return ('rot', p, q, (C[t], C[64 - t], C[64 - t], -C[t]))
    def Q(p, q, t):   # p' = -c[64-t] p + c[t] q ; q' = c[t] p + c[64-t] q
        return ('rot', p, q, (-C[64 - t], C[t], C[t], C[64 - t]))
    def AS(lo, dist, cnt):
        return [('had', lo + i, lo + i + dist, 1, 1, 1, -1) for i in range(cnt)]
    ops = []
    if N == 8:
        ops.append(('perm', [7, 0, 5, 2, 3, 4, 1, 6]))
        ops += [P(0, 1, 4), P(2, 3, 20), P(4, 5, 36), P(6, 7, 52)]
        ops += AS(0, 4, 4)
        ops += [P(4, 5, 16), Q(6, 7, 16)]
        ops += AS(0, 2, 2) + AS(4, 2, 2)
        ops += [P(2, 3, 32), P(6, 7, 32)]
        ops.append(('perm', [0, 4, 6, 2, 3, 7, 5, 1]))
        ops.append(('neg', [1, 3, 5, 7]))
    else:
        ops.append(('perm', [15, 0, 13, 2, 11, 4, 9, 6, 7, 8, 5, 10, 3, 12, 1, 14]))
        ops += [P(2 * i, 2 * i + 1, 2 + 8 * i) for i in range(8)]
        ops += AS(0, 8, 8)
        ops += [P(8, 9, 8), P(10, 11, 40), Q(12, 13, 8), Q(14, 15, 40)]
        ops += AS(0, 4, 4) + AS(8, 4, 4)
        ops += [P(4, 5, 16), Q(6, 7, 16), P(12, 13, 16), Q(14, 15, 16)]
        ops += AS(0, 2, 2) + AS(4, 2, 2) + AS(8, 2, 2) + AS(12, 2, 2)
        ops += [P(2, 3, 32), P(6, 7, 32), P(10, 11, 32), P(14, 15, 32)]
        ops.append(('perm', [0, 8, 12, 4, 6, 14, 10, 2, 3, 11, 15, 7, 5, 13, 9, 1]))
        ops.append(('neg', [1, 3, 5, 7, 9, 11, 13, 15]))
    return ops

# ---------------------------------------------------------------- evaluation (python ints / floats)
def R12(x): return (x + 2048) >> 12

def run(ops, x, exact=False):
    x = list(x)
    for op in ops:
        if op[0] == 'perm':
            x = [x[s] for s in op[1]]
        elif op[0] == 'neg':
            for i in op[1]: x[i] = -x[i]
        elif op[0] == 'rot':
            _, p, q, (a, b, c, d) = op
            xp, xq = x[p], x[q]
            if exact: x[p], x[q] = (a * xp + b * xq) / 4096.0, (c * xp + d * xq) / 4096.0
            else:     x[p], x[q] = R12(a * xp + b * xq), R12(c * xp + d * xq)
        elif op[0] == 'had':
            _, p, q, a, b, c, d = op
            xp, xq = x[p], x[q]
            x[p], x[q] = a * xp + b * xq, c * xp + d * xq
        elif op[0] == 'adst4':
            s = SINPI
            x0, x1, x2, x3 = x
            s0 = s[1] * x0; s1 = s[2] * x0; s2 = s[3] * x1; s3 = s[4] * x2
            s4 = s[1] * x2; s5 = s[2] * x3; s6 = s[4] * x3
            b7 = x0 - x2 + x3
            s0 = s0 + s3; s1 = s1 - s4; s3 = s2; s2 = s[3] * b7
            s0 = s0 + s5; s1 = s1 - s6
            y = [s0 + s3, s1 + s3, s2, s0 + s1 - s3]
            x = [v / 4096.0 for v in y] if exact else [R12(v) for v in y]
        elif op[0] == 'fadst4':
            s = SINPI
            x0, x1, x2, x3 = x
            y = [s[1] * x0 + s[2] * x1 + s[3] * x2 + s[4] * x3,
                 s[3] * (x0 + x1 - x3),
                 s[4] * x0 - s[1] * x1 - s[3] * x2 + s[2] * x3,
                 s[2] * x0 - s[4] * x1 + s[3] * x2 - s[1] * x3]
            x = [v / 4096.0 for v in y] if exact else [R12(v) for v in y]
    return x

def transpose_ops(ops):
    """Forward network = reversed list of transposed elementary ops."""
    out = []
    for op in reversed(ops):
        if op[0] == 'perm':
            inv = [0] * len(op[1])
            for i, s in enumerate(op[1]): inv[s] = i
            out.append(('perm', inv))
        elif op[0] == 'neg':
            out.append(op)
        elif op[0] == 'rot':
            _, p, q, (a, b, c, d) = op
            out.append(('rot', p, q, (a, c, b, d)))
        elif op[0] == 'had':
            _, p, q, a, b, c, d = op
            out.append(('had', p, q, a, c, b, d))
        elif op[0] == 'adst4':
            out.append(('fadst4',))
    return out

def selfcheck():
    import random
    random.seed(1)
    for N in (4, 8, 16, 32, 64):
        ops = idct_ops(N)
        X = [random.randint(-2000, 2000) for _ in range(N)]
        y = run(ops, X, exact=True)
        for n_ in range(N):
            ref = sum((X[k] * (math.sqrt(0.5) if k == 0 else 1.0)) * math.cos((2 * n_ + 1) * k * math.pi / (2 * N)) for k in range(N))
            assert abs(ref - y[n_]) < 4e-4 * sum(abs(v) for v in X), ('idct', N, n_, ref, y[n_])
        # forward then inverse ~ N/2 * identity
        f = transpose_ops(ops)
        z = run(ops, run(f, X))
        assert max(abs(z[i] - X[i] * N // 2) for i in range(N)) <= N, ('fdct roundtrip', N)
    for N in (8, 16):
        ops = iadst_ops(N)
        X = [random.randint(-2000, 2000) for _ in range(N)]
        y = run(ops, X, exact=True)
        for n_ in range(N):
            ref = sum(X[k] * math.sin((2 * n_ + 1) * (2 * k + 1) * math.pi / (4 * N)) for k in range(N))
            assert abs(ref - y[n_]) < 4e-4 * sum(abs(v) for v in X), ('iadst', N, n_, ref, y[n_])
    ops = iadst4_ops()
    X = [100, -50, 25, 70]
    y = run(ops, X, exact=True)
    for n_ in range(4):
        ref = sum(X[k] * math.sin((n_ + 1) * (2 * k + 1) * math.pi / 9) for k in range(4)) * (2 * math.sqrt(2) / 3)
        assert abs(ref - y[n_]) < 0.5, ('iadst4', n_, ref, y[n_])
    z = run(iadst4_ops(), run(transpose_ops(iadst4_ops()), X))
    assert max(abs(z[i] - 2 * X[i]) for i in range(4)) <= 2
    print('txfm selfcheck ok')

# ---------------------------------------------------------------- C emission
def emit_fn(name, ops, N, qual):
    M = 'TX_MUL24' if N <= 16 else 'TX_MUL32'
    L = [f'{qual} void {name}(int32_t *x) {{']
    L.append('  int32_t t0, t1;')
    for op in ops:
        if op[0] == 'perm':
            src = op[1]
            if src == list(range(N)): continue
            L.append('  { int32_t y[%d] = {%s};' % (N, ', '.join(f'x[{s}]' for s in src)))
            L.append('    ' + ' '.join(f'x[{i}] = y[{i}];' for i in range(N)) + ' }')
        elif op[0] == 'neg':
            L.append('  ' + ' '.join(f'x[{i}] = -x[{i}];' for i in op[1]))
        elif op[0] == 'rot':
            _, p, q, (a, b, c, d) = op
            if N <= 16:
                # one output = R12(a*t0 + b*t1): two chained 24-bit multiply-adds (TX_ROT), or -- equal magnitudes, the 2896 butterflies -- ONE on the sum / difference
                # (TX_ROT1: a*t0 + b*t1 == a*(t0 +- t1) in wrapping 32-bit arithmetic; |t0 +- t1| < 2^21 here)
                def one(u, v):
                    if abs(u) == abs(v): return f'TX_ROT1({u}, t0 {"+" if u == v else "-"} t1)'
                    return f'TX_ROT({u}, t0, {v}, t1)'
                L.append(f'  t0 = x[{p}]; t1 = x[{q}]; x[{p}] = {one(a, b)}; x[{q}] = {one(c, d)};')
            else:
                L.append(f'  t0 = x[{p}]; t1 = x[{q}]; x[{p}] = TX_R12({M}({a}, t0) + {M}({b}, t1)); x[{q}] = TX_R12({M}({c}, t0) + {M}({d}, t1));')
        elif op[0] == 'had':
            _, p, q, a, b, c, d = op
            def term(s, v): return ('-' if s < 0 else '+') + v
            L.append(f'  t0 = x[{p}]; t1 = x[{q}]; x[{p}] = {term(a, "t0")} {term(b, "t1")}; x[{q}] = {term(c, "t0")} {term(d, "t1")};')
        elif op[0] == 'adst4':
            L.append('  { int32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];')
            L.append('    int32_t s0 = TX_MUL24(1321, x0), s1 = TX_MUL24(2482, x0), s2 = TX_MUL24(3344, x1), s3 = TX_MUL24(3803, x2), s4 = TX_MUL24(1321, x2), s5 = TX_MUL24(2482, x3), s6 = TX_MUL24(3803, x3);')
            L.append('    int32_t b7 = x0 - x2 + x3; s0 += s3; s1 -= s4; s3 = s2; s2 = TX_MUL24(3344, b7); s0 += s5; s1 -= s6;')
            L.append('    x[0] = TX_R12(s0 + s3); x[1] = TX_R12(s1 + s3); x[2] = TX_R12(s2); x[3] = TX_R12(s0 + s1 - s3); }')
        elif op[0] == 'fadst4':
            L.append('  { int32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];')
            L.append('    x[0] = TX_R12(TX_MUL24(1321, x0) + TX_MUL24(2482, x1) + TX_MUL24(3344, x2) + TX_MUL24(3803, x3));')
            L.append('    x[1] = TX_R12(TX_MUL24(3344, x0 + x1 - x3));')
            L.append('    x[2] = TX_R12(TX_MUL24(3803, x0) - TX_MUL24(1321, x1) - TX_MUL24(3344, x2) + TX_MUL24(2482, x3));')
            L.append('    x[3] = TX_R12(TX_MUL24(2482, x0) - TX_MUL24(3803, x1) + TX_MUL24(3344, x2) - TX_MUL24(1321, x3)); }')
    L.append('}')
    return '\n'.join(L)

def emit(path, qual, guard, mul_defs):
    out = [f'/* GENERATED by tools/gen_txfm.py -- do not edit.  AV1 1-D transforms (inverse: spec 7.13.2, normative;',
           '   forward: transposed network, encoder-side choice). In-place on int32 x[N]. */',
           f'#ifndef {guard}', f'#define {guard}', '#include <stdint.h>',
           '#define TX_R12(v) (((v) + 2048) >> 12)',
           '/* N <= 16: operands stay far below 2^23 (|sample| <= 2^20 after the stage shifts, |constant| <= 4096), so the\n   product may use the 24-bit multiplier; the low 32 bits are identical to a full multiply. */',
           mul_defs]
    for N in (4, 8, 16, 32, 64):
        ops = idct_ops(N)
        out.append(emit_fn(f'av1_idct{N}', ops, N, qual))
        out.append(emit_fn(f'av1_fdct{N}', transpose_ops(ops), N, qual))
    out.append(emit_fn('av1_iadst4', iadst4_ops(), 4, qual))
    out.append(emit_fn('av1_fadst4', transpose_ops(iadst4_ops()), 4, qual))
    for N in (8, 16):
        ops = iadst_ops(N)
        out.append(emit_fn(f'av1_iadst{N}', ops, N, qual))
        out.append(emit_fn(f'av1_fadst{N}', transpose_ops(ops), N, qual))
    out.append('#endif')
    open(path, 'w').write('\n'.join(out) + '\n')

if __name__ == '__main__':
    selfcheck()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rot_c = ('#define TX_ROT(a, p, b, q) TX_R12((a) * (p) + (b) * (q))\n#define TX_ROT1(a, s) TX_R12((a) * (s))')
    # The device form pins the instruction selection: v_mad_i32_i24 chains with the rounding constant as the first addend.  Left to itself LLVM turns __mul24 into a
    # plain multiply of sign-extended operands, re-associates the two outputs of a rotation around a shared partial sum and ends with 32-bit multiplies, explicit
    # v_bfe_i32 sign extensions and three-operand adds: 11 .. 13 issue slots per rotation instead of 7 .. 9 (tools/probe/valu_rates.hip has the per-instruction rates).
    rot_hip = ('#ifndef MI_MAD24                                     /* (the CPU test harness tests/emu/ predefines it, like MI_SMUL32 in tile_entropy.h) */\n'
               '#define MI_MAD24(r, x, c, acc) asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(c), "v"(acc))\n#endif\n'
               'static __device__ __forceinline__ int32_t tx_mad24(int32_t x, int32_t c, int32_t acc) { int32_t r; MI_MAD24(r, x, c, acc); return r; }\n'
               '#define TX_ROT(a, p, b, q) (tx_mad24((q), (b), tx_mad24((p), (a), 2048)) >> 12)\n#define TX_ROT1(a, s) (tx_mad24((s), (a), 2048) >> 12)')
    emit(os.path.join(root, 'oracle', 'txfm_gen.h'), 'static inline', 'ORACLE_TXFM_GEN_H', '#define TX_MUL24(a, b) ((a) * (b))\n#define TX_MUL32(a, b) ((a) * (b))\n' + rot_c)
    emit(os.path.join(root, 'cavif_rs_amd', 'csrc', 'txfm_gen.hip.h'), 'static __device__ __forceinline__', 'MI_TXFM_GEN_HIP_H', '#define TX_MUL24(a, b) __mul24((a), (b))\n#define TX_MUL32(a, b) ((a) * (b))\n' + rot_hip)
    print('wrote oracle/txfm_gen.h, cavif_rs_amd/csrc/txfm_gen.hip.h')
