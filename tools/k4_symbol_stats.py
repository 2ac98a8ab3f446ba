#!/usr/bin/env python3
"""How the entropy stage's symbols spread over the CDF rows: builds an instrumented COPY of the CPU oracle in a
scratch directory (a counter per CDF row offset in the tile writer's symbol sink; the tree is not touched), encodes the bench image with it
(1920x1080 synthetic, speed 4, q80, 10-bit, 32 tiles) and prints symbols per tile, rows used, the hottest rows and how a `row & (A - 1)`
adapter assignment splits the load.  CPU only; about 15 s."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PATCH_FROM = 'static void ec_sym(void *u, int off, int s, int n) { TileW *w = (TileW *)u; re_symbol(&w->ec, s, w->cdf + off, n); }'
PATCH_TO = '''#include <stdio.h>
static long g_hist[CDF_TOTAL]; static int g_reg;
static void dump_hist(void) { FILE *fp = fopen(getenv("K4_HIST_OUT"), "w"); for (int i = 0; i < CDF_TOTAL; i++) if (g_hist[i]) fprintf(fp, "%d %ld\\n", i, g_hist[i]); fclose(fp); }
static void ec_sym(void *u, int off, int s, int n) { TileW *w = (TileW *)u; if (!g_reg) { g_reg = 1; atexit(dump_hist); } g_hist[off]++; re_symbol(&w->ec, s, w->cdf + off, n); }'''


def main():
    tmp = tempfile.mkdtemp(prefix='k4stats_')
    try:
        src = os.path.join(tmp, 'oracle')
        shutil.copytree(os.path.join(ROOT, 'oracle'), src, ignore=shutil.ignore_patterns('_build', '_ref'))
        p = os.path.join(src, 'av1o_entropy.c')
        text = open(p).read()
        assert text.count(PATCH_FROM) == 1, 'oracle/av1o_entropy.c: the symbol sink has changed, update PATCH_FROM'
        open(p, 'w').write(text.replace(PATCH_FROM, PATCH_TO))
        subprocess.check_call(['make', '-s', '-C', src], stderr=subprocess.DEVNULL)
        out = os.path.join(tmp, 'hist.txt')
        code = ("import sys; sys.path.insert(0, %r)\nfrom tests.helpers import oracle\nfrom cavif_rs_amd.synth import synth_image\n"
                "oracle.ravif_encode(synth_image(1920, 1080, index=0), quality=80, speed=4, depth=10)\n" % ROOT)
        subprocess.check_call([sys.executable, '-c', code], env=dict(os.environ, MI_ORACLE_LIB=os.path.join(src, '_build', 'liboracle.so'), K4_HIST_OUT=out))
        rows = sorted((tuple(map(int, l.split())) for l in open(out)), key=lambda r: -r[1])
        tot = sum(c for _, c in rows)
        print('adaptive symbols: %d in 32 tiles = %d per tile; CDF rows used: %d' % (tot, tot // 32, len(rows)))
        print('hottest rows (offset, share %):', [(o, round(100.0 * c / tot, 1)) for o, c in rows[:8]])
        acc = 0
        for i, (_, c) in enumerate(rows):
            acc += c
            if i + 1 in (16, 64, 128):
                print('top %d rows: %.1f %%' % (i + 1, 100.0 * acc / tot))
        for na in (2, 4, 8):
            load = [0] * na
            for o, c in rows:
                load[o & (na - 1)] += c
            print('%d adapters by row & %d: %s %%' % (na, na - 1, [round(100.0 * x / tot, 1) for x in load]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
