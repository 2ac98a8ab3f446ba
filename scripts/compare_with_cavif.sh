#!/bin/bash
# Deferred true-parity kit.  On a machine WITH cargo/rustc (none exists in the build image):
#   cargo install cavif --version <the version under test>      # or build kornelski/cavif-rs at the reference commit
#   scripts/compare_with_cavif.sh /path/to/cavif [/path/to/cavif_mi]
# For every BASELINE config it generates the synthetic PNG inputs (scripts/gen_synth_png.py, integer-only generator,
# byte-identical to cavif_rs_amd/synth.py), runs the reference binary with the reference-equivalent flags of SURVEY 8(d),
# and compares its output with (a) this build's cavif_mi when a GPU is present, (b) the sha256 manifest
# scripts/parity_manifest.json that this build committed.  Known status (recorded in BASELINE.md): this has NOT been run;
# the oracle is known not to be byte-identical to rav1e (e.g. ravif/src/lib.rs:71-119: 129x101 q33 s10 colour payload is
# "expected ~= 215" bytes in the reference, this build emits a different size -- see BASELINE.md section "Known divergence").
set -euo pipefail
CAVIF=${1:?usage: compare_with_cavif.sh /path/to/cavif [/path/to/cavif_mi]}
MI=${2:-$(dirname "$0")/../cavif_rs_amd/cavif_mi}
T=${T:-$(nproc)}                      # -j T on both sides: T bounds the tile target (ravif av1encoder.rs:665-668)
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
HERE=$(cd "$(dirname "$0")" && pwd)
python3 "$HERE/gen_synth_png.py" "$W/in" --configs 2 3 4s 5
declare -A FLAGS=( [cfg2]="-s4 -Q80 --depth=10" [cfg3]="-s4 -Q80" [cfg4s]="-s4 -Q80 --depth=10" [cfg5]="-s1 -Q80 --depth=10" )
rc=0
for cfg in cfg2 cfg3 cfg4s cfg5; do
  mkdir -p "$W/ref/$cfg" "$W/mi/$cfg"
  "$CAVIF" ${FLAGS[$cfg]} -j"$T" --overwrite -q -o "$W/ref/$cfg" "$W/in/$cfg"/*.png
  if [ -x "$MI" ] && "$MI" ${FLAGS[$cfg]} -j"$T" -f -q -o "$W/mi/$cfg" "$W/in/$cfg"/*.png 2>/dev/null; then
    for f in "$W/ref/$cfg"/*.avif; do
      b=$(basename "$f")
      if cmp -s "$f" "$W/mi/$cfg/$b"; then echo "IDENTICAL $cfg/$b"; else echo "DIFFERENT $cfg/$b ($(stat -c %s "$f") vs $(stat -c %s "$W/mi/$cfg/$b") bytes)"; rc=1; fi
    done
  else
    echo "cavif_mi not runnable here (no GPU?): comparing the reference with the committed manifest only"
  fi
  for f in "$W/ref/$cfg"/*.avif; do
    python3 - "$HERE/parity_manifest.json" "$cfg" "$f" "$T" <<'PY'
import hashlib, json, os, sys
man, cfg, f, T = json.load(open(sys.argv[1])), sys.argv[2], sys.argv[3], sys.argv[4]
want = man['files'].get('%s/%s@j%s' % (cfg, os.path.basename(f), T)) or man['files'].get('%s/%s' % (cfg, os.path.basename(f)))
got = hashlib.sha256(open(f, 'rb').read()).hexdigest()
print('manifest', cfg, os.path.basename(f), 'MATCH' if want and want['sha256'] == got else 'MISMATCH (reference %d bytes, this build %s)' % (os.path.getsize(f), want and want['bytes']))
PY
  done
done
exit $rc
