for lib in libmi_avif.so libmi_dbg1.so libmi_dbg2.so libmi_dbg3.so; do
  MI_AVIF_LIB=$(pwd)/cavif_rs_amd/$lib python - <<'PY'
import sys, os, time
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
B=8
b = m.BatchEncoder(e, B, 1920, 1080, 3)
for i in range(B): b.upload(i, synth_image(1920, 1080, index=i))
b.encode(); b.encode()
print(os.environ['MI_AVIF_LIB'].split('/')[-1], {k: round(v,2) for k,v in b.stage_ms().items()})
PY
done
