import sys, os, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
print(m.load_library().mi_version(), 'devices', m.device_count(), flush=True)
h=w=64
y,x=np.mgrid[0:h,0:w]
rng=np.random.default_rng(0)
pl=[np.clip(b+rng.integers(-20,21,size=(h,w)),0,255).astype(np.uint16) for b in ((x*2+y)%256,(y*3)%256,((x+y)//2)%256)]
obu,rec=m.encode_planes(pl,8,121,4,False)
print('ok', len(obu), flush=True)
