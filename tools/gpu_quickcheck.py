import sys, time, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from tests.helpers import oracle
import cavif_rs_amd as m
print(m.load_library().mi_version(), 'devices', m.device_count())
def synth(h,w,seed=0,bd=8):
    rng=np.random.default_rng(seed)
    y,x=np.mgrid[0:h,0:w]
    base=[(x*2+y)%256, (y*3)%256, ((x+y)//2)%256]
    pl=[]
    for b in base:
        v=b+rng.integers(-6,7,size=(h,w))+40*np.sin(x/9.0+seed)+30*np.cos(y/7.0)
        v[h//4:h//2, w//3:w//2]=200
        v=np.clip(v,0,255).astype(np.uint16)
        if bd==10: v=(v<<2)|(v>>6)
        pl.append(v)
    return pl
def check(w,h,bd=8,speed=4,q=121,mono=False,tiles=0,**over):
    pl=synth(h,w,0,bd)
    if mono: pl=pl[:1]
    cfg=oracle.make_config(w,h,bd,mono,q,speed,tiles=tiles)
    r=oracle.encode_planes(cfg,pl)
    t=time.time()
    obu,rec=m.encode_planes(pl,bd,q,speed,mono,tiles=tiles)
    dt=time.time()-t
    ok=obu==r['obu']; okr=all(np.array_equal(a,b) for a,b in zip(rec,r['recon']))
    print(f'{w}x{h} bd{bd} s{speed} q{q} mono{int(mono)} bytes gpu={len(obu)} cpu={len(r["obu"])} bytes_equal={ok} recon_equal={okr} t={dt:.2f}s', flush=True)
    if not okr:
        for i,(a,b) in enumerate(zip(rec,r['recon'])):
            bad=np.argwhere(a!=b)
            if len(bad): print('   plane',i,'mismatch count',len(bad),'first',bad[0])
    return ok and okr
for args in [(64,64),(128,96),(129,101),(129,101,10),(256,200,10,4,66,True),(300,270,10,4,121,False,4),(200,120,10,1),(136,72,8,10)]:
    try: check(*args)
    except Exception as e: print('ERR',args,e)
