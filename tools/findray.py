import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustraytracer_amd as rr
from tests import oracle_ffi as O
sc = rr.Scene("plastic_dragon", 1.0, 871414, None, 1)
osc = O.OracleScene(sc)
ctx = rr.Context(0); gs = ctx.upload(sc)
cfg = rr.make_cfg(1024, 1024, 32)
for (py, px) in [(213, 680), (383, 722), (396, 408), (489, 586)]:
    for s in range(32):
        o, d, tmin, t, prim = osc.sample_rays(sc.camera, cfg, px, py, s)
        tg = np.zeros(len(t)); pg = np.zeros(len(t), dtype=np.int32)
        for tm in np.unique(tmin):
            m = tmin == tm
            a, b = ctx.intersect_batch(gs, o[m], d[m], float(tm))
            tg[m] = a; pg[m] = b
        bad = np.flatnonzero((pg != prim) | (tg != t))
        if len(bad):
            i = bad[0]
            print("pixel", px, py, "sample", s, "ray", i, "of", len(t), "tmin", tmin[i])
            print("  o", o[i].tolist(), "d", d[i].tolist())
            print("  oracle t %r prim %d | gpu t %r prim %d" % (t[i], prim[i], tg[i], pg[i]))
            tb, pb = osc.intersect_batch(o[i:i+1], d[i:i+1], float(tmin[i]), mode=O.BRUTE)
            print("  brute  t %r prim %d" % (tb[0], pb[0]))
print("done")
