import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustraytracer_amd as rr
from tests import oracle_ffi as O
preset = sys.argv[1]; faces = int(sys.argv[2]); W = int(sys.argv[3]); spp = int(sys.argv[4]); variant = int(sys.argv[5])
sc = rr.Scene(preset, 1.0, faces, None, variant)
ctx = rr.Context(0); gs = ctx.upload(sc)
cfg = rr.make_cfg(W, W, spp)
rg, ng, sg = ctx.render(gs, sc.camera, cfg)
osc = O.OracleScene(sc)
t = time.time(); ro, no, so = osc.render(sc.camera, cfg, O.ORDERED, 16); dt = time.time() - t
print("oracle %.1fs rays %d gpu rays %d" % (dt, so.rays, sg.rays))
print("counts gpu", sg.rays_extension, sg.rays_shadow, sg.rays_probe, sg.vertices_shaded)
print("counts ora", so.rays_extension, so.rays_shadow, so.rays_probe, so.vertices_shaded)
diff = np.argwhere(np.any(rg != ro, axis=2))
print("pixels differing:", len(diff), diff[:10].tolist())
img_g, img_o = rg / ng[..., None], ro / no[..., None]
print("rmse", float(np.sqrt(np.mean((img_g - img_o) ** 2))))
if len(diff):
    # find the offending sample of the first differing pixel
    py, px = diff[0]
    for s in range(spp):
        v, st = osc.sample(sc.camera, cfg, int(px), int(py), s)
        # GPU single-sample via a 1x1 window cannot isolate a sample; report oracle values only
    np.save("gpurun_out/bigparity_diff.npy", diff)
