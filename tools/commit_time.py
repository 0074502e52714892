import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr
ctx = rr.Context(0)
for preset, faces in (("cornell_box_statue", 400000), ("plastic_dragon", 871414), ("two_dragons", 871414)):
    t = time.time(); sc = rr.Scene(preset, 1.0, faces, None, 0); t1 = time.time() - t
    t = time.time(); gs = ctx.upload(sc); t2 = time.time() - t
    print(preset, faces, "host scene build %.2fs  rt_scene_* + commit (BVH + upload) %.2fs" % (t1, t2), gs.info())
    gs.close()
