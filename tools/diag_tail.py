import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr
sc = rr.cornell_box_statue(mesh_faces=400000, variant=0)
ctx = rr.Context(0); gs = ctx.upload(sc)
cfg = rr.make_cfg(512,512,64, count_traversal=True)
for _ in range(2):
    r,n,st = ctx.render(gs, sc.camera, cfg)
print("tail launches", st.reserved[1], "avg in-kernel us", st.reserved[0]/max(st.reserved[1],1)/100.0, "avg rays", st.reserved[2]/max(st.reserved[1],1), "trace_ms", st.trace_ms, "launches", st.trace_launches)

r3=st.reserved[3]; print("max steps/ray", r3 & 0xffff, "rays>64 steps", (r3>>16)&0xffffff, "rays>256 steps", r3>>40, "of", st.rays)
