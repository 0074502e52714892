"""Frames in flight: the same C2 frames rendered from N contexts (N host threads) on one GPU, so that the
endgame of one frame overlaps the start of the next.  usage: python tools/inflight.py <contexts> <frames>"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustraytracer_amd as rr
W=H=512; spp=64
scene = rr.Scene("cornell_box_statue", 1.0, mesh_faces=400000, variant=0)
nctx = int(sys.argv[1]); K = int(sys.argv[2])
ctxs = [rr.Context(0) for _ in range(nctx)]
gss = [c.upload(scene) for c in ctxs]
films = [(torch.zeros((H,W,3),dtype=torch.float64,device="cuda"), torch.zeros((H,W),dtype=torch.int32,device="cuda")) for _ in range(nctx)]
cfg = rr.make_cfg(W,H,spp)
def work(i, n, out):
    rays = 0
    for _ in range(n):
        st = ctxs[i].render_device(gss[i], scene.camera, cfg, films[i][0].data_ptr(), films[i][1].data_ptr())
        rays += st.rays
    out[i] = rays
for i in range(nctx): work(i, 1, [0]*nctx)  # warmup
torch.cuda.synchronize()
t0 = time.perf_counter()
out = [0]*nctx
ths = [threading.Thread(target=work, args=(i, K//nctx, out)) for i in range(nctx)]
[t.start() for t in ths]; [t.join() for t in ths]
torch.cuda.synchronize()
dt = time.perf_counter()-t0
print(f"contexts {nctx} frames {K} -> {sum(out)/dt/1e6:.1f} Mrays/s, {dt/K*1e3:.2f} ms/frame")
