import os, sys, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap; pta_bootstrap.load()
from pose_transfer_amd.runtime import lib as L
from pose_transfer_amd.utils import synth
N, C, h, w, H0, W0, T = 1, 4, 16, 16, 16, 16, 10
wr, mk = synth.warps_and_masks(12, "dbg", N, H0, W0)
g = torch.from_numpy(synth.normal(12, "dbg/g", (N, h, w, C))).cuda()
feat = torch.from_numpy(synth.normal(12, "dbg/f", (N, h, w, C))).cuda()
wrd, mkd = torch.from_numpy(wr).cuda(), torch.from_numpy(mk).cuda()
lvl = torch.empty(N, h, w, T, device="cuda")
L.call("pg_mask_pyramid", L.ptr(mkd), 0, N, T, H0, W0, h, w, L.ptr(lvl), L.stream())
out = torch.empty(N, h, w, C, device="cuda"); arg = torch.empty(N, h, w, C, dtype=torch.uint8, device="cuda")
L.call("pg_warp_mask_max_fwd", L.ptr(feat), None, L.ptr(wrd), L.ptr(lvl), N, T, C, h, w, H0, W0, 0, L.ptr(out), L.ptr(arg), L.stream())
d = torch.full((N, h, w, C), 7.0, device="cuda")
L.call("pg_warp_mask_max_bwd", L.ptr(g), L.ptr(arg), L.ptr(wrd), L.ptr(lvl), N, T, C, h, w, H0, W0, 0, L.ptr(d), L.stream())
torch.cuda.synchronize()
if os.environ.get("PG_WARP_BWD_SCATTER"):
    torch.save(d.cpu(), "/tmp/dbg_scatter.pt"); sys.exit(0)
subprocess.run([sys.executable, __file__], env=dict(os.environ, PG_WARP_BWD_SCATTER="1"), check=True)
ref = torch.load("/tmp/dbg_scatter.pt")
diff = (d.cpu() - ref).abs()
print("max diff", float(diff.max()), "frac bad", float((diff > 1e-5).float().mean()))
bad = (diff > 1e-5).any(-1)[0]
print("bad pixel map (Y rows):"); print(bad.int().numpy())
print("argmax channel 0:"); print(arg[0, :, :, 0].cpu().numpy())
ys, xs = np.nonzero(bad.numpy())
for y, x in list(zip(ys, xs))[:6]:
    print("pixel", y, x, "got", d[0, y, x].cpu().numpy(), "ref", ref[0, y, x].numpy())
print("warps", wr[0, :, :6])
