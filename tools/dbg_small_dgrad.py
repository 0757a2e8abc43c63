"""debugging aid: data gradient of a Conv2d(512 -> 512, k4, s2, p1) on a 4 x 4 gradient map, batch 4, bf16 storage: the un-split launch
(64 x 64 tile) against the split-K + fix-up launch of the same contraction."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap; pta_bootstrap.load()
from pose_transfer_amd.runtime import engine as E, lib as L
E.PRECISION = 3
N, Hs, Ws, cout, cin = int(os.environ.get("DBG_N", "4")), 4, 4, 512, int(os.environ.get("DBG_CIN", "512"))
torch.manual_seed(0)
reg = lambda x: E._reg_bf16(x.cuda().bfloat16().contiguous())
gy = reg(torch.randn(N, Hs, Ws, cout))
fwd = reg(torch.randn(N, 2 * Hs, 2 * Ws, cin))
aff = torch.stack([torch.rand(N) + 0.5, torch.rand(N) - 0.5], 1).float().cuda()
wp = (0.05 * torch.randn(4, 4, cout, cin)).cuda()
res = {}
for ks in (int(os.environ.get("DBG_KS", "4")), 1):
    grad = E._reg_bf16(torch.full((N, 2 * Hs, 2 * Ws, cin), float("nan"), dtype=torch.bfloat16, device="cuda"))
    dst = L.make_dst(grad, cin, fwd=fwd, aff=aff, act=L.ACT_LEAKY, accumulate=False)
    info = E._conv_dgrad(E.Act(gy, cout).src(), N, Hs, Ws, 1, 4, 2, 1, 2 * Hs, 2 * Ws, wp, cout, cin, [dst], ksplit=ks)
    torch.cuda.synchronize()
    res[ks] = grad.float()
    print("ksplit", ks, "tile code", info & 15, "finite", bool(torch.isfinite(grad.float()).all()), flush=True)
a, b = list(res.values())
print("max diff", float((a - b).abs().max()), "of", float(b.abs().max()))
