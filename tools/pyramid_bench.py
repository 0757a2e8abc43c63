import sys, os, torch
sys.path.insert(0, os.getcwd())
import pta_bootstrap; pta_bootstrap.load()
from pose_transfer_amd.runtime import lib as L
N,T,H,W=32,10,256,256
m=torch.rand(N,T,H,W,device="cuda")
for lvl in range(4):
    h,w=H>>lvl,W>>lvl
    out=torch.empty(N,h,w,T,device="cuda")
    for _ in range(3): L.call("pg_mask_pyramid", L.ptr(m), 0, N,T,H,W,h,w,L.ptr(out),L.stream())
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): L.call("pg_mask_pyramid", L.ptr(m), 0, N,T,H,W,h,w,L.ptr(out),L.stream())
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*100
    by=(m.numel() if lvl==0 else 4*out.numel())*4+out.numel()*4
    print("level %d: %.1f us, %.2f TB/s"%(lvl,us,by/us/1e6))
