import os, sys, subprocess, time, threading, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import conv_bench as cb
from pose_transfer_amd.runtime import engine as E, lib as L
N=4
def mk(kind,h,w,srcC,cout,relu=True):
    cin=sum(srcC); ho,wo=(2*h,2*w)
    srcs=[(torch.randn(N,h,w,c,device="cuda")) for c in srcC]
    acts=[E.Act(s,c,aff=torch.rand(N,2,device="cuda")+0.5) for s,c in zip(srcs,srcC)]
    W=torch.randn(4,4,cout,cin,device="cuda")*0.05
    out=torch.empty(N,ho,wo,cout,device="cuda")
    def f(): E._conv([a.src() for a in acts],N,h,w,L.ACT_RELU if relu else L.ACT_NONE,1,4,2,1,ho,wo,W,cout,cin,out=out)
    return f
samples=[]
stop=False
def poll():
    while not stop:
        o=subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' ",shell=True,capture_output=True,text=True).stdout
        samples.append(o.strip().replace("\n"," | "))
for relu in (True,False):
    f=mk("convT",64,64,[512,256,256],256,relu)
    f(); torch.cuda.synchronize()
    samples.clear(); stop=False
    th=threading.Thread(target=poll); th.start()
    t0=time.time(); n=0
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time()-t0<6:
        for _ in range(50): f()
        n+=50; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop=True; th.join()
    print("relu",relu,"us/launch",e0.elapsed_time(e1)*1e3/n, "TF", 137.4e9/(e0.elapsed_time(e1)*1e-3/n)/1e12)
    for s in samples[::max(1,len(samples)//6)]: print("   ",s)
