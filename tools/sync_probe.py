#!/usr/bin/env python3
"""What a launch + a synchronisation cost on this box (HIP runtime, one stream): the floor under every per-call figure of the JPEG encoder
(a call = 27 us + 9.2 us per frame).  MI355X, ROCm 7.2: one tiny kernel + hipStreamSynchronize 12.6 us, a second kernel +5-7 us; an
event spin or a host spin on a pinned flag are no faster (11.6-17.5 us) -- hipStreamSynchronize already spins.  GPU box."""
import torch, time, ctypes as C
x=torch.zeros(64,device='cuda')
hip=C.CDLL('libamdhip64.so')
st=torch.cuda.current_stream().cuda_stream
def loop(n,f):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t)/n*1e6
def a():
    x.add_(1); torch.cuda.synchronize()
def b():
    x.add_(1); x.add_(1); torch.cuda.synchronize()
def c():
    x.add_(1); hip.hipStreamSynchronize(C.c_void_p(st))
ev=torch.cuda.Event()
def d():
    x.add_(1); ev.record()
    while not ev.query(): pass
print("1 tiny kernel + device sync  %.1f us"%loop(2000,a))
print("2 tiny kernels + device sync %.1f us"%loop(2000,b))
print("1 tiny kernel + stream sync  %.1f us"%loop(2000,c))
print("1 tiny kernel + event spin   %.1f us"%loop(2000,d))
# pinned flag spin
flag=torch.zeros(1,dtype=torch.int32).pin_memory()
import numpy as np
fl=flag.numpy()
dflag=torch.zeros(1,dtype=torch.int32,device='cuda')
def e():
    k=e.k=getattr(e,'k',0)+1
    dflag.fill_(k); flag.copy_(dflag,non_blocking=True)
    while fl[0]!=k: pass
print("fill + async D2H to pinned + host spin %.1f us"%loop(2000,e))
