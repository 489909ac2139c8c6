import time, torch
torch.cuda.init(); dev=torch.device("cuda:0")
x=torch.zeros(1024,device=dev)
def fence():
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
    torch.cuda.synchronize()
for _ in range(5): fence()
ts=[]
for _ in range(200):
    t0=time.perf_counter(); fence(); ts.append(time.perf_counter()-t0)
ts.sort(); print("idle fence us: median %.1f min %.1f" % (ts[100]*1e6, ts[0]*1e6))
ts=[]
for _ in range(200):
    t0=time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
ts.sort(); print("idle synchronize us: median %.1f" % (ts[100]*1e6))
ts=[]
for _ in range(200):
    t0=time.perf_counter(); e=torch.cuda.Event(); e.record(); ts.append(time.perf_counter()-t0)
ts.sort(); print("event create+record us: median %.1f" % (ts[100]*1e6))
e=torch.cuda.Event(); e.record(); torch.cuda.synchronize()
ts=[]
for _ in range(200):
    t0=time.perf_counter(); e.query(); ts.append(time.perf_counter()-t0)
ts.sort(); print("event query (done) us: median %.1f" % (ts[100]*1e6))
# kernel of ~50us then fence: overhead after the kernel's end
y=torch.zeros(64*1024*1024,device=dev)
s=torch.cuda.Event(enable_timing=True); f=torch.cuda.Event(enable_timing=True)
for _ in range(3):
    y.add_(1.0); fence()
ov=[]
for _ in range(50):
    fence()
    t0=time.perf_counter(); s.record(); y.add_(1.0); f.record(); fence(); dt=time.perf_counter()-t0
    ov.append(dt*1e6 - s.elapsed_time(f)*1e3)
ov.sort(); print("host region minus GPU event span us: median %.1f min %.1f" % (ov[25], ov[0]))
