"""Box facts for the host pipeline design: pinned alloc rate, H2D/D2H pageable vs pinned, host memcpy, file create rate."""
import os, time, tempfile, shutil, threading
import numpy as np, torch
def T(f, n=1):
    t=time.perf_counter(); [f() for _ in range(n)]; torch.cuda.synchronize(); return (time.perf_counter()-t)/n
torch.zeros(1).cuda()
G = 1<<30
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for mb in (64, 256, 1024):
    t=time.perf_counter(); p = torch.empty(mb<<20, dtype=torch.uint8, pin_memory=True); dt=time.perf_counter()-t
    print(f"pinned alloc {mb} MiB: {dt*1e3:.1f} ms")
pin = p
pag = torch.empty(G, dtype=torch.uint8); pag.fill_(1)
pin.fill_(2)
d = torch.empty(G, dtype=torch.uint8, device="cuda")
for name, h in (("pageable", pag), ("pinned", pin)):
    d.copy_(h); 
    print(f"H2D {name}: {G/T(lambda: d.copy_(h, non_blocking=True),3)/1e9:.1f} GB/s")
    h.copy_(d)
    print(f"D2H {name}: {G/T(lambda: h.copy_(d, non_blocking=True),3)/1e9:.1f} GB/s")
a = np.ones(G, np.uint8); b = np.empty(G, np.uint8); b[:] = 0
t=time.perf_counter(); b[:] = a; print(f"host memcpy 1 thread: {G/(time.perf_counter()-t)/1e9:.1f} GB/s")
pn = pin.numpy()
t=time.perf_counter(); pn[:] = a; print(f"host memcpy -> pinned: {G/(time.perf_counter()-t)/1e9:.1f} GB/s")
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
for nt in (1, 4, 16):
    dd = tempfile.mkdtemp(dir=base); n = 40000; buf = bytes(10240)
    def work(k):
        for i in range(k, n, nt):
            fd = os.open(f"{dd}/f{i}", os.O_CREAT|os.O_WRONLY|os.O_TRUNC, 0o644); os.pwrite(fd, buf, 0); os.close(fd)
    t=time.perf_counter(); ths=[threading.Thread(target=work,args=(k,)) for k in range(nt)]; [x.start() for x in ths]; [x.join() for x in ths]
    dt=time.perf_counter()-t; print(f"create+write+close {n} files, {nt} python threads: {dt:.2f}s = {dt/n*1e6:.1f} us/file")
    shutil.rmtree(dd)
