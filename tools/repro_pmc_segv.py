"""Diagnostic (profiles/README.md, "rocprofv3 --pmc SIGSEGV"): the launch pattern of the first gen_gpu._lcg — ~8,000
rounds of three small int64 torch launches — under `rocprofv3 --pmc ...`, with the process's library map written out
first so that the frames of the crash can be attributed (profiler SDK vs HIP runtime vs torch).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/segv -- python3 tools/repro_pmc_segv.py"""
import os, sys, faulthandler
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
faulthandler.enable(open(os.path.join(out, "segv_py_traceback.txt"), "w"))
x = torch.arange(1 << 16, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
open(os.path.join(out, "segv_maps.txt"), "w").write(open("/proc/self/maps").read())
A = torch.tensor(6364136223846793005, dtype=torch.int64, device="cuda")
C = torch.tensor(1442695040888963407, dtype=torch.int64, device="cuda")
outb = torch.empty(500 << 20, dtype=torch.uint8, device="cuda")
pos, n = 0, 1 << 16
rounds = int(os.environ.get("ROUNDS", 8000))
for i in range(rounds):
    outb[pos:pos + n] = ((x >> 33) & 0xFF).to(torch.uint8)
    x = x * A + C
    pos += n
    if i % 1000 == 0:
        print("round", i, flush=True)
torch.cuda.synchronize()
print("completed", rounds, "rounds without a crash")
