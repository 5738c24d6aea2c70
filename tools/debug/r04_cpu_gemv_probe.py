"""Why is the CPU baseline's decode step flat in the thread count?  Times the M = 1 fp32 linear of LLaMA-7B's gate/up shape in several forms
and thread counts on this host (run by tools/debug/r04_run7.sh; output -> profiles/r04_cpu_baseline_sweep.txt)."""
import os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "bind":
    os.environ["OMP_PROC_BIND"] = "close"; os.environ["OMP_PLACES"] = "cores"
if mode == "spread":
    os.environ["OMP_PROC_BIND"] = "spread"; os.environ["OMP_PLACES"] = "cores"
if mode in ("node0", "node0_bind"):
    def parse(l):
        out = []
        for p in l.strip().split(","):
            if "-" in p: a, b = p.split("-"); out += list(range(int(a), int(b) + 1))
            elif p: out.append(int(p))
        return out
    os.sched_setaffinity(0, parse(open("/sys/devices/system/node/node0/cpulist").read())[:64])
    if mode == "node0_bind":
        os.environ["OMP_PROC_BIND"] = "close"
import torch
import torch.nn.functional as F
if mode == "default":
    cfgs = torch.__config__.show()
    print("\n".join(l for l in cfgs.splitlines() if any(k in l for k in ("BLAS", "LAPACK", "OpenMP", "MKL", "oneDNN", "Build settings"))) [:1500])
    print("nodes:", sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")))
    print("parallel info:", torch.__config__.parallel_info().replace("\n", " | ")[:600])
N, K = 22016, 4096
W = torch.randn(N, K) * 0.02
Ws = [W.clone() for _ in range(4)]        # 1.4 GB working set: not cache-resident
x1 = torch.randn(1, K); xv = torch.randn(K); x8 = torch.randn(8, K)
by = N * K * 4
def t(fn, reps=6):
    fn(Ws[0]); t0 = time.time()
    for i in range(reps): fn(Ws[i % 4])
    return (time.time() - t0) / reps
for n in (1, 8, 16, 32, 64):
    torch.set_num_threads(n)
    a = t(lambda w: F.linear(x1, w)); b = t(lambda w: torch.mv(w, xv)); c = t(lambda w: F.linear(x8, w)); d = t(lambda w: (w * xv).sum(1))
    print(f"[{mode:10s}] threads={n:3d}  F.linear(M=1) {by/a/1e9:6.1f} GB/s | torch.mv {by/b/1e9:6.1f} | F.linear(M=8) {by/c/1e9:6.1f} | (w*x).sum(1) {by/d/1e9:6.1f}")
