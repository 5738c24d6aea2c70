#!/usr/bin/env python
"""ds_read_b64_tr_b16 semantics probe (run on the GPU box): prints, for a few per-lane address patterns, the LDS element index
every (lane, slot) receives.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/debug/tr_probe.hip -o tools/debug/libtr_probe.so"""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libtr_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "tr_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.tr_probe.argtypes = [C.c_void_p, C.c_void_p]
def run(tag, addr):
    a = torch.tensor(addr, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = lib.tr_probe(a.data_ptr(), out.data_ptr())
    o = out.cpu().view(64, 4).tolist()
    print(f"== {tag} (rc {rc}); element index received by lane: slot0..3")
    for l in range(64):
        if l < 20 or l % 16 == 0:
            print(f"  lane {l:2d} addr {addr[l]:5d}B -> {o[l]}")
lanes = list(range(64))
run("contiguous 8 B per lane (4x16 row-major blocks of 128 B per 16-lane group)", [l * 8 for l in lanes])
run("row stride 256 B: lane i -> row i//4 (256 B apart), 8-byte piece i%4; groups 1024 B apart", [(l % 16 // 4) * 256 + (l % 4) * 8 + (l // 16) * 1024 for l in lanes])
run("lane i -> row i%4 (256 B apart), piece i//4 (transposed assignment)", [(l % 4) * 256 + (l % 16 // 4) * 8 + (l // 16) * 1024 for l in lanes])
