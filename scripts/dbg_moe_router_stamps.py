"""Diagnostic (GPU box): build moe.hip with -DDFL_MOE_STAMPS into gpurun_out/dbg and print where the workgroups of
k_moe_router spend their time (100 MHz s_memrealtime stamps) at the 30B-A3B router shape (K 2048, E 128, top-8)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libdbg_moe.so")
src = [os.path.join(ROOT, "dflash_amd", "csrc", f) for f in ("moe.hip", "dfl_common.hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DDFL_MOE_STAMPS",
                "-I", os.path.join(ROOT, "include"), "-o", so, *src], check=True)
from dflash_amd import _lib, ops
dbg = C.CDLL(so)
dbg.dfl_moe_router.restype, dbg.dfl_moe_router.argtypes = _lib.SIGNATURES["dfl_moe_router"]
dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
K, E, TOPK = 2048, 128, 8
g = torch.Generator().manual_seed(1)
wp = ops.pack_weight((torch.randn(E, K, generator=g) * 0.3).to(BF16).to(dev))
nw = torch.ones(K, dtype=BF16, device=dev)
h = torch.randn(16, K, generator=g).to(BF16).to(dev)
xn = torch.zeros(16 * K, dtype=BF16, device=dev)
rlog = torch.zeros(16, E, dtype=BF16, device=dev)
wt = torch.zeros(16, E, dtype=BF16, device=dev)
active = torch.zeros(E, dtype=torch.int32, device=dev)
lst = torch.zeros(E, dtype=torch.int32, device=dev)
n = torch.zeros(1, dtype=torch.int32, device=dev)
ticket = torch.zeros(1, dtype=torch.int32, device=dev)
big = torch.empty(600_000_000, dtype=torch.uint8, device=dev)
names = ["rows + weights landed, ss met", "normalise + MFMA + LDS meet", "tile rounded", "logits stored sc1 + drained",
         "ticket returned", "(last) logits loaded to LDS", "flags zeroed + barrier", "softmax", "k rounds", "weights written + barrier", "list written", ]
for rep in range(4):
    big.zero_()     # cold caches, as behind the layer's o_proj
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    rc = dbg.dfl_moe_router(h.data_ptr(), h.stride(0), nw.data_ptr(), 1e-6, xn.data_ptr(), wp.data_ptr(), K, E, TOPK, 1, rlog.data_ptr(),
                            rlog.stride(0), wt.data_ptr(), active.data_ptr(), lst.data_ptr(), n.data_ptr(), None, 0, ticket.data_ptr(),
                            None)
    t1.record()
    torch.cuda.synchronize()
    assert rc == 0
    st = (C.c_ulonglong * (16 * 12))()
    assert dbg.dfl_debug_read_router_stamps(st) == 0
    if rep == 0:
        continue
    rows = [[st[b * 12 + i] for i in range(12)] for b in range(8)]
    t00 = min(r[0] for r in rows)
    print(f"rep {rep}: events around the launch {t0.elapsed_time(t1) * 1e3:.1f} us; n_active {int(n)}")
    for b, r in enumerate(rows):
        last = r[6] > r[0]
        seq = [f"{(r[i] - t00) / 100.0:6.2f}" for i in range(6)] + ([f"{(r[i] - t00) / 100.0:6.2f}" for i in (6, 8, 9, 10, 11, 7)] if last else [])
        print(f"  wg {b}: " + " ".join(seq) + ("   <- last arriver" if last else ""))
print("columns (us from the first workgroup's start): start | " + " | ".join(names))
