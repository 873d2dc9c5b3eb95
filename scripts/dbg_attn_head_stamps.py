"""Diagnostic (GPU box): build attn_head.hip with -DDFL_ATTN_STAMPS into gpurun_out/dbg and print where a
k_attn_head workgroup spends its time (100 MHz s_memrealtime stamps: first old-key split and the new-row split
of kv head 0 / query head 0), for the split knobs given in the environment (DFL_ATTN_HEAD_TILES / _WGS).
K/V are cold (a 400 MB fill runs before every launch), as in the decode cycle."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libdbg_head.so")
src = [os.path.join(ROOT, "dflash_amd", "csrc", f) for f in ("attn_head.hip", "dfl_common.hip")]
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DDFL_ATTN_STAMPS",
                *flags, "-o", so, *src], check=True)
from dflash_amd import _lib, ops
from dflash_amd.model import _rope_tables
dbg = C.CDLL(so)
fn = dbg.dfl_attn_head
fn.restype, fn.argtypes = _lib.SIGNATURES["dfl_attn_head"]
dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
n_q, n_kv, S, tau, bs = 32, 8, int(os.environ.get("S", "1040")), int(os.environ.get("TAU", "0")), 16
causal = 1 if tau == 0 else 0
ld = (n_q + 2 * n_kv) * 128
x = torch.randn(32, ld, device=dev).to(BF16)
qw = torch.ones(128, dtype=BF16, device=dev)
cos, sin = _rope_tables(128, 1e6, S + 2048, dev)
k = torch.randn(n_kv, S + 1024, 128, device=dev).to(BF16)
v = torch.randn(n_kv, S + 1024, 128, device=dev).to(BF16)
ws = ops.attn_head_ws(n_q, 32, 1, dev)
outf = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev)
big = torch.empty(400_000_000, dtype=torch.uint8, device=dev)
names = ["start", "prologue", "bar+qf+tiles", "wait others", "O->LDS+bar+merge", "publish+drain+bar", "ticket+bar", "final merge"]
tot = []
for rep in range(5):
    if os.environ.get("COLD", "1") == "1":   # COLD=0: K/V left in L2 by the previous launch (upper bound of a prefetch)
        big.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(x[16:].data_ptr(), ld, 0, n_q * 128, (n_q + n_kv) * 128, x.data_ptr(), ld, n_q * 128, (n_q + n_kv) * 128, n_q,
            n_kv, qw.data_ptr(), qw.data_ptr(), 1e-6, cos.data_ptr(), sin.data_ptr(), cos.shape[0], k.data_ptr(),
            v.data_ptr(), k.shape[1], 128 ** -0.5, causal, None, S, tau, bs, S, 1, ws.data_ptr(), 32, outf.data_ptr(), 0,
            None)
    assert rc == 0
    e1.record()
    torch.cuda.synchronize()
    tot.append(e0.elapsed_time(e1) * 1e3)
    st = (C.c_ulonglong * 16)()
    assert dbg.dfl_debug_read_head_stamps(st) == 0
    if rep:
        for w, label in ((0, "old split 0"), (1, "new split  ")):
            t = [st[w * 8 + i] for i in range(8)]
            d = [(t[i + 1] - t[i]) / 100.0 if t[i + 1] >= t[i] > 0 else float("nan") for i in range(7)]
            print(f"S={S} tau={tau} rep {rep} {label}: " + "  ".join(f"{n}={x_:.2f}" for n, x_ in zip(names[1:], d))
                  + f"  | total {(max(t) - t[0]) / 100.0:.2f} us")
print(f"S={S} tau={tau} TILES={os.environ.get('DFL_ATTN_HEAD_TILES', '8')} WGS={os.environ.get('DFL_ATTN_HEAD_WGS', '224')}: "
      f"event-to-event {sorted(tot)[len(tot) // 2]:.1f} us (median of {len(tot)})")
