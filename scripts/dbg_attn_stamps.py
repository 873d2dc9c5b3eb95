"""Diagnostic (GPU box): build attn_block.hip with -DDFL_ATTN_STAMPS into gpurun_out/dbg and print where a
k_attn_fused workgroup spends its time (100 MHz s_memrealtime stamps; first and last key split of kv head 0)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libdbg.so")
src = [os.path.join(ROOT, "dflash_amd", "csrc", f) for f in ("attn_block.hip", "dfl_common.hip", "rows.hip")]
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DDFL_ATTN_STAMPS",
                *flags, "-o", so, *src], check=True)
from dflash_amd import _lib, ops
from dflash_amd.model import _rope_tables
dbg = C.CDLL(so)
name = "dfl_attn_fused"
fn = getattr(dbg, name)
fn.restype, fn.argtypes = _lib.SIGNATURES[name]
dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
n_q, n_kv, S, tau, bs = 32, 8, int(os.environ.get("S", "1100")), 0, 16
ld = (n_q + 2 * n_kv) * 128
part = torch.randn(2, 16, ld, device=dev)
qw = torch.ones(128, dtype=BF16, device=dev)
cos, sin = _rope_tables(128, 1e6, S + 2048, dev)
k = torch.randn(n_kv, S + 1024, 128, device=dev).to(BF16)
v = torch.randn(n_kv, S + 1024, 128, device=dev).to(BF16)
dyn = torch.zeros(8, dtype=torch.int32, device=dev)
ops.set_dyn(dyn, S, tau, bs, S)
ws = ops.attn_fused_ws(n_q, n_kv, 32, dev)
outf = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev)
big = torch.empty(400_000_000, dtype=torch.uint8, device=dev)
names = ["start->ph0", "ph0 (q rope)", "ph1 (kv rope)+bar", "tile loop", "partials+wait", "fence+ticket", "merge"]
for rep in range(4):
    big.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(part.data_ptr(), 2, 16 * ld, ld, 0, n_q * 128, (n_q + n_kv) * 128, 0, 0, n_q, n_kv, qw.data_ptr(), qw.data_ptr(),
            1e-6, cos.data_ptr(), sin.data_ptr(), cos.shape[0], k.data_ptr(), v.data_ptr(), k.shape[1], 128 ** -0.5, 1,
            dyn.data_ptr(), S + tau + bs, ws.data_ptr(), 32, outf.data_ptr(), None)
    assert rc == 0
    e1.record()
    torch.cuda.synchronize()
    print(f"rep {rep} S={S}: launch-to-end {e0.elapsed_time(e1) * 1e3:.1f} us")
    st = (C.c_ulonglong * 16)()
    assert dbg.dfl_debug_read_stamps(st) == 0
    for w, label in ((0, "first split"), (1, "last split ")):
        t = [st[w * 8 + i] for i in range(7)]
        d = [(t[i + 1] - t[i]) / 100.0 if t[i + 1] >= t[i] > 0 else float("nan") for i in range(6)]
        print(f"rep {rep} {label}: " + "  ".join(f"{n}={x:.2f}us" for n, x in zip(names[1:], d)))
