"""Diagnostic (GPU box): build attn_head.hip with -DDFL_ATTN_STAMPS into gpurun_out/dbg and print the timeline of one
k_attn_oproj launch (100 MHz s_memrealtime stamps): attention workgroup (kv head 0, query head 0) of the first old-key
split and of the new-row split, and the first / last o_proj workgroup, all relative to the earliest stamp.
Knobs: DFL_ATTN_OPROJ_WGS / DFL_ATTN_OPROJ_TILES (environment), S, TAU.  K/V and weights are cold (a 400 MB fill runs
before every launch), as in the decode cycle."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libdbg_oproj.so")
src = [os.path.join(ROOT, "dflash_amd", "csrc", f) for f in ("attn_head.hip", "dfl_common.hip")]
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DDFL_ATTN_STAMPS",
                *flags, "-o", so, *src], check=True)
from dflash_amd import _lib, ops
from dflash_amd.model import _rope_tables
dbg = C.CDLL(so)
fn = dbg.dfl_attn_head_oproj
fn.restype, fn.argtypes = _lib.SIGNATURES["dfl_attn_head_oproj"]
dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
n_q, n_kv, H, S, tau, bs = 32, 8, 4096, int(os.environ.get("S", "1040")), int(os.environ.get("TAU", "0")), 16
causal = 1 if tau == 0 else 0
ld = (n_q + 2 * n_kv) * 128
x = torch.randn(32, ld, device=dev).to(BF16)
qw = torch.ones(128, dtype=BF16, device=dev)
cos, sin = _rope_tables(128, 1e6, S + 2048, dev)
k = torch.randn(n_kv, S + 1024, 128, device=dev).to(BF16)
v = torch.randn(n_kv, S + 1024, 128, device=dev).to(BF16)
wo = ops.pack_weight((torch.randn(H, n_q * 128, device=dev) * 0.02).to(BF16))
h = torch.zeros(16, H, dtype=BF16, device=dev)
ss = torch.zeros(H, dtype=torch.float32, device=dev)
sync = torch.zeros(ops.ATTN_OPROJ_SYNC_WORDS, dtype=torch.int32, device=dev)
ws = ops.attn_head_ws(n_q, 32, 1, dev)
outf = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev)
big = torch.empty(400_000_000, dtype=torch.uint8, device=dev)
tot = []
for rep in range(5):
    big.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(x[16:].data_ptr(), ld, 0, n_q * 128, (n_q + n_kv) * 128, x.data_ptr(), ld, n_q * 128, (n_q + n_kv) * 128, n_q,
            n_kv, qw.data_ptr(), qw.data_ptr(), 1e-6, cos.data_ptr(), sin.data_ptr(), cos.shape[0], k.data_ptr(),
            v.data_ptr(), k.shape[1], 128 ** -0.5, causal, None, S, tau, bs, S, ws.data_ptr(), 32, outf.data_ptr(),
            wo.data_ptr(), H, h.data_ptr(), H, ss.data_ptr(), sync.data_ptr(), None)
    assert rc == 0
    e1.record()
    torch.cuda.synchronize()
    assert int(sync.abs().sum()) == 0, sync.nonzero().tolist()
    tot.append(e0.elapsed_time(e1) * 1e3)
    hs, os_ = (C.c_ulonglong * 16)(), (C.c_ulonglong * 16)()
    assert dbg.dfl_debug_read_head_stamps(hs) == 0 and dbg.dfl_debug_read_oproj_stamps(os_) == 0
    am = (C.c_ulonglong * 16)()
    assert dbg.dfl_debug_read_attn_max(am) == 0
    if rep:
        allv = [t for t in list(hs) + list(os_) if t > 0]
        z = min(allv)
        f = lambda t: f"{(t - z) / 100.0:6.2f}" if t > 0 else "   nan"  # noqa: E731
        print(f"rep {rep}: attention old split 0 [start prologue tiles wait merge publish ticket final]: " + " ".join(f(hs[i]) for i in range(8)))
        print(f"rep {rep}: attention new split   [                                                    ]: " + " ".join(f(hs[8 + i]) for i in range(8)))
        print(f"rep {rep}: latest ticket per key split: " + " ".join(f(am[i]) for i in range(15)) + f" | latest head-done signal {f(am[15])}")
        for j, lab in ((0, "first"), (1, "last ")):
            print(f"rep {rep}: o_proj {lab} [start issued landed heads-done mfma-done barrier end]: " + " ".join(f(os_[8 * j + i]) for i in range(7)))
print(f"S={S} tau={tau} WGS={os.environ.get('DFL_ATTN_OPROJ_WGS', '256')} TILES={os.environ.get('DFL_ATTN_OPROJ_TILES', '4')}: "
      f"event-to-event {sorted(tot)[len(tot) // 2]:.1f} us (median of {len(tot)})")
