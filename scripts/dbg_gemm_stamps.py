"""Diagnostic (GPU box): build the GEMM with -DDFL_GEMM_STAMPS into gpurun_out/dbg and print where workgroup 0 of a
single-request k_gemm spends its time (100 MHz s_memrealtime stamps), per shape and row source."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libdbg_gemm.so")
src = [os.path.join(ROOT, "dflash_amd", "csrc", f) for f in ("gemm_skinny.hip", "dfl_common.hip", "rows.hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DDFL_GEMM_STAMPS",
                "-o", so, *src], check=True)
from dflash_amd import _lib, ops
dbg = C.CDLL(so)
for n in ("dfl_gemm_f32", "dfl_gemm_resid", "dfl_gemm_silu_mul", "dfl_set_dyn"):
    getattr(dbg, n).restype, getattr(dbg, n).argtypes = _lib.SIGNATURES[n]
dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
H, I = 4096, 12288
dyn = torch.zeros(8, dtype=torch.int32, device=dev)
ops.set_dyn(dyn, 0, 16, 16, 0)
h = torch.randn(16, H, device=dev).to(BF16)
nw = torch.ones(H, device=dev, dtype=BF16)
ss = torch.rand(256 * 16, device=dev) * 16
normed = ops.rows_normed(h, ss, 256, nw, 1e-6, ops.DYN_BS)
fragH = ops.rows_frag(torch.randn(16 * H, device=dev).to(BF16))
fragI = ops.rows_frag(torch.randn(16 * I, device=dev).to(BF16))
big = torch.empty(600_000_000, dtype=torch.uint8, device=dev)
names = ["prologue loads issued", "lengths + rstd", "build x", "first item done", "loop", "tail"]


def run(label, fn):
    for rep in range(3):
        big.zero_()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        assert fn() == 0
        t1.record()
        torch.cuda.synchronize()
        st = (C.c_ulonglong * 8)()
        assert dbg.dfl_debug_read_gemm_stamps(st) == 0
        t = [st[i] for i in range(6)]
        d = [(t[i + 1] - t[i]) / 100.0 for i in range(5)]
        print(f"{label:28s} rep {rep}: first weights asked={(st[6] - st[0]) / 100.0:.2f}  " + "  ".join(f"{n}={x:.2f}" for n, x in zip(names, d)) +
              f"  | wg0 total {(t[5] - t[0]) / 100.0:.2f} us, events {t0.elapsed_time(t1) * 1e3:.1f} us")


S = torch.cuda.current_stream().cuda_stream
wq = torch.randn(6144 * H // 2, device=dev).view(BF16)[:6144 * H].contiguous()
part = torch.empty(2 * 16 * 6144, device=dev)
run("qkv f32 normed ksplit2", lambda: dbg.dfl_gemm_f32(wq.data_ptr(), normed.ref, None, 1, 6144, H, 2, part.data_ptr(), dyn.data_ptr(), S))
run("qkv f32 frag   ksplit2", lambda: dbg.dfl_gemm_f32(wq.data_ptr(), fragH.ref, None, 1, 6144, H, 2, part.data_ptr(), dyn.data_ptr(), S))
wo = torch.randn(H * H // 2, device=dev).view(BF16)[:H * H].contiguous()
hh = torch.zeros(16, H, device=dev, dtype=BF16)
sso = torch.zeros(H, device=dev)
run("o resid frag", lambda: dbg.dfl_gemm_resid(wo.data_ptr(), fragH.ref, H, H, hh.data_ptr(), H, 1, None, 0, sso.data_ptr(), dyn.data_ptr(), S))
wg = torch.randn(2 * I * H // 2, device=dev).view(BF16)[:2 * I * H].contiguous()
act = torch.empty(16 * I, device=dev, dtype=BF16)
run("gate/up silu normed", lambda: dbg.dfl_gemm_silu_mul(wg.data_ptr(), normed.ref, I, H, act.data_ptr(), dyn.data_ptr(), S))
run("gate/up silu frag", lambda: dbg.dfl_gemm_silu_mul(wg.data_ptr(), fragH.ref, I, H, act.data_ptr(), dyn.data_ptr(), S))
wd = torch.randn(H * I // 2, device=dev).view(BF16)[:H * I].contiguous()
run("down resid frag (chunked)", lambda: dbg.dfl_gemm_resid(wd.data_ptr(), fragI.ref, H, I, hh.data_ptr(), H, 1, None, 0, sso.data_ptr(), dyn.data_ptr(), S))
