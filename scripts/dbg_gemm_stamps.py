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
def dbg_zero():
    pass


names = ["prologue loads issued", "lengths + rstd", "build x", "first item done", "loop", "tail"]


def run(label, fn):
    for rep in range(3):
        big.zero_()
        dbg_zero()
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
        if rep == 2:
            wg = (C.c_ulonglong * 1024)()
            assert dbg.dfl_debug_read_wg_stamps(wg) == 0
            rows = [(wg[4 * b], wg[4 * b + 1], wg[4 * b + 2]) for b in range(256) if wg[4 * b]]
            t00 = min(r[0] for r in rows)
            for name, k in (("start", 0), ("first item done", 1), ("end", 2)):
                v = sorted((r[k] - t00) / 100.0 for r in rows)
                print(f"    all {len(rows)} workgroups, {name:16s}: min {v[0]:6.2f}  p10 {v[len(v) // 10]:6.2f}  median {v[len(v) // 2]:6.2f}  "
                      f"p90 {v[9 * len(v) // 10]:6.2f}  max {v[-1]:6.2f} us after the first start")
            late = sorted(range(len(rows)), key=lambda b: -rows[b][2])[:12]
            print("    latest workgroups (block: end us, start us): " + "  ".join(f"{b}: {(rows[b][2] - t00) / 100.0:.1f}, {(rows[b][0] - t00) / 100.0:.2f}" for b in late))
            byx = [[] for _ in range(8)]
            for b, r in enumerate(rows):
                byx[b % 8].append((r[2] - t00) / 100.0)
            print("    mean end by block % 8: " + "  ".join(f"{sum(v) / len(v):.2f}" for v in byx if v))
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
