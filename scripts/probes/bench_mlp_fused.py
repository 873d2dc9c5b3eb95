"""EXPERIMENT (GPU box): gate/up + down as one cooperative launch (dfl_mlp_fused) against the two
product launches (dfl_gemm_silu_mul + dfl_gemm_resid): same bits, device time per MLP with the
weights rotated through > 256 MiB."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from dflash_amd import _lib, ops

out = os.path.join(ROOT, "gpurun_out", "dbg")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libexp_mlp.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DDFL_EXPERIMENTAL_MLP",
                "-o", so, os.path.join(ROOT, "dflash_amd", "csrc", "gemm_skinny.hip"),
                os.path.join(ROOT, "dflash_amd", "csrc", "dfl_common.hip")], check=True)
exp = C.CDLL(so)
_p, _i = C.c_void_p, C.c_int
exp.dfl_mlp_fused.restype = _i
exp.dfl_mlp_fused.argtypes = [_p, C.POINTER(_lib.Rows), _i, _i, _p, _p, _i, _p, C.c_int64, _p, _p, _p, C.c_uint, _p]


def mlp_fused(wp_gu, x, I, K, act, wp_down, N, h_io, ss_out, dyn, bar_ws, epoch):
    rc = exp.dfl_mlp_fused(wp_gu.data_ptr(), x.ref, I, K, act.data_ptr(), wp_down.data_ptr(), N, h_io.data_ptr(),
                           h_io.stride(0), ss_out.data_ptr(), dyn.data_ptr(), bar_ws.data_ptr(), epoch,
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
H, I = 4096, 12288
nbuf = 3
g = torch.Generator(device=dev).manual_seed(0)
gus = [ops.pack_weight_gateup((torch.randn(I, H, device=dev, generator=g) * 0.02).to(BF16),
                              (torch.randn(I, H, device=dev, generator=g) * 0.02).to(BF16)) for _ in range(nbuf)]
downs = [ops.pack_weight((torch.randn(H, I, device=dev, generator=g) * 0.02).to(BF16)) for _ in range(nbuf)]
h0 = torch.randn(16, H, device=dev, generator=g).to(BF16)
nw = (1 + 0.1 * torch.randn(H, device=dev, generator=g)).to(BF16)
ss_in = h0.float().pow(2).view(16, H // 16, 16).sum(-1).T.contiguous().view(-1)      # [tile][row]
dyn = torch.zeros(8, dtype=torch.int32, device=dev)
ops.set_dyn(dyn, 0, 16, 16, 0)
src = ops.rows_normed(h0, ss_in, H // 16, nw, 1e-6, ops.DYN_BS)
bar = torch.zeros(4096, dtype=torch.uint8, device=dev)
epoch = [0]


def separate(i, h, act, ss):
    ops.gemm_silu_mul(gus[i], src, I, H, act, dyn)
    ops.gemm_resid(downs[i], ops.rows_frag(act), H, I, h, add_residual=True, ss_out=ss, dyn=dyn)


def fused(i, h, act, ss):
    epoch[0] += 1
    mlp_fused(gus[i], src, I, H, act, downs[i], H, h, ss, dyn, bar, epoch[0])


def fused_plain(i, h, act, ss):
    epoch[0] += 1
    mlp_fused(gus[i], src, I, H, act, downs[i], H, h, ss, dyn, bar, epoch[0] | 0x80000000)


outs = []
for fn in (separate, fused, fused_plain):
    h = h0.clone()
    act = torch.zeros(16 * I, dtype=BF16, device=dev)
    ss = torch.zeros(H, device=dev)
    # the normed source points at h0 (not h), so both variants read identical inputs
    fn(0, h, act, ss)
    torch.cuda.synchronize()
    outs.append((h.clone(), act.clone(), ss.clone()))
print("barrier gave up:", int(bar[2048:2052].view(torch.int32)[0]))
for v in (1, 2):
    for name, a, b in zip(("h", "act", "ss"), outs[0], outs[v]):
        print(f"variant {v} {name}: identical = {torch.equal(a, b)}  max|d| = {float((a.float() - b.float()).abs().max()):.3g}")

h = h0.clone()
act = torch.zeros(16 * I, dtype=BF16, device=dev)
ss = torch.zeros(H, device=dev)
for name, fn in (("two launches", separate), ("one cooperative launch", fused), ("one plain launch", fused_plain)):
    for i in range(nbuf):
        fn(i, h, act, ss)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(nbuf):
            fn(i, h, act, ss)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / nbuf * 1e3)
    print(f"{name:24s}: {best:7.1f} us per MLP ({(2 * I * H + H * I) * 2 / best / 1e6:.2f} TB/s)")
print("barrier gave up:", int(bar[2048:2052].view(torch.int32)[0]))
