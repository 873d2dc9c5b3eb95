import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from dflash_amd import ops
from dflash_amd.model import _rope_tables
from oracle import dflash_oracle as O
BF16 = torch.bfloat16
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(9)
n_q, n_kv = 8, 2
ld = (n_q + 2 * n_kv) * 128
part = torch.randn(1, 32, ld, generator=g)
ones = torch.ones(128, dtype=BF16)
qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16)
cos, sin = _rope_tables(128, 1e6, 256, dev)

def run(S, tau, bs, w, pos0):
    dyn = torch.zeros(8, dtype=torch.int32, device=dev)
    ops.set_dyn(dyn, S, tau, bs, pos0)
    kc = torch.zeros(n_kv, 128, 128, dtype=BF16, device=dev); vc = torch.zeros_like(kc)
    q_out = torch.zeros(n_q, 16, 128, dtype=BF16, device=dev)
    ops.qknorm_rope_append(qkv=part.to(dev), nsplit=1, split_stride=32 * ld, ld=ld, q_col=0, k_col=n_q * 128,
                           v_col=(n_q + n_kv) * 128, ctx_row0=0, blk_row0=16, n_q=n_q, n_kv=n_kv,
                           q_norm_w=w.to(dev), k_norm_w=w.to(dev), eps=1e-6, cos_tab=cos, sin_tab=sin,
                           q_out=q_out, kcache=kc, vcache=vc, dyn=dyn)
    return kc.cpu(), q_out.cpu()

lin = part[0].to(BF16)
# exp 1: one ctx row at position 0 (rope = identity), unit weights -> pure rms norm
kc, _ = run(0, 1, 0, ones, 0)
k = lin[:1, n_q*128:(n_q+n_kv)*128].view(1, n_kv, 128)
ref = O.rms_norm(k, ones, 1e-6)
print('norm only   mismatch', (kc[:, 0] != ref[0]).float().mean().item())
hf = k.float(); var = hf.pow(2).mean(-1, keepdim=True)
print('  rstd torch', torch.rsqrt(var + 1e-6).flatten().tolist())
# exp 2: weights non-unit
kc, _ = run(0, 1, 0, qw, 0)
ref = O.rms_norm(k, qw, 1e-6)
print('norm+weight mismatch', (kc[:, 0] != ref[0]).float().mean().item())
# exp 3: position 5, unit weights
kc, _ = run(5, 1, 0, ones, 5)
kn = O.rms_norm(k, ones, 1e-6).view(1, 1, n_kv, 128).transpose(1, 2)
c, s = O.rope_cos_sin(torch.tensor([[5]]), O.rope_inv_freq(128, 1e6), BF16)
_, kr = O.apply_rotary_dflash(kn[:, :, :0], kn, c, s)
print('rope pos5   mismatch', (kc[:, 5] != kr[0, :, 0]).float().mean().item())
print('table cos row5 equal', torch.equal(cos[5].cpu(), c[0, 0, :64]), torch.equal(sin[5].cpu(), s[0, 0, :64]))
d = (kc[:, 5].float() - kr[0, :, 0].float())
idx = d.nonzero()
print('first mismatches', idx[:10].tolist())
for h, dd in idx[:6].tolist():
    print(h, dd, 'got', kc[h, 5, dd].item(), 'ref', kr[0, h, 0, dd].item(), 'n', kn[0, h, 0, dd].item(), 'pair', kn[0, h, 0, (dd + 64) % 128].item(), 'cos', c[0, 0, dd].item(), 'sin', s[0, 0, dd].item())
