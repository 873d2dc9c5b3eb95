import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers as H
from test_hip_batch import _setup, dev
from dflash_amd.batch import BatchedDecoder
from dflash_amd.generate import DecodeSession
cfg, m, hf, nt, perm = _setup()
lens = (45, 23, 70)
prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(7 + i)).to(dev()) for i, P in enumerate(lens)]
dec = BatchedDecoder(m, nt, 3, max_rows=200, out_len=200, mask_token_id=cfg.mask_token_id)
sess = []
for r, p in enumerate(prompts):
    dec.admit(r, p)
    s = DecodeSession(m, nt, p, mask_token_id=cfg.mask_token_id, max_new_tokens=100, max_block_size=16, stop_token_ids=None, temperature=0.0)
    s.prefill(); sess.append(s)
for cyc in range(3):
    dec.draft(); dec.verify(); res = dec.accept()
    for r, s in enumerate(sess):
        start = s.start
        out = s.cycle(16)
        print(f"cyc {cyc} r {r} start {start} tau single {out.tau} batch {res[r][0]} dyn_d {dec.dyn_d[r].tolist()}")
        for li in range(cfg.num_hidden_layers):
            for nm, a, b in (("k", dec.dk[r, li][:, :start], s.dcache.k[li][:, :start]), ("v", dec.dv[r, li][:, :start], s.dcache.v[li][:, :start])):
                dd = (a.float() - b.float()).abs().amax(dim=(0, 2))
                bad = (dd > 0.1).nonzero().flatten().tolist()
                if bad: print("   layer", li, nm, "bad rows", bad, "a zero?", [float(a[:, i].abs().max()) for i in bad], "b", [float(b[:, i].abs().max()) for i in bad])
