import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
for dt in (torch.bfloat16, torch.float16):
  for (B, H, L, E) in [(4, 4, 4096, 64), (2, 2, 1024, 32), (2, 2, 512, 16)]:
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda: torch.randn(B, H, L, E, generator=g, device=dev).to(dt)
    q, k, v = mk(), mk(), mk()
    outs = {}
    for cfg in [("8", "1"), ("4", "2"), ("4", "2"), ("4", "1")]:
        os.environ["NNOP_FWD_NW"], os.environ["NNOP_FWD_QB"] = cfg
        o, ms, ls = pkg._flash_attention(q, k, v, causal=False)
        torch.cuda.synchronize()
        outs.setdefault(cfg, []).append((o, ms, ls))
    ref = outs[("8", "1")][0]
    ok = [all(torch.equal(a, b) for a, b in zip(ref, x)) for x in outs[("4", "2")] + outs[("4", "1")]]
    print(dt, (B, H, L, E), "QB=2 run1, QB=2 run2, NW4QB1 bitwise == NW8QB1:", ok, flush=True)
