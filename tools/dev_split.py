import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
for dt in (torch.bfloat16, torch.float16):
  for (B, H, L, KL, E) in [(4, 4, 4096, 4096, 64), (2, 2, 1024, 1024, 32), (2, 2, 512, 576, 16), (1, 2, 300, 192, 64)]:
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B, H, L, E, generator=g, device=dev).to(dt)
    k = torch.randn(B, H, KL, E, generator=g, device=dev).to(dt); v = torch.randn(B, H, KL, E, generator=g, device=dev).to(dt)
    os.environ["NNOP_FWD_SPLIT"] = "0"
    ref = pkg._flash_attention(q, k, v, causal=False)
    os.environ["NNOP_FWD_SPLIT"] = "1"
    a = pkg._flash_attention(q, k, v, causal=False); b2 = pkg._flash_attention(q, k, v, causal=False)
    torch.cuda.synchronize()
    rel = lambda x, y: float((x.float() - y.float()).abs().max() / y.float().abs().max())
    print(dt, (B, H, L, KL, E), "o", f"{rel(a[0], ref[0]):.2e}", "ms", f"{rel(a[1], ref[1]):.2e}", "ls", f"{rel(a[2], ref[2]):.2e}",
          "deterministic", all(torch.equal(x, y) for x, y in zip(a, b2)), flush=True)
