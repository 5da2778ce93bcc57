import os, sys, torch, numpy as np
os.environ["NNOP_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nnop.jl_amd/lib_abl9/libnnop_hip.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
B, H, L, E = 4, 4, 4096, 64
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda: torch.randn(B, H, L, E, generator=g, device=dev).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
o = torch.empty_like(q); ms = torch.empty(B, H, L, dtype=torch.bfloat16, device=dev); ls = torch.empty_like(ms)
for _ in range(30): pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False)
torch.cuda.synchronize()
raw = o.view(torch.int64).reshape(B * H * L, E // 4).cpu().numpy()      # 16 int64 per row
rows = raw[::32]                                                        # first row of every wave
tot, rt, a, b, c, d, e, nt = (rows[:, i].astype(np.float64) for i in range(8))
n = nt.mean()
print(f"waves {len(rows)}  tiles {n:.0f}")
print(f"in-kernel clock: {np.median(tot / rt) * 100:.0f} MHz   wave lifetime {np.median(tot):.0f} cyc = {np.median(rt) / 100:.1f} us")
print(f"per interval (cycles, median over waves): total loop part {np.median((c + d + e) / n):.0f}")
print(f"  issue staging loads + frag loads + wait all frags : {np.median(a / n):.0f}")
print(f"  compute block (MFMA + softmax)                    : {np.median(b / n):.0f}")
print(f"  up to staging writes (incl. row max of next tile) : {np.median(c / n):.0f}")
print(f"  vmcnt wait + LDS writes                           : {np.median(d / n):.0f}")
print(f"  barrier                                           : {np.median(e / n):.0f}")
