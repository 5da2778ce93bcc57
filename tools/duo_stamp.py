"""dev: in-kernel time stamps of the two-waves-per-SIMD forward (library built with `make DEV=1 VAR=-DNNOP_DUO_STAMP=1 OUTDIR=../lib_stamp
BUILD=../build_stamp`, passed through NNOP_LIB_PATH): prologue / loop / epilogue per workgroup in shader cycles, cycles per kv tile pair
(= per half-step pair: one tile of either key group), where wave 0 spends them (matrix phase, barrier, vector phase, DMA wait, barrier)
and the in-kernel clock.  usage: duo_stamp.py dt:E:L:QH:KH:B[:causal] ..."""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
pkg._lib.debug_set("fwd_duo", 1)
for c in sys.argv[1:] or ["bf16:64:4096:4:4:4"]:
    f = c.split(":"); dt, (E, L, QH, KH, B) = f[0], map(int, f[1:6]); causal = len(f) > 6 and f[6] == "causal"
    q = torch.randn(B, QH, L, E, device=dev).to(DT[dt]); k = torch.randn(B, KH, L, E, device=dev).to(DT[dt]); v = torch.randn_like(k)
    o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
    n = max(200, int(2.0e6 / (4.0 * L * L * QH * B * E / 0.9e9 + 10)))    # ~2 s of back-to-back launches (in-kernel clock settles)
    for _ in range(n // 2): pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 2): pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal)
    e1.record(); torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / (n // 2) * 1e3
    rows = o.reshape(B * QH, L // 256, 256, E)[:, :, 0, :].contiguous().view(torch.int64)[..., :16].reshape(-1, 16).cpu().double()
    t = rows[:, 0:8:2]; r = rows[:, 1:8:2]; nt = rows[:, 8]
    clk = ((t[:, 3] - t[:, 0]) / (r[:, 3] - r[:, 0]) * 0.1).median().item()          # GHz
    pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    per_pair = (loop / (nt + 2) * 2).median().item()                                  # one M + one V = two half-steps = one tile of EACH group
    span = (r[:, 3].max() - r[:, 0].min()).item() * 10e-3
    acc = rows[:, 11:16]; it = ((nt + 1) / 2).clamp_min(1)                            # iterations of wave 0 (group 0: tiles 0, 2, ...)
    ph = [(acc[:, i] / it).median().item() for i in range(5)]
    print(f"{c}: {n} launches; clock {clk:.3f} GHz; per WG median cycles: prologue {pro.median().item():.0f} loop {loop.median().item():.0f} epilogue {epi.median().item():.0f};"
          f" tiles {nt.median().item():.0f}; cycles per (M + V) iteration {per_pair:.0f} = per 64x64 wave-tile per SIMD {per_pair / 2:.0f}; wall {wall:.1f} us / launch, first entry -> last exit {span:.1f} us", flush=True)
    print(f"{c}: wave 0 per iteration: M {ph[0]:.0f}  barrier {ph[1]:.0f}  V {ph[2]:.0f}  DMA wait {ph[3]:.0f}  barrier {ph[4]:.0f}   (sum {sum(ph):.0f})", flush=True)
    rows4 = o.reshape(B * QH, L // 256, 256, E)[:, :, 1, :].contiguous().view(torch.int64)[..., :5].reshape(-1, 5).cpu().double()
    it4 = (nt / 2).clamp_min(1)
    ph4 = [(rows4[:, i] / it4).median().item() for i in range(5)]
    print(f"{c}: wave 4 per iteration: M {ph4[0]:.0f}  barrier {ph4[1]:.0f}  V {ph4[2]:.0f}  DMA wait {ph4[3]:.0f}  barrier {ph4[4]:.0f}   (sum {sum(ph4):.0f})", flush=True)
