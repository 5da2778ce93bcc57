"""dev: in-kernel time stamps of the 64-row forward (library built with `make DEV=1 VAR=-DNNOP_W64_STAMP=1`, passed through
NNOP_LIB_PATH): prologue / loop / epilogue duration per workgroup in shader cycles, cycles per kv tile, cycles per MFMA and the
in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz).  usage: w64_stamp.py dt:E:L:QH:KH:B[:causal] ..."""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
pkg._lib.debug_set("fwd_w64", 1); pkg._lib.debug_set("fwd_duo", 0)      # (the two-waves-per-SIMD form has its own tool: duo_stamp.py)
pkg._lib.debug_set("fwd_persist", 0)          # the stamps describe ONE block per workgroup (the persistent form would overwrite them per block)
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
    rows = o.reshape(B * QH, L // 256, 256, E)[:, :, 0, :].contiguous().view(torch.int64)[..., :11].reshape(-1, 11).cpu().double()
    t = rows[:, 0:8:2]; r = rows[:, 1:8:2]; nt = rows[:, 8]
    clk = ((t[:, 3] - t[:, 0]) / (r[:, 3] - r[:, 0]) * 0.1).median().item()          # GHz
    pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    mf = (2 * 2 * (E // 16) + 2 * 4 * (E // 32) + (8 if E <= 64 else 0))
    per_tile = (loop / nt).median().item()
    p1, p2 = (rows[:, 9] - t[:, 0]).median().item(), (rows[:, 10] - rows[:, 9]).median().item()
    span = (r[:, 3].max() - r[:, 0].min()).item() * 10e-3                              # us, first entry -> last exit
    busy = (r[:, 3] - r[:, 0]).sum().item() * 10e-3 / 256                              # us of workgroup residency per CU (256 CUs, one workgroup each)
    by_phase = [(x * 1.0).sum().item() / (t[:, 3] - t[:, 0]).sum().item() for x in (pro, loop, epi)]
    print(f"{c}: residency per CU {busy:.1f} us of {span:.1f} us ({busy / span:.3f}); share of resident cycles: prologue {by_phase[0]:.3f} loop {by_phase[1]:.3f} epilogue {by_phase[2]:.3f}", flush=True)
    print(f"{c}: {n} launches; clock {clk:.3f} GHz; per WG median cycles: prologue {pro.median().item():.0f} (K0+Q landed {p1:.0f}, S0 +{p2:.0f})  loop {loop.median().item():.0f}"
          f"  epilogue {epi.median().item():.0f}; tiles {nt.median().item():.0f}; cycles / tile {per_tile:.0f} = {per_tile / mf:.1f} per MFMA ({mf} MFMA / tile)"
          f"; wall {wall:.1f} us / launch, first entry -> last exit {span:.1f} us; entry spread {(r[:, 0].max() - r[:, 0].min()).item() * 10e-3:.2f} us", flush=True)
