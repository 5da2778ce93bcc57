"""dev: in-kernel time stamps of the one-wave-per-SIMD backward kernels (library built with `make DEV=1 VAR=-DNNOP_BW64_STAMP=1
OUTDIR=../lib_stamp BUILD=../build_stamp`, passed through NNOP_LIB_PATH): prologue / loop / epilogue per workgroup in shader cycles,
cycles per iteration and per MFMA, in-kernel clock.  The stamps overwrite the first gradient row of every workgroup (results WRONG).
usage: bw64_stamp.py dt:E:L:QH:KH:B[:causal] ..."""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
pkg._lib.debug_set("bwd_persist", 0)          # the stamps describe ONE block per workgroup
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
for c in sys.argv[1:] or ["bf16:64:4096:4:4:4"]:
    f = c.split(":"); dt, (E, L, QH, KH, B) = f[0], map(int, f[1:6]); causal = len(f) > 6 and f[6] == "causal"
    g = torch.Generator(device=dev).manual_seed(1)
    q, do = (torch.randn(B, QH, L, E, generator=g, device=dev).to(DT[dt]) for _ in range(2))
    k, v = (torch.randn(B, KH, L, E, generator=g, device=dev).to(DT[dt]) for _ in range(2))
    o, ms, ls = pkg._flash_attention(q, k, v, causal=causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal), dtype=torch.uint8, device=dev)
    run = lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal)
    fl = 2.5 * pkg.workmodel.attention_flops(E, L, L, QH, B, causal=causal)
    n = max(40, int(1.5e6 / (fl / 0.7e9 + 20)))
    for _ in range(n): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / n * 1e3
    for kind, t, H, rows_wg, mf in (("dK/dV", dk, KH, 256 if E <= 128 else 128, 8 * (E // 16)), ("dQ", dq, QH, 256 if E <= 128 else 128, 6 * (E // 16))):
        nb = L // rows_wg
        rows = t.reshape(B * H, nb, rows_wg, E)[:, :, 0, :].contiguous().view(torch.int64)[..., :7].reshape(-1, 7).cpu().double()
        tt, rr, ns = rows[:, 0:6:2], rows[:, 1:6:2], rows[:, 6]
        clk = ((tt[:, 2] - tt[:, 0]) / (rr[:, 2] - rr[:, 0]) * 0.1).median().item()
        pro, loop = tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1]
        live = ns > 0
        per_it = (loop[live] / (ns[live] + 1)).median().item()
        span = (rr[:, 2].max() - rr[:, 0].min()).item() * 10e-3
        busy = (rr[:, 2] - rr[:, 0]).sum().item() * 10e-3 / 256
        span_all = (rr[:, 2].max() - rr[:, 0].min()).item() * 10e-3
        print(f"{c} {kind}: residency per CU (entry -> loop exit, epilogue not included) {busy:.1f} us of {span_all:.1f} us ({busy / span_all:.3f}); {rows.shape[0]} workgroups", flush=True)
        print(f"{c} {kind}: clock {clk:.3f} GHz; per WG median cycles: prologue {pro.median().item():.0f}  loop {loop.median().item():.0f} (max {loop.max().item():.0f}); steps {ns.median().item():.0f};"
              f" cycles / iteration {per_it:.0f} = {per_it / mf:.1f} per MFMA ({mf} MFMA / iteration); first entry -> last loop exit {span:.1f} us; "
              f"entry spread {(rr[:, 0].max() - rr[:, 0].min()).item() * 10e-3:.2f} us", flush=True)
    print(f"{c}: wall {wall:.1f} us / backward", flush=True)
