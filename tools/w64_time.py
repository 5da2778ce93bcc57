"""dev: steady-state time of the forward the launcher picks (or NNOP knobs given via --w64 0|1) per shape: ~1 s of back-to-back
launches as warm-up (clock settles), then 3 timed batches.  usage: w64_time.py [--w64 N] dt:E:L:QH:KH:B:mode ...   (mode as in w64_check.py)
A/B between libraries: run it once per NNOP_LIB_PATH (tools/w64_ab.sh alternates them on one box)."""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
args = sys.argv[1:]
if args and args[0] == "--w64":
    pkg._lib.debug_set("fwd_w64", int(args[1])); args = args[2:]
out = []
for c in args:
    dt, E, L, QH, KH, B, mode = c.split(":"); E, L, QH, KH, B = int(E), int(L), int(QH), int(KH), int(B)
    KL = L - 37 if mode == "ragged" else L
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(B, QH, L, E, generator=g, device=dev).to(DT[dt])
    k = torch.randn(B, KH, KL, E, generator=g, device=dev).to(DT[dt])
    v = torch.randn(B, KH, KL, E, generator=g, device=dev).to(DT[dt])
    mask = None; lens = None
    if mode == "lens":
        lens = torch.randint(KL // 4, KL + 1, (B,), generator=torch.Generator().manual_seed(2))
        mask = (torch.arange(KL)[None, :] < lens[:, None]).to(dev).contiguous()
    if mode == "allvalid":                                  # masked-mode kernel on a problem without any masked element
        mask = torch.ones(B, KL, dtype=torch.bool, device=dev)
    causal = mode == "causal"
    o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
    f = lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal, kpad_mask=mask)
    fl = pkg.workmodel.attention_flops(E, L, KL, QH, B, causal=causal, kv_lens=None if lens is None else lens.tolist())
    est = fl / 0.8e9 + 5                                   # us per launch at ~800 TF
    nwarm, n = max(20, int(1.0e6 / est)), max(20, int(0.4e6 / est))
    for _ in range(nwarm): f()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    t = sorted(ts)[1]
    out.append(f"{c} {t:.1f}us {fl / t / 1e6:.0f}TF")
print(" | ".join(out), flush=True)
