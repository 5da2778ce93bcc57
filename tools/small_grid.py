"""dev: forward and backward time of launches that do not fill the chip with 256-row blocks, with the small-grid forms of round 4 (32-row
waves: fa_fwd_duo NZ = 1 / E = 128, BwdW64Shape NARROW) at the launcher's own choice and switched off, alternated on ONE box.
usage: small_grid.py [out.jsonl] [dt:E:L:QH:KH:B ...]      every shape runs non-causal and causal"""
import json, os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
args = sys.argv[1:]
out = args.pop(0) if args and args[0].endswith(".jsonl") else None
shapes = args or ["bf16:64:2048:4:4:4", "f16:64:2048:4:4:4", "f32:64:2048:4:4:4", "bf16:128:2048:4:4:4", "f16:128:2048:4:4:4",
                  "bf16:64:4096:8:8:1", "bf16:128:4096:8:8:1", "bf16:64:1024:8:8:4", "bf16:128:1024:8:8:4", "bf16:128:8192:8:8:1"]


def timeit(f, n):
    for _ in range(n): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# "off": the forms of round 3 -- 64-row waves everywhere (fwd_duo = 2 keeps the two-wave forward of E = 64 at 64 rows; E = 128 has none)
ARMS = {"small-grid forms off": lambda E: dict(fwd_duo=2 if E == 64 else 0, bwd_narrow=0), "default": lambda E: dict(fwd_duo=-1, bwd_narrow=-1)}
rows = []
for c in shapes:
    dt, E, L, QH, KH, B = c.split(":"); E, L, QH, KH, B = int(E), int(L), int(QH), int(KH), int(B)
    g = torch.Generator(device=dev).manual_seed(1)
    q, do = (torch.randn(B, QH, L, E, generator=g, device=dev).to(DT[dt]) for _ in range(2))
    k, v = (torch.randn(B, KH, L, E, generator=g, device=dev).to(DT[dt]) for _ in range(2))
    for causal in (False, True):
        fl = pkg.workmodel.attention_flops(E, L, L, QH, B, causal=causal)
        rec = dict(dtype=dt, E=E, L=L, QH=QH, KH=KH, B=B, causal=causal)
        for name, knobs in ARMS.items():
            for kk, vv in knobs(E).items(): pkg._lib.debug_set(kk, vv)
            o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
            fwd = lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal)
            fwd(); torch.cuda.synchronize()
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal), dtype=torch.uint8, device=dev)
            bwd = lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal)
            n = max(20, int(0.2e6 / (fl / 0.3e9 + 10)))
            tf, tb = timeit(fwd, n), timeit(bwd, max(10, n // 3))
            tag = "off" if name != "default" else "on"
            rec.update({f"fwd_us_{tag}": round(tf, 1), f"bwd_us_{tag}": round(tb, 1), f"fwd_kernel_{tag}": pkg._lib.fwd_form(pkg._lib.FaDesc(
                dtype={"f32": 0, "f16": 1, "bf16": 2}[dt], emb=E, ql=L, kl=L, qh=QH, kh=KH, batch=B, causal=int(causal), emb_k=0, emb_v=0, kl_v=0, kh_v=0))})
        for kk in ("fwd_duo", "bwd_narrow"): pkg._lib.debug_set(kk, -1)
        rec["fwd_tflops_on"] = round(fl / rec["fwd_us_on"] / 1e6, 1); rec["bwd_tflops_on"] = round(2.5 * fl / rec["bwd_us_on"] / 1e6, 1)
        rows.append(rec)
        print(json.dumps(rec), flush=True)
if out:
    with open(out, "w") as f:
        for r in rows: f.write(json.dumps(r) + "\n")
