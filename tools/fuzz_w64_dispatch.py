"""dev: 60 random shapes (GQA, ragged lengths, causal, variable-length masks, bf16 / fp16, E 64 / 128) through whatever forward the
launcher picks against the 32-row forms (knob fwd_w64=0) on the same inputs: max difference and NaN pattern."""
import sys, torch, numpy as np
import os
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device('cuda:0')
rng = np.random.default_rng(123)
bad = 0; n_w64 = 0
for it in range(60):
    E = int(rng.choice([64, 128])); dt = [torch.bfloat16, torch.float16][int(rng.integers(2))]
    KH = int(rng.choice([1, 2, 4, 8])); rep = int(rng.choice([1, 2, 4])); QH = KH * rep
    B = int(rng.integers(1, 5)); QL = int(rng.integers(200, 2300)); KL = int(rng.integers(200, 2300))
    causal = bool(rng.integers(2)); pad = bool(rng.integers(2))
    if causal: KL = QL
    # enough workgroups for the 64-row form
    while ((QL + 255) // 256) * QH * B < 160: B += 1
    g = torch.Generator(device=dev).manual_seed(it)
    q = torch.randn(B, QH, QL, E, generator=g, device=dev).to(dt); k = torch.randn(B, KH, KL, E, generator=g, device=dev).to(dt); v = torch.randn(B, KH, KL, E, generator=g, device=dev).to(dt)
    mask = None
    if pad:
        lens = torch.tensor(rng.integers(1, KL + 1, size=B), device=dev)
        mask = (torch.arange(KL, device=dev)[None, :] < lens[:, None]).contiguous()
    form = pkg._lib.fwd_form(pkg.attention._desc(q, k, v, causal), False, mask is not None)
    outs = {}
    for w in (0, -1):
        pkg._lib.debug_set("fwd_w64", w)
        outs[w] = pkg._flash_attention(q, k, v, causal=causal, kpad_mask=mask)
    pkg._lib.debug_set("fwd_w64", -1)
    torch.cuda.synchronize()
    n_w64 += form == "fa_fwd_w64_kernel"
    o0, o1 = outs[0][0].float(), outs[-1][0].float()
    nan_ok = bool((torch.isnan(o0) == torch.isnan(o1)).all())
    err = (torch.nan_to_num(o0) - torch.nan_to_num(o1)).abs().max().item() / max(torch.nan_to_num(o0).abs().max().item(), 1e-9)
    tol = 2e-2 if dt == torch.bfloat16 else 4e-3
    ok = nan_ok and err < tol
    if not ok: bad += 1
    print(f"{it:2d} {str(dt)[6:]:9s} E{E} B{B} QH{QH} KH{KH} QL{QL} KL{KL} causal={int(causal)} pad={int(pad)} form={form[7:13]} err={err:.2e} nan_ok={nan_ok} {'OK' if ok else 'BAD'}", flush=True)
print("bad", bad, "w64 cases", n_w64)
