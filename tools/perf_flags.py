"""The 8 flag combinations of flash_attention (pair bias x causal x key-padding mask) on the reference's pair-bias benchmark shape
(benchmarks/main.jl: E=64 L=2048 H=4 B=4) and one larger shape: forward / backward time, TFLOP/s, and for the pair-bias runs the
bytes of the pair (forward) / pair + dpair (backward) streams against the time they would take at the HBM peak (8 TB/s) -- the
floor of a kernel that has to read (and write) that tensor once.  One JSON line per run.  usage: perf_flags.py [out.jsonl]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
HBM = 8.0e12
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize()
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    n = max(5, min(200, int(0.3e3 / max(e0.elapsed_time(e1), 1e-3))))       # ~0.3 s per measurement
    for _ in range(n): f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for (E, L, H, B) in ((64, 2048, 4, 4), (128, 2048, 8, 2)):
    for dtn, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
        g = torch.Generator(device=dev).manual_seed(0)
        mk = lambda *s: torch.randn(*s, generator=g, device=dev).to(dt)
        q, k, v, do = mk(B, H, L, E), mk(B, H, L, E), mk(B, H, L, E), mk(B, H, L, E)
        pair = mk(B, L, L, H)
        lens = torch.tensor([L - 11 * (i + 1) for i in range(B)], device=dev)
        mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).contiguous()
        for use_pair in (False, True):
            for causal in (False, True):
                for use_mask in (False, True):
                    pr, mk_ = (pair if use_pair else None), (mask if use_mask else None)
                    o, ms, ls = pkg._flash_attention(q, k, v, pr, causal=causal, kpad_mask=mk_)
                    tf = timeit(lambda: pkg._flash_attention(q, k, v, pr, causal=causal, kpad_mask=mk_))
                    tb = timeit(lambda: pkg.grad_flash_attention(do, o, ms, ls, q, k, v, pr, causal=causal, kpad_mask=mk_))
                    fl = pkg.workmodel.attention_flops(E, L, L, H, B, causal=causal, kv_lens=lens.tolist() if (use_mask and not causal) else None)
                    rec = dict(dtype=dtn, E=E, L=L, H=H, B=B, pair=use_pair, causal=causal, kpad=use_mask,
                               fwd_us=round(tf * 1e6, 1), bwd_us=round(tb * 1e6, 1), fwd_tflops=round(fl / tf / 1e12, 1),
                               bwd_tflops=round(2.5 * fl / tb / 1e12, 1))
                    if use_pair:
                        pb = pair.numel() * pair.element_size()
                        rec.update(pair_bytes=pb, fwd_floor_us=round(pb / HBM * 1e6, 1), bwd_floor_us=round(2 * pb / HBM * 1e6, 1),
                                   fwd_over_floor=round(tf / (pb / HBM), 2), bwd_over_floor=round(tb / (2 * pb / HBM), 2))
                    line = json.dumps(rec)
                    print(line, flush=True)
                    if out: out.write(line + "\n")
