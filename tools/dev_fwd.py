import sys, time, numpy as np, torch
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from util import make_inputs, oracle_fwd, to64
pkg = ge.load_package()
dev = torch.device("cuda:0")
for dt in ["bf16", "f16", "f32"]:
    for (B, QH, KH, QL, KL, E, causal, pad, pair) in [
        (1, 1, 1, 256, 256, 64, False, None, False),
        (2, 2, 2, 512, 512, 64, False, None, False),
        (2, 2, 2, 512, 512, 128, False, None, False),
        (2, 2, 2, 256, 256, 16, False, None, False),
        (2, 2, 2, 256, 256, 32, False, None, False),
        (2, 2, 2, 255, 300, 64, False, None, False),
        (2, 2, 2, 512, 512, 64, True, None, False),
        (2, 4, 2, 511, 511, 64, True, "ref", False),
        (2, 2, 2, 300, 300, 32, False, "lens", True),
    ]:
        d = make_inputs(0, B, QH, KH, QL, KL, E, dt, dev, pair=pair, pad=pad, need_do=False)
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
        err = np.nanmax(np.abs(to64(o) - o_ref)) / np.nanmax(np.abs(o_ref))
        merr = np.nanmax(np.abs(to64(ms) - ms_ref))
        lse = to64(ms) + np.log(to64(ls)); lse_ref = ms_ref + np.log(ls_ref)
        print(f"{dt} B{B} QH{QH} KH{KH} QL{QL} KL{KL} E{E} c{int(causal)} pad={pad} pair={pair}: o relerr {err:.2e} ms err {merr:.2e} lse err {np.nanmax(np.abs(lse-lse_ref)):.2e} nan={np.isnan(to64(o)).sum()}", flush=True)
# quick perf at C2
for dt in ["bf16", "f16", "f32"]:
    d = make_inputs(0, 4, 4, 4, 4096, 4096, 64, dt, dev, need_do=False)
    for nw in ["8", "4"]:
        import os; os.environ["NNOP_FWD_NW"] = nw
        for _ in range(3): pkg._flash_attention(d["q"], d["k"], d["v"], causal=False)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): pkg._flash_attention(d["q"], d["k"], d["v"], causal=False)
        e1.record(); torch.cuda.synchronize()
        ms_ = e0.elapsed_time(e1) / 20
        print(f"C2 {dt} NW={nw}: {ms_*1e3:.1f} us  {68.719476736/ms_:.1f} TFLOP/s", flush=True)
