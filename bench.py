#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Flash-Attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): attention TFLOP/s (+ GB/s), forward and forward+backward, at
E=64, L=4096, H=4, B=4.  Workload at N=1 = BASELINE config C2: bf16, non-causal, forward -- a
"step" is ONE call of nnop_fa_fwd over one (E,L,H,B) batch of synthetic N(0,1) Q/K/V already
resident in HBM.  `value` = algorithmic forward FLOPs of all ranks / wall time of the K timed steps
(max over ranks).  N > 1: one process per GPU, no data-path collective ((batch, kv-head) slices are
independent); the optional all-gather of O is timed separately.
  --scaling weak   (default, what the driver runs): every rank runs the same per-GPU batch.
  --scaling strong : the configuration is ONE global problem (C2: 16 (batch, kv-head) units; C5: B = 64)
                     partitioned with nnop.jl_amd/shard.py `rectangles` -- contiguous unit ranges, <= 3 dense
                     rectangles per rank, pointer offsets only; a step = the rank's rectangles, one launch each.

Extra objects on the JSON line:
  roofline     -- dominant kernel (fa_fwd_duo_kernel at C2 since round 4; fa_fwd_w64_kernel at C3-C5): algorithmic FLOPs per launch / average launch
                  duration measured here with HIP events on the launch stream, against the dense
                  bf16 MFMA peak (2516.6 TFLOP/s = 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz).
  cpu_baseline -- the oracle's fp32 port of the reference's naive attention
                  (benchmarks/main.jl:26-43) timed on this host's cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2516.6, "f16": 2516.6, "f32": 157.3}
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (dtype, E, L, QH, KH, B, causal)
    "c2": ("bf16", 64, 4096, 4, 4, 4, False),           # headline (BASELINE.json configs[1])
    "c1gpu": ("f32", 64, 4096, 4, 4, 4, False),         # fp32 twin of configs[0] on the GPU
    "c3": ("bf16", 128, 8192, 32, 32, 8, True),          # configs[2]
    "c4": ("f16", 128, 4096, 32, 8, 16, False),          # configs[3]: GQA 32/8, variable sequence length
    "c5": ("bf16", 128, 16384, 32, 32, 8, True),         # configs[4]: ONE GPU's shard (B = 64 / 8) of the 8-GPU config
}
# --scaling strong: the GLOBAL problem that is partitioned over the ranks
GLOBAL_CONFIGS = {
    "c2": ("bf16", 64, 4096, 4, 4, 4, False),            # 16 units: 8 GPUs -> 2 units each (B < G: split over kv heads)
    "c5": ("bf16", 128, 16384, 32, 32, 64, True),        # configs[4] itself: B = 64 -> 8 batches per GPU on 8 GPUs
}
# C4 lengths: numpy.random.default_rng(0).integers(1024, 4097, size=16)
C4_LENS = [3637, 2981, 2594, 1853, 1969, 1149, 1255, 1074, 1562, 3523, 3019, 3828, 2571, 2888, 4007, 3265]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bwd", action="store_true", help="skip the (untimed-region) fwd+bwd leg")
    ap.add_argument("--gather", action="store_true", help="also time the optional all-gather of O (N>1)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed: run the step back to back for this long BEFORE the W warm-up steps, so that the timed K steps "
                         "see the sustained (power-managed) clock instead of the first milliseconds after idle; 0 = off")
    return ap.parse_args()


def cpu_baseline(E, L, H, B):
    """oracle 'port' of benchmarks/main.jl naive attention, fp32, C1 shape, all host cores."""
    import numpy as np
    from oracle.naive_attention import naive_attention_f32, naive_attention_f32_fwd_bwd, attention_flops
    # all host cores; BLAS builds cap their pool (OpenBLAS: 128), so report what was granted
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threadpool_limits(limits=os.cpu_count() or 1)
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:                                   # noqa: BLE001
        threads = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    q, k, v, do = (rng.standard_normal((B, H, L, E), dtype=np.float32) for _ in range(4))
    naive_attention_f32(q[:1, :1], k[:1, :1], v[:1, :1])          # warm BLAS threads
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        naive_attention_f32(q, k, v)
        ts.append(time.perf_counter() - t0)
    t_fwd = statistics.median(ts)
    tb = []
    for _ in range(3):
        t0 = time.perf_counter()
        naive_attention_f32_fwd_bwd(q, k, v, do)
        tb.append(time.perf_counter() - t0)
    t_fb = statistics.median(tb)
    f = attention_flops(E, L, L, H, B, causal=False)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(f / t_fwd / 1e12, 4), "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
        "sample": f"full C1 workload (fp32 E={E} L={L} H={H} B={B}, 1 GiB score tensor): forward median of 5 "
                  f"runs = {t_fwd:.2f} s; forward+backward median of 3 runs = {t_fb:.2f} s",
        "fwd_s": round(t_fwd, 3), "fwd_bwd_s": round(t_fb, 3),
        "fwd_bwd_tflops": round(f * 3.5 / t_fb / 1e12, 4),
        "host_cpus": os.cpu_count(), "cpu_model": model,
    }


def main():
    args = parse()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    attention_flops, attention_bytes = pkg.workmodel.attention_flops, pkg.workmodel.attention_bytes
    pkg._lib.load()                                     # fail loudly if the HIP library is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # NNOP_BENCH_REHEARSE=1: N ranks on ONE GPU with gloo, to rehearse the multi-rank control flow on a 1-GPU box
    # (numbers are meaningless then); the real run is one rank per GPU over RCCL.
    rehearse = os.environ.get("NNOP_BENCH_REHEARSE") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    red_dev = torch.device("cpu") if rehearse else dev        # where the timing all-reduce lives

    strong = args.scaling == "strong"
    if strong and args.config not in GLOBAL_CONFIGS:
        raise SystemExit("--scaling strong: --config c2 or c5")
    dtn, E, L, QH, KH, B, causal = (GLOBAL_CONFIGS if strong else CONFIGS)[args.config]
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtn]
    # this rank's share: weak = the whole per-GPU configuration; strong = its rectangles of (batch, kv-head) units
    if strong:
        rects = pkg.shard.rectangles(B, KH, world, rank)
        shapes = [(r.b1 - r.b0, (r.kh1 - r.kh0) * (QH // KH), r.kh1 - r.kh0) for r in rects]
    else:
        shapes = [(B, QH, KH)]
    g = torch.Generator(device=dev).manual_seed(1000 + rank)           # N(0,1), never zeros
    mk = lambda b, h: torch.randn(b, h, L, E, generator=g, device=dev, dtype=torch.float32).to(dt)
    parts = []
    for (b_, qh_, kh_) in shapes:
        q, k, v, do = mk(b_, qh_), mk(b_, kh_), mk(b_, kh_), mk(b_, qh_)
        parts.append(dict(q=q, k=k, v=v, do=do, o=torch.empty_like(q),
                          ms=torch.empty(b_, qh_, L, dtype=dt, device=dev), ls=torch.empty(b_, qh_, L, dtype=dt, device=dev)))
    q, k, v, do, o, ms, ls = (parts[0][n] for n in ("q", "k", "v", "do", "o", "ms", "ls")) if parts else (None,) * 7

    kpad, kv_lens = None, None
    if args.config == "c4":
        kv_lens = C4_LENS
        kpad = (torch.arange(L, device=dev)[None, :] < torch.tensor(kv_lens, device=dev)[:, None]).contiguous()

    def step():
        for t in parts:
            pkg.fa_fwd_into(t["o"], t["ms"], t["ls"], t["q"], t["k"], t["v"], causal=causal, kpad_mask=kpad)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def settle(fn):
        # A launch of the headline workload lasts ~70 us: K = 20 steps straight after idle measure the DVFS ramp, not the kernel.
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while args.settle_ms > 0 and time.perf_counter() < t_end:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()

    settle(step)
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    barrier()
    t_dev = ev0.elapsed_time(ev1) * 1e-3                 # HIP events on the launch stream
    t = torch.tensor([t_wall, t_dev], device=red_dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_wall, t_dev = float(t[0]), float(t[1])

    esz = 2 if dtn != "f32" else 4
    f_fwd = attention_flops(E, L, L, QH, B, causal=causal, kv_lens=kv_lens)      # weak: per GPU; strong: the global problem
    by_fwd = attention_bytes(E, L, L, QH, KH, B, esz)
    n_rep = 1 if strong else world                       # how many copies of (f_fwd, by_fwd) the job processed per step
    ms_per_step = t_wall / args.steps * 1e3
    value = n_rep * f_fwd / (t_wall / args.steps) / 1e12
    # roofline of the dominant kernel, measured on THIS rank (rank 0): its own FLOPs per step / its device time per launch
    my_b = sum(t["q"].shape[0] * t["q"].shape[1] for t in parts)                  # (batch x q-head) slices of this rank
    f_mine = f_fwd * my_b // (B * QH) if strong else f_fwd
    by_mine = by_fwd * my_b // (B * QH) if strong else by_fwd
    n_launch = max(len(parts), 1)
    kern_s = t_dev / args.steps / n_launch               # average launch duration
    achieved = (f_mine / n_launch) / kern_s / 1e12 if parts else 0.0

    extra = {}
    # ---- the opt-in forward variant (NNOP_FWD_EXACT_SCALE=0: scale * log2e folded into Q, rounded once -- outside the parity
    # tolerance on large logits), reported beside the default; `value` is the default's -------------------
    if not strong and parts and dtn != "f32":
        prev = pkg._lib.debug_set("fwd_exact_scale", 0)
        try:
            settle(step)
            barrier()
            ev0.record()
            for _ in range(max(args.steps, 20)):
                step()
            ev1.record()
            torch.cuda.synchronize()
            t_fold = ev0.elapsed_time(ev1) * 1e-3 / max(args.steps, 20)
        finally:
            pkg._lib.debug_set("fwd_exact_scale", prev)
        step()                                            # leave the default's outputs behind
        torch.cuda.synchronize()
        extra["fwd_folded_scale_tflops"] = round(f_mine / t_fold / 1e12, 2)
        extra["fwd_folded_scale_note"] = "opt-in variant (NNOP_FWD_EXACT_SCALE=0), this rank, by HIP events; not the default and not `value`"

    # ---- fwd+bwd leg (outside the timed region above; same inputs) -----------------------------
    if not args.no_bwd and not strong:
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal), dtype=torch.uint8, device=dev)
        fb = lambda: (pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal, kpad_mask=kpad),
                      pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal, kpad_mask=kpad))
        nb = max(args.steps // 4, 20)                    # >= 20 timed iterations whatever --steps says
        settle(fb)
        for _ in range(max(args.warmup // 4, 2)):
            fb()
        barrier()
        ev0.record()
        for _ in range(nb):
            fb()
        ev1.record()
        torch.cuda.synchronize()
        tb = torch.tensor([ev0.elapsed_time(ev1) * 1e-3 / nb], device=red_dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        t_fb = float(tb[0])
        by_fb = attention_bytes(E, L, L, QH, KH, B, q.element_size(), mode="fwd+bwd")
        extra.update({
            "fwd_bwd_tflops": round(world * f_fwd * 3.5 / t_fb / 1e12, 2),
            "fwd_bwd_ms": round(t_fb * 1e3, 4),
            "fwd_bwd_iters": nb,
            "fwd_bwd_gbps": round(world * by_fb / t_fb / 1e9, 1),
        })
        # ---- the backward alone, and its kernels one by one (rank 0's device; HIP events on the launch stream) -----------------
        # nnop_fa_bwd = preprocess + dK/dV + dQ on one stream.  The per-kernel durations come from running prefixes of the three
        # passes (test-only knob bwd_stages, csrc/tuning.hpp) and differencing: t(pre), t(pre + dK/dV) - t(pre), t(all) - t(pre + dK/dV).
        bw = lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal, kpad_mask=kpad)

        def timed(fn, n):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(n):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) * 1e-3 / n

        settle(bw)
        t_all = timed(bw, nb)
        stage_t = {}
        for mask in (1, 3):
            prev = pkg._lib.debug_set("bwd_stages", mask)
            try:
                stage_t[mask] = timed(bw, nb)
            finally:
                pkg._lib.debug_set("bwd_stages", prev)
        bw()                                              # leave complete gradients behind
        torch.cuda.synchronize()
        bdesc = pkg._lib.FaDesc(dtype={"f32": 0, "f16": 1, "bf16": 2}[dtn], emb=E, ql=L, kl=L, qh=q.shape[1], kh=k.shape[1],
                                batch=q.shape[0], causal=int(causal), emb_k=0, emb_v=0, kl_v=0, kh_v=0)
        kn_kv, kn_q = pkg._lib.bwd_kernels(bdesc, False, kpad is not None)
        fused_pre = kn_kv.startswith("fa_bwd_w64") and kn_q.startswith("fa_bwd_w64")      # csrc/fa_bwd_inst.hpp: p.fused
        f_bwd = 2.5 * f_fwd                               # algorithmic (S and dP counted once: SURVEY.md section 8(d))
        t_pre, t_kv, t_q = stage_t[1], max(stage_t[3] - stage_t[1], 1e-9), max(t_all - stage_t[3], 1e-9)
        extra["bwd_tflops"] = round(world * f_bwd / t_all / 1e12, 2)
        extra["roofline_bwd"] = {
            "bound": "mfma", "achieved": round(f_bwd / t_all / 1e12, 2), "peak": PEAK_TFLOPS[dtn], "unit": "TFLOP/s",
            "frac": round(f_bwd / t_all / 1e12 / PEAK_TFLOPS[dtn], 4), "avg_call_us": round(t_all * 1e6, 2), "iters": nb,
            "flops_per_call": int(f_bwd),
            "kernels": [
                ({"name": "fa_bwd_pre_kernel", "avg_us": round(t_pre * 1e6, 2), "bound": "hbm"} if not fused_pre else
                 {"name": "(no preprocess launch: fused into the dQ kernel; this is the time of the call with no kernel in it)",
                  "avg_us": round(t_pre * 1e6, 2)}),
                # products per pass: dK/dV = S, dP, dV, dK (2 of the 2.5 fwd-units); dQ = S, dP recomputed + dQ (0.5 algorithmic, 1.5 executed)
                {"name": kn_kv, "avg_us": round(t_kv * 1e6, 2), "executed_tflops": round(2.0 * f_fwd / t_kv / 1e12, 1),
                 "executed_frac": round(2.0 * f_fwd / t_kv / 1e12 / PEAK_TFLOPS[dtn], 4)},
                {"name": kn_q, "avg_us": round(t_q * 1e6, 2), "executed_tflops": round(1.5 * f_fwd / t_q / 1e12, 1),
                 "executed_frac": round(1.5 * f_fwd / t_q / 1e12 / PEAK_TFLOPS[dtn], 4)},
            ],
            "how": "HIP events around prefixes of the three passes (bwd_stages knob), differenced",
        }
    if args.gather and dist is not None and (not rehearse or strong):
        # the one optional collective: replicate the unit-sharded O on every rank (RCCL all-gather over xGMI; gloo in rehearsal)
        if strong:
            rep = QH // KH
            loc = [t["o"].reshape(-1, rep, L, E) for t in parts]
            local = (torch.cat(loc, dim=0) if len(loc) > 1 else loc[0]) if loc else torch.empty((0, rep, L, E), dtype=dt, device=dev)
            local = local.cpu() if rehearse else local
            gat = lambda: pkg.shard.all_gather_units(local, B * KH)
        else:
            full = torch.empty((world,) + tuple(o.shape), dtype=o.dtype, device=dev)
            gat = lambda: dist.all_gather_into_tensor(full, o)
        for _ in range(2):
            gat()
        barrier()
        tg0 = time.perf_counter()
        for _ in range(5):
            gat()
        torch.cuda.synchronize()
        tg = torch.tensor([(time.perf_counter() - tg0) / 5], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        extra["allgather_o_ms"] = round(float(tg[0]) * 1e3, 4)
        extra["compute_plus_gather_tflops"] = round(n_rep * f_fwd / (t_wall / args.steps + float(tg[0])) / 1e12, 2)
        if not strong and not rehearse and B >= 2:
            # ... and overlapped: the batch in chunks, chunk i's gather (RCCL's stream) beside chunk i+1's forward (shard.py)
            nch = 2 if B < 8 else 4
            cuts = [B * i // nch for i in range(nch + 1)]
            sl = [slice(cuts[i], cuts[i + 1]) for i in range(nch)]
            o_ch = [o[c] for c in sl]
            full_ch = [torch.empty((world * oc.shape[0],) + tuple(oc.shape[1:]), dtype=o.dtype, device=dev) for oc in o_ch]
            kp_ch = [None if kpad is None else kpad[c] for c in sl]
            step_chunk = lambda i: pkg.fa_fwd_into(o_ch[i], ms[sl[i]], ls[sl[i]], q[sl[i]], k[sl[i]], v[sl[i]], causal=causal, kpad_mask=kp_ch[i])
            ov = lambda: pkg.shard.forward_with_overlapped_gather(step_chunk, o_ch, full_ch)
            for _ in range(2):
                ov()
            barrier()
            to0 = time.perf_counter()
            for _ in range(5):
                ov()
            torch.cuda.synchronize()
            to = torch.tensor([(time.perf_counter() - to0) / 5], device=red_dev, dtype=torch.float64)
            dist.all_reduce(to, op=dist.ReduceOp.MAX)
            extra["compute_overlapped_with_gather_ms"] = round(float(to[0]) * 1e3, 4)
            extra["compute_overlapped_with_gather_tflops"] = round(n_rep * f_fwd / float(to[0]) / 1e12, 2)
            extra["gather_chunks"] = nch

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # HBM bytes per launch from the PMC counters: NOT measured in this run -- the value of the last committed rocprofv3
    # --pmc collection of this workload (profiles/traffic.json names the run it came from), or null
    traffic, traffic_src = None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj) and not strong:
        try:
            ent = json.load(open(tj)).get(args.config, {})
            traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source")
        except Exception:                               # noqa: BLE001
            traffic = None
    fdesc = pkg._lib.FaDesc(dtype={"f32": 0, "f16": 1, "bf16": 2}[dtn], emb=E, ql=L, kl=L, qh=parts[0]["q"].shape[1] if parts else QH,
                            kh=parts[0]["k"].shape[1] if parts else KH, batch=parts[0]["q"].shape[0] if parts else B,
                            causal=int(causal), emb_k=0, emb_v=0, kl_v=0, kh_v=0)
    kernel_name = pkg._lib.fwd_form(fdesc, False, kpad is not None)
    out = {
        "metric": "attention TFLOPs/s + GB/s (fwd, fwd+bwd) at E=64,L=4096,H=4,B=4" if args.config == "c2" else
                  f"attention TFLOPs/s + GB/s (fwd, fwd+bwd), workload {args.config}",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle_ms": args.settle_ms, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": dtn, "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: {dtn} {'causal' if causal else 'non-causal'} "
                               f"flash_attention forward, E={E} L={L} QH={QH} KH={KH} B={B} " + ("global" if strong else "per GPU"),
                   "per_gpu_batch": (B if not strong else None), "global_batch": (B if strong else B * world),
                   "parallelism": f"dp{world} over (batch, kv-head) slices" + (f", {B * KH} units in contiguous ranges" if strong else ""),
                   "step": "one nnop_fa_fwd call" + (" per rectangle of the rank's unit range" if strong else "")},
        "fwd_gbps": round(n_rep * by_fwd / (t_wall / args.steps) / 1e9, 1),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[dtn], "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_TFLOPS[dtn], 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": kernel_name, "avg_launch_us": round(kern_s * 1e6, 2),
                     "flops_per_launch": f_mine // n_launch, "algorithmic_bytes_per_launch": by_mine // n_launch,
                     "hbm_frac_at_this_rate": round(by_mine / n_launch / kern_s / 1e9 / HBM_PEAK_GBS, 4)},
    }
    out.update(extra)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(64, 4096, 4, 4)
        out["cpu_baseline"]["gpu_speedup_fwd"] = round(value / max(out["cpu_baseline"]["value"], 1e-12), 1)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
