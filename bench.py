#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Flash-Attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): attention TFLOP/s (+ GB/s), forward and forward+backward, at
E=64, L=4096, H=4, B=4.  Workload at N=1 = BASELINE config C2: bf16, non-causal, forward -- a
"step" is ONE call of nnop_fa_fwd over one (E,L,H,B) batch of synthetic N(0,1) Q/K/V already
resident in HBM.  `value` = algorithmic forward FLOPs of all ranks / wall time of the K timed steps
(max over ranks).  N > 1: one process per GPU, each rank runs the same per-GPU batch (weak scaling
on the H x B axis -- (batch, head) slices are independent), no data-path
collective; the optional all-gather of O is timed separately.

Extra objects on the JSON line:
  roofline     -- dominant kernel (fa_fwd_split_kernel at C2): algorithmic FLOPs per launch / average launch
                  duration measured here with HIP events on the launch stream, against the dense
                  bf16 MFMA peak (2516.6 TFLOP/s = 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz).
  cpu_baseline -- the oracle's fp32 port of the reference's naive attention
                  (benchmarks/main.jl:26-43) timed on this host's cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
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
    return ap.parse_args()


def cpu_baseline(E, L, H, B):
    """oracle 'port' of benchmarks/main.jl naive attention, fp32, C1 shape, all host cores."""
    import numpy as np
    from oracle.naive_attention import naive_attention_f32, naive_attention_f32_fwd_bwd, attention_flops
    try:
        from threadpoolctl import threadpool_info
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

    dtn, E, L, QH, KH, B, causal = CONFIGS[args.config]
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtn]
    g = torch.Generator(device=dev).manual_seed(1000 + rank)           # N(0,1), never zeros
    mk = lambda h: torch.randn(B, h, L, E, generator=g, device=dev, dtype=torch.float32).to(dt)
    q, k, v, do = mk(QH), mk(KH), mk(KH), mk(QH)
    o = torch.empty_like(q)
    ms = torch.empty(B, QH, L, dtype=dt, device=dev)
    ls = torch.empty_like(ms)

    kpad, kv_lens = None, None
    if args.config == "c4":
        kv_lens = C4_LENS
        kpad = (torch.arange(L, device=dev)[None, :] < torch.tensor(kv_lens, device=dev)[:, None]).contiguous()

    def step():
        pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal, kpad_mask=kpad)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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

    f_fwd = attention_flops(E, L, L, QH, B, causal=causal, kv_lens=kv_lens)
    by_fwd = attention_bytes(E, L, L, QH, KH, B, q.element_size())
    ms_per_step = t_wall / args.steps * 1e3
    value = world * f_fwd / (t_wall / args.steps) / 1e12
    kern_s = t_dev / args.steps                          # average launch duration of fa_fwd_kernel
    achieved = f_fwd / kern_s / 1e12

    # ---- fwd+bwd leg (outside the timed region above; same inputs) -----------------------------
    extra = {}
    if not args.no_bwd:
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal), dtype=torch.uint8, device=dev)
        fb = lambda: (pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal, kpad_mask=kpad),
                      pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal, kpad_mask=kpad))
        nb = max(args.steps // 4, 5)
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
            "bwd_tflops": round(world * f_fwd * 2.5 / max(t_fb - kern_s, 1e-9) / 1e12, 2),
            "fwd_bwd_gbps": round(world * by_fb / t_fb / 1e9, 1),
        })
    if args.gather and dist is not None and not rehearse:
        full = torch.empty((world,) + tuple(o.shape), dtype=o.dtype, device=dev)
        for _ in range(2):
            dist.all_gather_into_tensor(full, o)
        barrier()
        ev0.record()
        for _ in range(5):
            dist.all_gather_into_tensor(full, o)
        ev1.record()
        torch.cuda.synchronize()
        extra["allgather_o_ms"] = round(ev0.elapsed_time(ev1) / 5, 4)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    traffic = None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        try:
            traffic = json.load(open(tj)).get(args.config, {}).get("hbm_bytes_per_launch")
        except Exception:                               # noqa: BLE001
            traffic = None
    out = {
        "metric": "attention TFLOPs/s + GB/s (fwd, fwd+bwd) at E=64,L=4096,H=4,B=4" if args.config == "c2" else
                  f"attention TFLOPs/s + GB/s (fwd, fwd+bwd), workload {args.config}",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtn, "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: {dtn} {'causal' if causal else 'non-causal'} "
                               f"flash_attention forward, E={E} L={L} QH={QH} KH={KH} B={B} per GPU",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world} over (batch, head) slices",
                   "step": "one nnop_fa_fwd call"},
        "fwd_gbps": round(world * by_fwd / (t_wall / args.steps) / 1e9, 1),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[dtn], "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_TFLOPS[dtn], 4), "traffic": traffic,
                     "kernel": ("fa_fwd_split_kernel" if (dtn != "f32" and E <= 64 and not causal and kpad is None)
                                else "fa_fwd_kernel"), "avg_launch_us": round(kern_s * 1e6, 2),
                     "flops_per_launch": f_fwd, "algorithmic_bytes_per_launch": by_fwd,
                     "hbm_frac_at_this_rate": round(by_fwd / kern_s / 1e9 / HBM_PEAK_GBS, 4)},
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
