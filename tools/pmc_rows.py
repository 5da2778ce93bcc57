"""Workload for the HBM-traffic check of the row-wise operators: one shape per operator, 5 calls each, so that every
kernel name in the rocprofv3 counter output belongs to exactly one shape.  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_rows.py
and again with WRITE_SIZE (tools/pmc_rows.sh does both and summarises)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
dev = "cuda:0"
N, EMB = 16384, 4096
x = torch.randn(N, EMB, device=dev).to(torch.bfloat16)
dy = torch.randn(N, EMB, device=dev).to(torch.bfloat16)
w = torch.randn(EMB, device=dev); b = torch.randn(EMB, device=dev)
q = torch.randn(4, 32, 4096, 128, device=dev).to(torch.bfloat16)
k = torch.randn(4, 8, 4096, 128, device=dev).to(torch.bfloat16)
cos, sin = pkg.LlamaRotaryEmbedding(128)(torch.arange(4096, device=dev, dtype=torch.float32).expand(4, 4096).contiguous())
for _ in range(5):
    y = pkg.online_softmax(x)
    pkg.grad_online_softmax(dy, y)
    yr, rms = pkg._rms_norm(x, w)
    pkg.grad_rms_norm(dy, rms, x, w)
    yl, mu, sg = pkg._layer_norm(x, w, b)
    pkg.grad_layer_norm(dy, mu, sg, x, w, b)
    pkg.llama_rope(q, k, cos=cos, sin=sin)
torch.cuda.synchronize()
wm = pkg.workmodel
print("algorithmic bytes: softmax fwd/bwd", wm.softmax_bytes(EMB, N, 2), wm.softmax_bytes(EMB, N, 2, bwd=True),
      "norm fwd/bwd", wm.norm_bytes(EMB, N, 2), wm.norm_bytes(EMB, N, 2, bwd=True), "rope", wm.rope_bytes(128, 4096, 32, 8, 4, 2))
