import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device('cuda:0')
x = torch.randn(1024, 1024, device=dev); w = torch.randn(1024, device=dev); b = torch.randn(1024, device=dev)
def t(f, n=2000):
    for _ in range(200): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("layer_norm 1024x1024 fp32 call-to-call: nnop %.1f us, torch %.1f us" % (t(lambda: pkg._layer_norm(x, w, b)), t(lambda: torch.nn.functional.layer_norm(x, (1024,), w, b))))
print("rms_norm   1024x1024 fp32 call-to-call: nnop %.1f us" % t(lambda: pkg._rms_norm(x, w, offset=0.0)))
q = torch.randn(1, 4, 256, 64, device=dev).to(torch.bfloat16)
print("flash_attention tiny (1x4x256x64 bf16) call-to-call: %.1f us" % t(lambda: pkg._flash_attention(q, q, q, causal=False)))
