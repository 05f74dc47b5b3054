"""GPU micro-benchmark of the LayerNorm kernel on the encoder's shape (8 ViT-H tiles: 32768 rows x 1280, fp32 -> ET)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
rows, D = int(os.environ.get("ROWS", "32768")), int(os.environ.get("D", "1280"))
reps = int(os.environ.get("REPS", "20"))
g = torch.Generator().manual_seed(0)
# two input buffers, alternated, so that a launch never finds its input in L2 / MALL from the previous launch
xs = [torch.randn(rows, D, generator=g).to(dev) * 3 + 0.5 for _ in range(2)]
gam = torch.randn(D, generator=g).to(dev); bet = torch.randn(D, generator=g).to(dev)
out = torch.empty(rows, D, dtype=torch.float16, device=dev)
def run(i):
    return lib.samrs_k_layernorm(1, xs[i & 1].data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, out.data_ptr(), None, rows, D, 0, 1, 64, 0, s)
assert run(0) == 0
torch.cuda.synchronize()
ref = torch.nn.functional.layer_norm(xs[0], (D,), gam, bet, 1e-6)
err = (out.float() - ref).abs().max().item()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps): run(i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gb = rows * D * (4 + 2) / 1e9
print(f"layernorm {rows}x{D}: {ms*1e3:.1f} us  ({gb/ms:.2f} TB/s algorithmic), max err vs torch {err:.2e}", flush=True)
