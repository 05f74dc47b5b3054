"""GPU micro-benchmark of the two attention kernels on ViT-H shapes (8 images)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
n_img, heads, hd, grid = int(os.environ.get("NIMG", "8")), 16, 80, 64
D = heads * hd
reps = int(os.environ.get("REPS", "20"))
g = torch.Generator().manual_seed(0)
bias = torch.randn(3 * D, generator=g).to(dev)
for name, rows, tab in (("window", n_img * 4096, 27), ("global", n_img * 4096, 127)):
    qkv = torch.randn(rows, 3 * D, generator=g).to(dev).to(torch.float16)
    rh = (0.02 * torch.randn(tab, hd, generator=g)).to(dev); rw = (0.02 * torch.randn(tab, hd, generator=g)).to(dev)
    out = torch.empty(n_img * 4096, D, dtype=torch.float16, device=dev)
    def run():
        if name == "window":
            return lib.samrs_k_window_attention(1, qkv.data_ptr(), bias.data_ptr(), rh.data_ptr(), rw.data_ptr(), out.data_ptr(), n_img, grid, 14, heads, hd, s)
        return lib.samrs_k_global_attention(1, qkv.data_ptr(), rh.data_ptr(), rw.data_ptr(), out.data_ptr(), n_img, grid, heads, hd, s)
    assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    keys = 196 if name == "window" else 4096
    fl = 2.0 * 2.0 * n_img * 4096 * keys * hd * heads
    import hashlib
    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]      # variants of a kernel must agree bit for bit
    print(f"{name} attention n_img={n_img}: {ms*1e3:.1f} us  ({fl/ms/1e9:.1f} TF algorithmic)  out sha {digest}  "
          f"[WIN_PIPE={os.environ.get('SAMRS_WIN_PIPE', '-')} GLB_PIPE={os.environ.get('SAMRS_GLB_PIPE', '-')}]", flush=True)
