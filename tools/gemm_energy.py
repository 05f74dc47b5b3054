"""GPU: energy per FLOP of the encoder's GEMMs -- ours (default tile choice, and the opt-in 32x32x16 / four-wave kernels) against the
kernel hipBLASLt picks for torch.matmul -- on the same operands, each run as a ~2.5 s continuous loop while a sampler thread reads the
socket power (hwmon power1_average / rocm-smi).  VERDICT r03 item 3b: under the socket power cap a kernel is "faster" exactly when it
spends less energy per FLOP; the wall-time columns of tools/gemm_bench.py cannot tell a better schedule from a lucky clock.
    python tools/gemm_energy.py            (prints one line per shape and kernel: TF, W, pJ / FLOP over idle, sclk)"""
import glob, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine

lib = engine.load_library()
import ctypes
lib.samrs_debug_set_gemm_variant.argtypes = [ctypes.c_int]
lib.samrs_debug_set_gemm_variant.restype = None
s = torch.cuda.current_stream().cuda_stream
SECONDS = float(os.environ.get("SECONDS_PER_ARM", "2.5"))


def _hwmon():
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/hwmon/hwmon*/power1_average"):
        try:
            int(open(p).read())
            return p
        except Exception:
            pass
    return None


HW = _hwmon()


def read_power():
    if HW:
        return int(open(HW).read()) / 1e6              # microwatts
    out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True).stdout
    import json
    d = json.loads(out)["card0"]
    return float(next(v for k, v in d.items() if "power" in k.lower()))


def read_sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        import json
        d = json.loads(out)["card0"]
        v = next(v for k, v in d.items() if "sclk" in k.lower())
        return v
    except Exception:
        return "?"


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                self.samples.append(read_power())
            except Exception:
                pass
            time.sleep(0.02 if HW else 0.0)


def arm(fn, flop):
    fn(); torch.cuda.synchronize()
    sm = Sampler(); sm.start()
    time.sleep(0.3)                                          # let the running average settle into the loop
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    clk = read_sclk()
    sm.stop = True; sm.join()
    ms = e0.elapsed_time(e1) / n
    smp = sm.samples[len(sm.samples) // 3:]                  # drop the ramp
    watts = sum(smp) / max(1, len(smp))
    return ms, watts, clk


idle = []
for _ in range(20):
    idle.append(read_power()); time.sleep(0.05)
idle_w = sum(idle) / len(idle)
print(f"power source: {HW or 'rocm-smi'}; idle {idle_w:.0f} W", flush=True)
g = torch.Generator().manual_seed(0)
M = 32768
for name, N, K, of32, gelu, acc in (("qkv", 3840, 1280, 0, 0, 0), ("proj+res", 1280, 1280, 1, 0, 1), ("lin1+gelu", 5120, 1280, 0, 1, 0), ("lin2+res", 1280, 5120, 1, 0, 1)):
    A = torch.randn(M, K, generator=g).cuda().half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().half()
    bias = torch.randn(N, generator=g).cuda()
    C = torch.zeros(M, N, dtype=torch.float32 if of32 else torch.int16, device="cuda")
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    flop = 2.0 * M * N * K
    # round 5: the four-wave 16x16x32 kernel (v38) next to the default; the 32x32x16 kernels of round 2 (v30 / v34) with ARMS_FULL=1
    arms = [("ours (default)", 8), ("ours w4x (v38)", 38)] + ([("ours m32 (v30)", 30), ("ours w4 (v34)", 34)] if os.environ.get("ARMS_FULL") else [])
    for label, var in arms:
        lib.samrs_debug_set_gemm_variant(var)
        ms, w, clk = arm(lambda: lib.samrs_k_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), None, 0, M, N, K, of32, gelu, acc, s), flop)
        print(f"{name:10s} {label:16s} {ms * 1e3:7.1f} us {flop / ms / 1e9:7.0f} TF  {w:6.0f} W  {(w - idle_w) * ms * 1e-3 / flop * 1e12:5.2f} pJ/FLOP over idle  sclk {clk}", flush=True)
    lib.samrs_debug_set_gemm_variant(8)
    ms, w, clk = arm(lambda: torch.matmul(A, W.t(), out=out), flop)
    print(f"{name:10s} {'hipBLASLt':16s} {ms * 1e3:7.1f} us {flop / ms / 1e9:7.0f} TF  {w:6.0f} W  {(w - idle_w) * ms * 1e-3 / flop * 1e12:5.2f} pJ/FLOP over idle  sclk {clk}", flush=True)
