import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import samrs_amd
from samrs_amd import synth
sam = samrs_amd.sam_model_registry["vit_tiny"](precision="f16", max_prompts=8, max_images=1).to("cuda")
eng = sam.engine; lib = eng.lib
lib.samrs_debug_copy_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]
lib.samrs_debug_copy_buffer.restype = C.c_int
img = torch.as_tensor(synth.make_noise_image(40)).cuda()[None]
eng.set_images(img, 0)
boxes, _ = synth.make_boxes(40, 4)
b = torch.from_numpy(boxes).cuda()
n, T = 4, 7
bufs = {"TOK0": n*T*256*4, "Q": n*T*256*4, "QP": n*T*128*4, "O128": n*T*128*4, "MH": n*T*2048*4, "KT": n*T*128*4, "VT": n*T*128*4,
        "HYPER": n*4*32*4, "K0F": 4096*256*4, "K0E": 4096*256*2, "KF": n*4096*256*4, "KE": n*4096*256*2, "KVQ": n*4096*256*2,
        "OI": n*4096*128*2, "U1raw": n*4096*256*4, "U1": n*4096*256*2, "U2": n*4096*4*128*2}
def run():
    m, q, l = eng.predict(0, b, None, None, None, False, False, (1024, 1024), (1024, 1024))
    out = {}
    s = torch.cuda.current_stream().cuda_stream
    for k, nb in bufs.items():
        t = torch.empty(nb, dtype=torch.uint8, device="cuda")
        assert lib.samrs_debug_copy_buffer(eng.handle, k.encode(), t.data_ptr(), nb, s) == 0
        out[k] = t
    torch.cuda.synchronize()
    out["low"] = l.clone().view(torch.uint8).flatten()
    return out
r = [run() for _ in range(4)]
for k in list(bufs) + ["low"]:
    nd = [int((r[i][k] != r[0][k]).sum().item()) for i in range(1, 4)]
    print(f"{k:6s} bytes differing vs run 0: {nd}")
