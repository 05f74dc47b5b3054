import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import samrs_amd
from samrs_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny"
sam = samrs_amd.sam_model_registry[name](precision="f16", max_prompts=32, max_images=1).to("cuda")
eng = sam.engine
img = torch.as_tensor(synth.make_noise_image(40)).cuda()[None]
eng.set_images(img, 0)
boxes, _ = synth.make_boxes(40, 32)
b = torch.from_numpy(boxes).cuda()
size = (1024, 1024)
def run(bb):
    m, q, l = eng.predict(0, bb, None, None, None, False, False, size, size)
    torch.cuda.synchronize()
    return m.clone(), q.clone(), l.clone()
m1, q1, l1 = run(b); m2, q2, l2 = run(b)
s = l1.std().item()
print("repeat same call: lowres max diff/std %.2e, iou diff %.2e" % ((l1 - l2).abs().max().item() / s, (q1 - q2).abs().max().item()))
for n in (1, 2, 12, 20, 31):
    m, q, l = run(b[:n])
    d = (l - l1[:n]).abs().flatten(1).max(1).values / s
    print(f"first {n:2d} boxes alone vs inside 32: max diff/std {d.max().item():.2e}; iou-pred diff {(q - q1[:n]).abs().max().item():.2e}; per-box", [f"{x:.1e}" for x in d[:6].tolist()])
m, q, l = run(b[20:])
d = (l - l1[20:]).abs().flatten(1).max(1).values / s
print("boxes 20..31 alone:", f"{d.max().item():.2e}")
