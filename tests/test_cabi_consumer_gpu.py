"""GPU: the drop-in boundary used from compiled code.  tests/cabi/cabi_consumer.cpp includes ONLY include/samrs_hip.h, links libsamrs_hip.so,
and replays the reference driver's per-image sequence (build from a state dict, set_image, predict on the boxes, ordered painting) with
hipMalloc'ed buffers and its own hipStream_t -- no Python, no torch in the process.  Its outputs must equal the Python host's (which goes
through the same library by ctypes) bit for bit."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from samrs_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_from_a_compiled_consumer(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc on this machine")
    import samrs_amd
    csrc = os.path.join(ROOT, "samrs_amd", "csrc")
    exe = str(tmp_path / "cabi_consumer")
    r = subprocess.run(["hipcc", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cabi", "cabi_consumer.cpp"), "-I", os.path.join(ROOT, "include"),
                        "-L", csrc, "-lsamrs_hip", "-Wl,-rpath," + csrc, "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = synth.CONFIGS["vit_tiny"]
    sd = synth.make_state_dict(cfg, 11)
    d = str(tmp_path)
    off, lines, chunks = 0, [], []
    for name, t in sd.items():
        a = np.ascontiguousarray(t.to(torch.float32).numpy())
        lines.append(f"{name} {a.ndim} " + " ".join(str(s) for s in a.shape) + f" {off}")
        chunks.append(a.ravel())
        off += a.size
    open(os.path.join(d, "manifest.txt"), "w").write("\n".join(lines) + "\n")
    np.concatenate(chunks).astype(np.float32).tofile(os.path.join(d, "weights.bin"))
    g = cfg.global_attn_indexes
    open(os.path.join(d, "config.txt"), "w").write(f"{cfg.embed_dim} {cfg.depth} {cfg.num_heads} {len(g)} " + " ".join(str(x) for x in g) +
                                                    f" {cfg.img_size} {cfg.patch_size} {cfg.window_size} {cfg.out_chans}\n")
    img = synth.make_image(3)
    boxes, labels = synth.make_boxes(3, 12)                                   # 12 boxes > the consumer's max_prompts = 8: chunked inside samrs_predict
    img.tofile(os.path.join(d, "image.u8"))
    boxes.astype(np.float32).tofile(os.path.join(d, "boxes.f32"))             # 1024^2 tile: the input frame is the original frame
    labels.astype(np.int32).tofile(os.path.join(d, "labels.i32"))
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "12 boxes decoded and painted" in r.stdout
    # the Python host on the same inputs (same library, through ctypes)
    sam = samrs_amd.sam_model_registry["vit_tiny"](state_dict=sd, precision="f16", max_prompts=8, max_points=1).to("cuda")
    pred = samrs_amd.SamPredictor(sam)
    pred.set_image(img)
    tb = pred.transform.apply_boxes_torch(torch.from_numpy(boxes).cuda(), img.shape[:2])
    masks, iou, _ = pred.predict_torch(None, None, tb, None, multimask_output=False)
    seg = torch.full(img.shape[:2], 255, dtype=torch.uint8, device="cuda")
    areas = sam.engine.paint(masks[:, 0], torch.from_numpy(labels.astype(np.int32)), seg)
    m_c = np.fromfile(os.path.join(d, "masks.u8"), dtype=np.uint8).reshape(12, 1024, 1024)
    assert np.array_equal(m_c.astype(bool), masks[:, 0].cpu().numpy())
    assert np.array_equal(np.fromfile(os.path.join(d, "seg.u8"), dtype=np.uint8).reshape(1024, 1024), seg.cpu().numpy())
    assert np.array_equal(np.fromfile(os.path.join(d, "iou.f32"), dtype=np.float32), iou[:, 0].cpu().numpy())
    assert np.array_equal(np.fromfile(os.path.join(d, "areas.i64"), dtype=np.int64), areas.cpu().numpy())
    sam.engine.close()
