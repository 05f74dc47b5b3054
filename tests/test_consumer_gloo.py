"""CPU, world_size 2, gloo: the downstream consumer check (SURVEY.md 8f N4 / BASELINE.json configs[4]) -- files written by the
generation driver's writers (samrs_amd.generate.write_outputs: gray/*.png, color/*.png, ins/*.pkl) go through the reference's
training read path (oracle/consumer_check.py: datasets.py:185-273, DistributedSampler, CrossEntropyLoss(ignore_index=255)) under
DDP.  The GPU suite runs the same consumer on real `generate.run` output (tests/test_parity_gpu.py::test_generation_driver_end_to_end)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from samrs_amd import generate, tile_io

N_CLASSES, N_IMAGES, SIDE = 18, 6, 64


def _make_dataset(root):
    """Synthetic tiles + class maps written by the PRODUCT's writers (the class maps are synthetic: no GPU here)."""
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    out = os.path.join(root, "hbox_segs_init")
    palette = generate.default_palette(N_CLASSES)
    names = [str(i) for i in range(N_CLASSES)]
    stems = []
    for i in range(N_IMAGES):
        rng = np.random.default_rng(70 + i)
        img = rng.integers(0, 256, (SIDE, SIDE, 3), dtype=np.uint8)
        seg = np.full((SIDE, SIDE), 255, np.uint8)                                  # main_sam_hbox_semantic.py:162
        boxes, labels, areas = [], [], []
        for _ in range(4):
            x0, y0 = rng.integers(0, SIDE - 8, 2)
            w, h = rng.integers(4, 24, 2)
            lab = int(rng.integers(0, N_CLASSES))
            seg[y0:y0 + h, x0:x0 + w] = lab
            boxes.append(np.array([x0, y0, x0 + w, y0 + h], np.float32)); labels.append(lab); areas.append(int(w * h))
        stem = f"P{i:04d}"
        tile_io.write_rgb(os.path.join(root, "images", stem + ".png"), img)
        generate.write_outputs(out, stem, seg, None, np.stack(boxes), np.asarray(labels), np.asarray(areas), palette, names)
        stems.append(stem)
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join(stems[:4]) + "\n")
    with open(os.path.join(root, "valid.txt"), "w") as f:
        f.write("\n".join(stems[4:]) + "\n")
    return os.path.join(root, "images"), os.path.join(out, "gray")


def _worker(rank, world, port, root, out):
    from oracle import consumer_check as cc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = cc.SegmentationDataset(root, os.path.join(root, "images"), os.path.join(root, "hbox_segs_init", "gray"), flag="trn")
    losses, w, drawn, hist = cc.train_steps(ds, N_CLASSES, rank, world, steps=2, batch_size=2)
    out[rank] = (losses, w.numpy().copy(), drawn, hist)
    dist.barrier()
    dist.destroy_process_group()


def test_generated_labels_train_under_ddp(tmp_path):
    root = str(tmp_path)
    _make_dataset(root)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, root, out), nprocs=2, join=True)
    (l0, w0, d0, h0), (l1, w1, d1, h1) = out[0], out[1]
    assert sorted(d0 + d1) == [0, 1, 2, 3]                          # the sampler's shards partition the train list
    assert all(np.isfinite(l0 + l1)) and l0[1] < l0[0] + 1.0
    assert np.allclose(w0, w1)                                      # DDP: both ranks hold the same weights after the steps
    hist = h0 + h1
    assert hist[N_CLASSES:255].sum() == 0 and hist[255] > 0 and hist[:N_CLASSES].sum() > 0   # class ids or the ignore label, nothing else
    # the serial loop (world 1) sees the same files and the same label values
    from oracle import consumer_check as cc
    ds = cc.SegmentationDataset(root, os.path.join(root, "images"), os.path.join(root, "hbox_segs_init", "gray"), flag="tes")
    assert len(ds) == 2
    x, y = ds[0]
    assert x.shape == (3, SIDE, SIDE) and y.shape == (SIDE, SIDE) and y.dtype == torch.uint8
    # a label map that breaks the contract is refused by the consumer (16-bit / colour / wrong size)
    tile_io.write_rgb(os.path.join(root, "hbox_segs_init", "gray", "P0004.png"), np.zeros((SIDE, SIDE, 3), np.uint8))
    try:
        ds[0]
        raise AssertionError("an RGB label map must be refused")
    except ValueError:
        pass
