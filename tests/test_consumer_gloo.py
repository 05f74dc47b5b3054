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

from oracle import consumer_check as _cc

N_CLASSES, N_IMAGES, SIDE = _cc.SAMPLE_CLASSES, _cc.SAMPLE_IMAGES, _cc.SAMPLE_SIDE
_make_dataset = _cc.make_sample_dataset


def _worker(rank, world, port, root, out, use_reference=False):
    from oracle import consumer_check as cc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if use_reference:        # the REFERENCE's own SegmentationDataset class (datasets.py:182-273) reads the product's files
        ds = cc.reference_dataset(root, "trn")
    else:
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


def test_restated_reader_equals_the_reference_dataset_fixture(tmp_path):
    """Round 6: the reader of oracle/consumer_check.py is pinned to the reference's OWN SegmentationDataset --
    tests/golden/consumer_ref.npz holds what `Pretraining and Finetuning/End_to_End/datasets.py:182-273` (imported in the authoring
    container by oracle/ref_import.import_reference_consumer) returned for every item of the sample dataset the product's writers
    produce: file lists per flag (train.txt / valid.txt, the val[-500:] rule), image tensors and label tensors.  The restated
    reader must build the same lists and return the same tensors from freshly written files (the writers are deterministic)."""
    root = str(tmp_path)
    _make_dataset(root)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "consumer_ref.npz"))
    for flag in ("trn", "val", "tes"):
        ds = _cc.SegmentationDataset(root, os.path.join(root, "images"), os.path.join(root, "hbox_segs_init", "gray"), flag=flag)
        assert [os.path.relpath(f, root) for f in ds.files] == list(gold[f"{flag}_files"])
        assert [os.path.relpath(f, root) for f in ds.targets] == list(gold[f"{flag}_targets"])
        for i in range(len(ds)):
            x, y = ds[i]
            assert np.array_equal(y.numpy(), gold[f"{flag}_y"][i])                      # labels: bit for bit
            assert np.allclose(x.numpy(), gold[f"{flag}_x"][i], rtol=0, atol=1e-6)      # images: fp32 (x / 255 - mean) / std


def test_reference_dataset_class_trains_on_generated_files_under_ddp(tmp_path):
    """Where the reference tree is present (the authoring container): two DDP-gloo iterations with the REFERENCE's SegmentationDataset
    itself over the files the product's writers wrote, and item-by-item equality with the restated reader.  (UperNet-ViT-B needs
    mmseg / mmengine / timm, which cannot be installed: the model stays the one-layer stand-in.)"""
    import pytest
    from oracle import ref_import
    if not ref_import.consumer_available():
        pytest.skip("/root/reference is not on this machine")
    root = str(tmp_path)
    _make_dataset(root)
    ref = _cc.reference_dataset(root, "trn")
    mine = _cc.SegmentationDataset(root, os.path.join(root, "images"), os.path.join(root, "hbox_segs_init", "gray"), flag="trn")
    assert ref.files == mine.files and ref.targets == mine.targets and len(ref) == 4
    for i in range(len(ref)):
        (xr, yr), (xm, ym) = ref[i], mine[i]
        assert torch.equal(yr, ym) and torch.allclose(xr, xm, rtol=0, atol=1e-6)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, root, out, True), nprocs=2, join=True)
    (l0, w0, d0, h0), (l1, w1, d1, h1) = out[0], out[1]
    assert sorted(d0 + d1) == [0, 1, 2, 3] and all(np.isfinite(l0 + l1)) and np.allclose(w0, w1)
    hist = h0 + h1
    assert hist[N_CLASSES:255].sum() == 0 and hist[255] > 0 and hist[:N_CLASSES].sum() > 0


def _upernet_worker(rank, world, port, root, out):
    from oracle import consumer_check as cc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                             # the same initial weights on both ranks (DDP broadcasts rank 0's anyway)
    ds = cc.reference_dataset(root, "trn")
    model = cc.reference_upernet_vit_b(N_CLASSES, image_size=224)
    losses, probe, shape = cc.train_steps_model(model, ds, rank, world, steps=2, batch_size=2)     # (BatchNorm over the 1 x 1 PPM bin needs > 1 sample)
    out[rank] = (losses, probe.numpy().copy(), shape)
    dist.barrier()
    dist.destroy_process_group()


def test_reference_upernet_vit_b_trains_on_generated_files_under_ddp(tmp_path):
    """BASELINE.json configs[4] / SURVEY 8f N4 with the reference's OWN model: where the reference tree exists (the authoring container),
    `Pretraining and Finetuning/Encoder_Decoder`'s ViT-B + RVSA backbone and UPerHead (imported file by file through
    oracle/ref_import.import_reference_upernet; mmcv / mmengine / timm are replaced by the few lines those files use of them) are
    assembled as models.py:81-82,174-187 does, fed by the reference's own SegmentationDataset reading the files the PRODUCT's writers
    wrote (224 x 224 tiles: the ViT's 7 x 7 windows need a 14 x 14 token grid), and trained for two iterations under DDP on two gloo
    ranks with the reference's loss (`CrossEntropyLoss(ignore_index=255)`, main_pretrain.py:321): finite losses, logits of the
    label map's size, identical weights on both ranks afterwards.  (8 GPUs / RCCL / mmseg's training loop: not available here.)"""
    import pytest
    from oracle import ref_import
    if not (ref_import.upernet_available() and ref_import.consumer_available()):
        pytest.skip("/root/reference is not on this machine")
    root = str(tmp_path)
    _cc.make_sample_dataset(root, side=224)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_upernet_worker, args=(2, port, root, out), nprocs=2, join=True)
    (l0, p0, s0), (l1, p1, s1) = out[0], out[1]
    assert s0 == s1 == (2, N_CLASSES, 224, 224)
    assert all(np.isfinite(l0 + l1)) and 0.5 < l0[0] < 10.0           # ~ log(18) at initialisation
    assert np.allclose(p0, p1)                                        # DDP: the same weights on both ranks after the two steps
