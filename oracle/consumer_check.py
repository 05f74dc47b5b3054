"""TEST INFRASTRUCTURE ONLY -- the downstream consumer of the generated labels (BASELINE.json configs[4]).

What it restates, from ``Pretraining and Finetuning/End_to_End`` of the reference:
  datasets.py:185-232   ``SegmentationDataset.__init__``: ``root/train.txt`` / ``valid.txt`` list the stems; image =
                        ``image_path/<stem><ext_img>``, label = ``label_path/<stem><ext_lbl>`` -- for SAMRS the label path is the
                        generation driver's ``.../hbox_segs_init/gray/`` directory (main_pretrain.py:186-190);
  datasets.py:247-273   ``__getitem__``: ``np.array(Image.open(img))``, ``np.array(Image.open(lbl))`` -> normalised CHW float
                        tensor + HxW label tensor (the albumentations pipeline in between is an identity here: albumentations /
                        torchvision cannot be installed, and the check is about the FILE CONTRACT, not the augmentation);
  main_pretrain.py:206-208  ``DistributedSampler(dataset, num_replicas=world, rank=rank)``;
  main_pretrain.py:60,321   ``nn.CrossEntropyLoss(ignore_index=255)`` over ``classes1 = 18`` logits (:183): every label value
                        must be a class id below the class count or exactly 255.
The UperNet / ViT backbone itself (mmseg, mmengine, timm: not installable, profiles/r02_optional_install_attempt.log) is replaced
by a one-layer per-pixel classifier: what is exercised is that the files ``python -m samrs_amd.generate`` writes load, shard,
batch and train through the reference's read path under DDP -- a loss that is finite, ignores the unlabeled pixels and gives
every rank the same averaged gradient.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch
from PIL import Image

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)      # timm.data.constants, used by datasets.py:238
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
IGNORE_LABEL = 255                                  # main_pretrain.py:60


class SegmentationDataset(torch.utils.data.Dataset):
    """datasets.py:181-273 without the augmentation pipeline."""

    def __init__(self, root: str, image_path: str, label_path: str, ext_img: str = ".png", ext_lbl: str = ".png", flag: str = "trn"):
        def read(name):
            with open(os.path.join(root, name)) as f:
                return [l.strip() for l in f.readlines()]                          # :189-190, :205-206
        trn, val = read("train.txt"), read("valid.txt")
        stems = {"trn": trn, "val": val[-500:], "tes": val}[flag]                  # :219-227
        self.files = [os.path.join(image_path, s + ext_img) for s in stems]        # :198
        self.targets = [os.path.join(label_path, s + ext_lbl) for s in stems]      # :199

    def __len__(self) -> int:
        return len(self.targets)

    def __getitem__(self, i: int) -> Tuple[torch.Tensor, torch.Tensor]:
        image = np.array(Image.open(self.files[i]))                                # :250
        label = np.array(Image.open(self.targets[i]))                              # :251
        if image.ndim != 3 or image.shape[2] != 3 or label.ndim != 2 or label.shape != image.shape[:2] or label.dtype != np.uint8:
            raise ValueError(f"{self.targets[i]}: not an 8-bit single-channel label map of the image's size")
        x = torch.from_numpy(image.astype(np.float32) / 255.0).permute(2, 0, 1)    # ToTensor (:236)
        x = (x - torch.tensor(IMAGENET_DEFAULT_MEAN)[:, None, None]) / torch.tensor(IMAGENET_DEFAULT_STD)[:, None, None]   # :237
        return x, torch.from_numpy(label)                                          # :256


def train_steps(dataset: SegmentationDataset, n_classes: int, rank: int, world: int, steps: int = 2, batch_size: int = 2, seed: int = 0):
    """A DDP training loop in miniature over the reference's sampler + loss.  Returns (losses, flattened weight after the steps,
    indices this rank drew, label histogram this rank saw)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    torch.manual_seed(seed)
    model = torch.nn.Conv2d(3, n_classes, kernel_size=1)                           # stands in for UperNet-ViT-B (not installable)
    ddp = DDP(model) if (dist.is_available() and dist.is_initialized() and world > 1) else model
    sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=False)      # main_pretrain.py:206
    loader = DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=0, drop_last=False)
    criterion = torch.nn.CrossEntropyLoss(ignore_index=IGNORE_LABEL)               # main_pretrain.py:321
    opt = torch.optim.SGD(ddp.parameters(), lr=0.1)
    losses: List[float] = []
    hist = np.zeros(256, dtype=np.int64)
    drawn = list(iter(sampler))
    it = iter(loader)
    for _ in range(steps):
        try:
            x, y = next(it)
        except StopIteration:
            it = iter(loader)
            x, y = next(it)
        bad = (y != IGNORE_LABEL) & (y >= n_classes)
        if bool(bad.any()):
            raise ValueError(f"label values outside 0..{n_classes - 1} and != {IGNORE_LABEL}: {torch.unique(y[bad]).tolist()}")
        hist += np.bincount(y.numpy().ravel(), minlength=256)
        opt.zero_grad()
        loss = criterion(ddp(x), y.long())                                         # the training scripts call .long() on the target
        loss.backward()
        opt.step()
        losses.append(float(loss))
    w = torch.cat([p.detach().flatten() for p in model.parameters()])
    return losses, w, drawn, hist


# ------------------------------------------------------------------------------------------------
# Pinning (round 6): the reader above against the reference's OWN SegmentationDataset on the product's files
# ------------------------------------------------------------------------------------------------
SAMPLE_CLASSES, SAMPLE_IMAGES, SAMPLE_SIDE = 18, 6, 64


def make_sample_dataset(root: str, side: int = 0):
    """A small dataset in the layout main_pretrain.py:186-190 expects, written by the PRODUCT's writers (samrs_amd.generate.write_outputs,
    samrs_amd.tile_io: gray / color PNG + ins pickles; the class maps are synthetic rectangles so that it can be made without a GPU):
    6 tiles of 64 x 64, 18 classes, train.txt = 4 stems, valid.txt = 2.  Returns (image dir, label dir)."""
    from samrs_amd import generate, tile_io
    SAMPLE_SIDE = side or globals()["SAMPLE_SIDE"]          # (side: other tile sizes, e.g. 224 for the reference's ViT-B + UperNet)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    out = os.path.join(root, "hbox_segs_init")
    palette = generate.default_palette(SAMPLE_CLASSES)
    names = [str(i) for i in range(SAMPLE_CLASSES)]
    stems = []
    for i in range(SAMPLE_IMAGES):
        rng = np.random.default_rng(70 + i)
        img = rng.integers(0, 256, (SAMPLE_SIDE, SAMPLE_SIDE, 3), dtype=np.uint8)
        seg = np.full((SAMPLE_SIDE, SAMPLE_SIDE), 255, np.uint8)                    # main_sam_hbox_semantic.py:162
        boxes, labels, areas = [], [], []
        for _ in range(4):
            x0, y0 = rng.integers(0, SAMPLE_SIDE - 8, 2)
            w, h = rng.integers(4, 24, 2)
            lab = int(rng.integers(0, SAMPLE_CLASSES))
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


def reference_dataset(root: str, flag: str):
    """The REFERENCE's SegmentationDataset (datasets.py:182-273, imported through oracle/ref_import.py) on `root`, as
    main_pretrain.py:186-202 constructs it for an UperNet decoder; the albumentations pipeline is the identity (not installable)."""
    import types
    from oracle import ref_import
    ds = ref_import.import_reference_consumer()
    args = types.SimpleNamespace(decoder="upernet")
    return ds.SegmentationDataset(args, SAMPLE_SIDE, root, os.path.join(root, "images"), os.path.join(root, "hbox_segs_init", "gray"),
                                  ext_img=".png", ext_lbl=".png", flag=flag, transform=lambda image, mask: {"image": image, "mask": mask})


def reference_upernet_vit_b(n_classes: int, image_size: int = 224):
    """The reference's OWN segmentation model for BASELINE.json configs[4]: ViT-B + RVSA backbone and UPerHead, assembled as
    `Pretraining and Finetuning/Encoder_Decoder/models.py:81-82,174-186` does (`vit_b_rvsa(args)`; `UPerHead(in_channels =
    encoder.out_channels[1:], channels = encoder.out_channels[2], in_index = (0, 1, 2, 3), dropout_ratio = 0.1, norm_cfg = SyncBN)`;
    head = `Dropout2d(0.1)` + `Conv2d(channels, classes, 1)`) with its forward (`:277-280`: `head(decoder(*encoder(x)))`).  Both classes are
    imported from the reference tree (oracle/ref_import.import_reference_upernet); only where that tree exists."""
    import types
    from oracle import ref_import
    bb, up = ref_import.import_reference_upernet()
    args = types.SimpleNamespace(image_size=image_size, use_ckpt="False")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = bb.vit_b_rvsa(args, inchannels=3)                                                     # models.py:81-82
            self.decoder = up.UPerHead(in_channels=self.encoder.out_channels[1:], channels=self.encoder.out_channels[2],
                                       in_index=(0, 1, 2, 3), dropout_ratio=0.1, norm_cfg=dict(type="SyncBN", requires_grad=True))   # :176-182
            self.semseghead_1 = torch.nn.Sequential(torch.nn.Dropout2d(0.1),
                                                    torch.nn.Conv2d(self.encoder.out_channels[2], n_classes, kernel_size=1))     # :184-187

        def forward(self, x):
            return self.semseghead_1(self.decoder(*self.encoder(x)))                                            # :277-280

    return Net()


def train_steps_model(model, dataset, rank: int, world: int, steps: int = 2, batch_size: int = 1, lr: float = 0.01):
    """`train_steps` with a given model (the reference's UperNet-ViT-B): DistributedSampler + CrossEntropyLoss(ignore_index=255) + DDP."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    # find_unused_parameters=True: as the reference wraps it (Encoder_Decoder/main_pretrain.py:445, End_to_End/main_pretrain.py:464)
    ddp = DDP(model, find_unused_parameters=True) if (dist.is_available() and dist.is_initialized() and world > 1) else model
    sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=False)
    loader = DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=0, drop_last=False)
    criterion = torch.nn.CrossEntropyLoss(ignore_index=IGNORE_LABEL)
    opt = torch.optim.SGD(ddp.parameters(), lr=lr)
    losses, it = [], iter(loader)
    for _ in range(steps):
        try:
            x, y = next(it)
        except StopIteration:
            it = iter(loader)
            x, y = next(it)
        out = ddp(x)
        loss = criterion(out, y.long())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    probe = torch.cat([p.detach().flatten()[:64] for p in list(model.parameters())[:8]])
    return losses, probe, tuple(out.shape)


def make_golden(path: str) -> None:
    """tests/golden/consumer_ref.npz: what the reference's SegmentationDataset returns for every item of the sample dataset
    (flags trn / val / tes): the file lists it built and the (image tensor, label tensor) pairs.  Run where /root/reference exists:
        python -m oracle.consumer_check tests/golden/consumer_ref.npz"""
    import tempfile
    with tempfile.TemporaryDirectory() as root:
        make_sample_dataset(root)
        blob = {}
        for flag in ("trn", "val", "tes"):
            ds = reference_dataset(root, flag)
            blob[f"{flag}_files"] = np.array([os.path.relpath(f, root) for f in ds.files])
            blob[f"{flag}_targets"] = np.array([os.path.relpath(f, root) for f in ds.targets])
            xs, ys = zip(*(ds[i] for i in range(len(ds))))
            blob[f"{flag}_x"] = torch.stack(xs).numpy()
            blob[f"{flag}_y"] = torch.stack(ys).numpy()
        np.savez_compressed(path, **blob)
        print({k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    import sys
    make_golden(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                                   "consumer_ref.npz"))
