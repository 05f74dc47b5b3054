"""TEST INFRASTRUCTURE ONLY -- import the real reference ``segment_anything`` read-only.

Works only where ``/root/reference`` exists (the authoring container); the GPU box does not
have it, so nothing under ``tests/ -m gpu``, ``smoke()`` or ``bench.py`` may call this.
It exists to (a) validate ``oracle/sam_oracle.py`` against the reference itself and (b)
generate the golden fixtures committed under ``tests/golden/`` (``oracle/make_golden.py``).

``torchvision`` is not installed and the reference imports it at module import time
(Generate Dataset/segment_anything/utils/transforms.py:10, automatic_mask_generator.py:9),
so a stub is injected.  For 1024-long-side inputs that are already 1024 the
stubbed ``resize`` is the identity.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = "/root/reference/Generate Dataset"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "segment_anything"))


def _install_torchvision_stub() -> None:
    if "torchvision" in sys.modules:
        return
    import numpy as np
    from PIL import Image

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")

    def to_pil_image(a):
        return Image.fromarray(np.asarray(a))

    def resize(img, size):
        h, w = size
        return img.resize((w, h), Image.BILINEAR)

    fn.to_pil_image, fn.resize = to_pil_image, resize
    boxes.batched_nms = boxes.box_area = None
    tv.transforms, tr.functional, tv.ops, ops.boxes = tr, fn, ops, boxes
    for name, mod in [("torchvision", tv), ("torchvision.transforms", tr),
                      ("torchvision.transforms.functional", fn), ("torchvision.ops", ops),
                      ("torchvision.ops.boxes", boxes)]:
        sys.modules[name] = mod


def import_reference():
    """Returns the reference's ``segment_anything`` module (never writes into /root/reference)."""
    if not reference_available():
        raise RuntimeError("reference tree not present on this machine")
    sys.dont_write_bytecode = True
    _install_torchvision_stub()
    # this repo ships an import-name alias package `segment_anything` (the drop-in boundary): if it is already imported,
    # `import segment_anything` would hand the ENGINE back and golden generation would validate the engine against itself
    for name in [n for n in sys.modules if n == "segment_anything" or n.startswith("segment_anything.")]:
        if not (getattr(sys.modules[name], "__file__", None) or "").startswith(REF_ROOT):
            del sys.modules[name]
    if sys.path[:1] != [REF_ROOT]:
        sys.path.insert(0, REF_ROOT)
    import segment_anything  # noqa: E402  (the reference's, not ours)
    assert (segment_anything.__file__ or "").startswith(REF_ROOT), segment_anything.__file__
    return segment_anything


def build_reference_sam(cfg, state_dict):
    """The reference ``Sam`` module for ``cfg`` with ``state_dict`` loaded strictly."""
    sa = import_reference()
    from segment_anything.build_sam import _build_sam  # type: ignore

    sam = _build_sam(
        encoder_embed_dim=cfg.embed_dim,
        encoder_depth=cfg.depth,
        encoder_num_heads=cfg.num_heads,
        encoder_global_attn_indexes=list(cfg.global_attn_indexes),
        checkpoint=None,
    )
    sam.load_state_dict(state_dict, strict=True)
    sam.eval()
    return sa, sam


# ------------------------------------------------------------------------------------------------
# The downstream consumer: `Pretraining and Finetuning/End_to_End/datasets.py`
# ------------------------------------------------------------------------------------------------
PF_ROOT = "/root/reference/Pretraining and Finetuning/End_to_End"


def consumer_available() -> bool:
    return os.path.isfile(os.path.join(PF_ROOT, "datasets.py"))


def import_reference_consumer():
    """The reference's ``datasets`` module (``SegmentationDataset``, datasets.py:182-273) imported read-only.

    It imports cv2, skimage, torchvision.transforms, mmcv, mmengine and mmseg at module import (datasets.py:2-17); none of them is
    installable here.  What ``SegmentationDataset`` itself USES of them for a non-mask2former decoder is
    ``T.Compose([T.ToPILImage(), T.ToTensor(), T.Normalize(mean, std)])`` (:233-238) -- stubbed below with torchvision's documented
    semantics (uint8 HWC -> float CHW / 255; (x - mean) / std) -- everything else (file lists from train.txt / valid.txt, the
    ``val[-500:]`` rule, PIL reads, the order of the calls, the label pass-through) is the reference's own code running."""
    if not consumer_available():
        raise RuntimeError("reference tree not present on this machine")
    sys.dont_write_bytecode = True
    import numpy as np
    import torch
    from PIL import Image

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, a):
            return Image.fromarray(np.asarray(a))

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic)
            return torch.from_numpy(np.array(a, copy=True)).permute(2, 0, 1).to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean)[:, None, None], torch.tensor(std)[:, None, None]

        def __call__(self, t):
            return (t - self.mean) / self.std

    class _Base:
        def __init__(self, *a, **k):
            pass

    stub_names = ("torchvision", "torchvision.transforms", "cv2", "skimage", "skimage.io", "mmcv", "mmcv.transforms", "mmcv.transforms.base",
                  "mmengine", "mmengine.structures", "mmseg", "mmseg.structures")
    keep = {k: sys.modules.get(k) for k in stub_names}
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=Compose, ToPILImage=ToPILImage, ToTensor=ToTensor, Normalize=Normalize)
    mod("cv2")
    sk = mod("skimage")
    sk.io = mod("skimage.io")
    mod("mmcv")
    mt = mod("mmcv.transforms", to_tensor=torch.as_tensor)
    mt.base = mod("mmcv.transforms.base", BaseTransform=_Base)
    mod("mmengine")
    mod("mmengine.structures", PixelData=_Base, BaseDataElement=_Base)
    mod("mmseg")
    mod("mmseg.structures", SegDataSample=_Base)
    import importlib.util
    spec = importlib.util.spec_from_file_location("samrs_reference_pf_datasets", os.path.join(PF_ROOT, "datasets.py"))
    ds = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ds)
    for k, v in keep.items():       # the stand-ins must not outlive the import: the SAM-side stub (import_reference) wants its own torchvision
        if v is None:               # back, and `import cv2` elsewhere (samrs_amd.transforms.resolve_fill_rule) must keep failing honestly
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return ds


# ------------------------------------------------------------------------------------------------
# UperNet + ViT-B (RVSA) of the reference's segmentation training: `Pretraining and Finetuning/Encoder_Decoder`
# ------------------------------------------------------------------------------------------------
ED_ROOT = "/root/reference/Pretraining and Finetuning/Encoder_Decoder"


def upernet_available() -> bool:
    return os.path.isfile(os.path.join(ED_ROOT, "upernet_mmseg_30.py"))


def import_reference_upernet():
    """(backbone module, head module): the reference's OWN `backbone/vit_win_rvsa_v3_wsz7.py` (ViT-B + RVSA, `vit_b_rvsa`) and
    `upernet_mmseg_30.py` (`UPerHead`), imported read-only, file by file (`models.py` itself imports every backbone of the repository, incl.
    CUDA-only ops).  What they need from packages that cannot be installed here, and what stands in:
      timm.models.layers  drop_path / to_2tuple / trunc_normal_      -> their documented one-liners (torch.nn.init.trunc_normal_)
      mmengine.dist.get_dist_info                                    -> (rank, world) of torch.distributed
      mmengine.model.BaseModule                                      -> torch.nn.Module
      mmcv.cnn.ConvModule                                            -> conv -> norm -> activation with mmcv's `bias='auto'` rule;
                                                                        norm_cfg type 'SyncBN' becomes BatchNorm2d (SyncBatchNorm has no
                                                                        CPU / gloo implementation: the statistics stay per rank)
      mmseg.structures.build_pixel_sampler, mmseg.utils types        -> unused by the forward pass
    The model code -- attention with rotated varied-size windows, the FPN neck, PPM, the UPerNet fusion -- is the reference's, unmodified."""
    if not upernet_available():
        raise RuntimeError("reference tree not present on this machine")
    sys.dont_write_bytecode = True
    import importlib.util
    import torch
    import torch.nn as nn

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def drop_path(x, drop_prob: float = 0., training: bool = False):
        if not drop_prob or not training:
            return x
        keep = 1 - drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x.div(keep) * mask

    def to_2tuple(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

    def get_dist_info():
        import torch.distributed as dist
        return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                     conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **kw):
            super().__init__()
            with_norm = norm_cfg is not None
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                  bias=(not with_norm) if bias == "auto" else bias)
            self.bn = nn.BatchNorm2d(out_channels) if with_norm else None
            if with_norm and not norm_cfg.get("requires_grad", True):
                for p in self.bn.parameters():
                    p.requires_grad = False
            self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

        def forward(self, x):
            x = self.conv(x)
            if self.bn is not None:
                x = self.bn(x)
            return self.activate(x) if self.activate is not None else x

    stub_names = ("timm", "timm.models", "timm.models.layers", "mmengine", "mmengine.dist", "mmengine.model", "mmcv", "mmcv.cnn", "mmseg",
                  "mmseg.structures", "mmseg.utils")
    keep = {k: sys.modules.get(k) for k in stub_names}
    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", drop_path=drop_path, to_2tuple=to_2tuple, trunc_normal_=nn.init.trunc_normal_)
    mod("mmengine")
    mod("mmengine.dist", get_dist_info=get_dist_info)
    mod("mmengine.model", BaseModule=BaseModule)
    mod("mmcv")
    mod("mmcv.cnn", ConvModule=ConvModule)
    mod("mmseg")
    mod("mmseg.structures", build_pixel_sampler=lambda *a, **k: None)
    mod("mmseg.utils", ConfigType=dict, SampleList=list)
    out = []
    for name, rel in (("samrs_reference_vit_rvsa", "backbone/vit_win_rvsa_v3_wsz7.py"), ("samrs_reference_upernet", "upernet_mmseg_30.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ED_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        out.append(m)
    for k, v in keep.items():       # (the imported files hold what they imported; the stand-ins leave sys.modules again)
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return tuple(out)
