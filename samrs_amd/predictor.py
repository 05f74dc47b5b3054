"""Drop-in ``SamPredictor`` (reference: Generate Dataset/segment_anything/predictor.py:17-271).

Same constructor, methods, attributes, return types and error behaviour; the work happens in
``libsamrs_hip.so`` through ``samrs_amd.engine.Engine``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .build_sam import Sam
from .transforms import ResizeLongestSide


class SamPredictor:
    def __init__(self, sam_model: Sam) -> None:
        if sam_model.engine is None:
            raise RuntimeError("move the model to a HIP device first: sam.to(device='cuda') "
                               "(main_sam_hbox_semantic.py:88); samrs_amd has no CPU path")
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.image_encoder.img_size)
        self.slot = 0
        self.reset_image()

    # -- image side (predictor.py:34-90) ---------------------------------------------------------
    def set_image(self, image: np.ndarray, image_format: str = "RGB") -> None:
        assert image_format in ["RGB", "BGR"], f"image_format must be in ['RGB', 'BGR'], is {image_format}."
        if image_format != self.model.image_format:
            image = image[..., ::-1]
        # ResizeLongestSide.apply_image (utils/transforms.py:26-31) on the device: bit-exact with the
        # reference's PIL bilinear resize, identity for tiles whose long side is already 1024
        t = torch.as_tensor(np.ascontiguousarray(image), device=self.device)            # uint8 HWC
        t = self.transform.apply_image_device(t)
        self._set_hwc(t[None], tuple(image.shape[:2]))

    @torch.no_grad()
    def set_torch_image(self, transformed_image: torch.Tensor, original_image_size: Tuple[int, ...]) -> None:
        s = self.model.image_encoder.img_size
        assert (len(transformed_image.shape) == 4 and transformed_image.shape[1] == 3
                and max(*transformed_image.shape[2:]) == s), f"set_torch_image input must be BCHW with long side {s}."
        # One image per predictor, like the reference: its `features` of a B > 1 input could not be used by
        # predict_torch, whose prompts carry no image index (predictor.py:84-90,168-245).
        assert transformed_image.shape[0] == 1, "set_torch_image takes ONE image (1x3xHxW); batch through Engine.set_images"
        # The reference normalises whatever values it is given (sam.py:167); the engine's input is the uint8 HWC tile
        # every SAMRS driver starts from (set_image, predictor.py:52-58), so 0..255 values are rounded to uint8 here.
        # Non-integer pixel values would be quantised by that rounding -- refuse them instead of answering differently.
        t = transformed_image.to(self.device)
        if t.is_floating_point():
            assert bool((t == t.round()).all()) and float(t.min()) >= 0 and float(t.max()) <= 255, \
                "set_torch_image needs integer pixel values in 0..255 (the engine takes uint8 tiles)"
        hwc = t.permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous()
        self._set_hwc(hwc, tuple(original_image_size))

    def _set_hwc(self, hwc_u8: torch.Tensor, original_size: Tuple[int, int]) -> None:
        self.reset_image()
        self.original_size = tuple(int(v) for v in original_size)
        self.input_size = (int(hwc_u8.shape[1]), int(hwc_u8.shape[2]))
        self.model.engine.set_images(hwc_u8.contiguous(), self.slot)
        self.is_image_set = True

    # -- prompt side (predictor.py:92-245) -------------------------------------------------------
    def predict(self, point_coords: Optional[np.ndarray] = None, point_labels: Optional[np.ndarray] = None,
                box: Optional[np.ndarray] = None, mask_input: Optional[np.ndarray] = None,
                multimask_output: bool = True, return_logits: bool = False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        coords_torch = labels_torch = box_torch = mask_input_torch = None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            point_coords = self.transform.apply_coords(point_coords, self.original_size)
            coords_torch = torch.as_tensor(point_coords, dtype=torch.float, device=self.device)[None, :, :]
            labels_torch = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)[None, :]
        if box is not None:
            box = self.transform.apply_boxes(box, self.original_size)
            box_torch = torch.as_tensor(box, dtype=torch.float, device=self.device)[None, :]
        if mask_input is not None:
            mask_input_torch = torch.as_tensor(mask_input, dtype=torch.float, device=self.device)[None, :, :, :]
        masks, iou_predictions, low_res_masks = self.predict_torch(
            coords_torch, labels_torch, box_torch, mask_input_torch, multimask_output, return_logits=return_logits)
        return (masks[0].detach().cpu().numpy(), iou_predictions[0].detach().cpu().numpy(),
                low_res_masks[0].detach().cpu().numpy())

    @torch.no_grad()
    def predict_torch(self, point_coords: Optional[torch.Tensor], point_labels: Optional[torch.Tensor],
                      boxes: Optional[torch.Tensor] = None, mask_input: Optional[torch.Tensor] = None,
                      multimask_output: bool = True, return_logits: bool = False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        return self.model.engine.predict(self.slot, boxes, point_coords, point_labels, mask_input,
                                         multimask_output, return_logits, self.input_size, self.original_size)

    def get_image_embedding(self) -> torch.Tensor:
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        return self.model.engine.get_embedding(self.slot)

    @property
    def features(self) -> Optional[torch.Tensor]:
        return self.model.engine.get_embedding(self.slot) if self.is_image_set else None

    @property
    def device(self) -> torch.device:
        return self.model.device

    def reset_image(self) -> None:
        self.is_image_set = False
        self.model.engine.reset_image(self.slot)
        self.orig_h = self.orig_w = self.input_h = self.input_w = None
