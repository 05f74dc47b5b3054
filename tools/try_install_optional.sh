#!/bin/bash
# Records the attempt to obtain the third-party packages the reference's CALLERS use and this image lacks:
#   cv2 (mask-prompt construction, main_sam_rbox_mask_instance.py:125-141), pycocotools (RLE, main_sam_hbox_semantic.py:201),
#   mmengine / mmsegmentation / timm / albumentations (Pretraining and Finetuning/End_to_End, the N4 consumer check).
# No network on the authoring container or on the GPU box: the expected outcome is "no matching distribution".
for pkg in opencv-python-headless pycocotools mmengine mmsegmentation timm albumentations; do
  echo "== pip install $pkg"
  timeout 40 python -m pip install --no-input --disable-pip-version-check "$pkg" 2>&1 | tail -2
done
python - <<'PY'
for m in ("cv2", "pycocotools", "mmengine", "mmseg", "timm", "albumentations"):
    try:
        __import__(m); print(m, "importable")
    except Exception as e:
        print(m, "NOT importable:", type(e).__name__)
PY
