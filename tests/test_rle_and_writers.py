"""CPU: COCO RLE restatement (pycocotools is not installed) + the on-disk output contract of the
generation driver (Generate Dataset/main_sam_hbox_semantic.py:195-216)."""
import os
import pickle

import numpy as np
import pytest

from samrs_amd import generate, rle


def test_counts_are_column_major_and_start_with_zeros():
    m = np.array([[0, 1], [1, 1], [0, 0]], dtype=bool)          # columns: [0,1,0], [1,1,0]
    assert rle.mask_to_counts(m) == [1, 1, 1, 2, 1]
    assert rle.mask_to_counts(np.ones((2, 2), bool)) == [0, 4]
    assert rle.mask_to_counts(np.zeros((2, 3), bool)) == [6]


def test_known_cocoapi_strings():
    # hand-computed with the cocoapi rleToString rule (5-bit groups, +48, delta vs counts[i-2] for i > 2)
    assert rle.counts_to_string([6]) == "6"
    assert rle.counts_to_string([0, 4]) == "04"
    assert rle.counts_to_string([1, 1, 1, 2, 1]) == "11110"     # 4th: 2-1 = 1 ; 5th: 1-1 = 0
    assert rle.counts_to_string([100]) == "T3"                  # 100 = 0b00011_00100 -> (4|0x20)+48='T', 3+48='3'
    assert rle.counts_to_string([5, 3, 5, 1]) == "535N"         # 4th: 1-3 = -2 -> single group 0b11110 (sign bit set, x == -1 stops)


def test_counts_match_the_reference_mask_to_rle_pytorch():
    """Pins the counts half against the reference's own statement of COCO RLE, `mask_to_rle_pytorch`
    (Generate Dataset/segment_anything/utils/amg.py:107-135), imported read-only: same counts for random, blob, all-zero,
    all-one and first-pixel-set masks, square and ragged.  (Runs where /root/reference exists; the string half has no
    reference on this machine -- pycocotools is not installable -- and stays a restatement of cocoapi's rleToString.)"""
    from oracle import ref_import
    if not ref_import.reference_available():
        pytest.skip("reference tree not present on this machine")
    import torch
    ref_import.import_reference()
    from segment_anything.utils.amg import mask_to_rle_pytorch            # the reference's
    rng = np.random.default_rng(11)
    masks = []
    for shape in [(64, 64), (37, 91), (128, 50)]:
        for p in (0.0, 0.02, 0.5, 1.0):
            m = rng.random(shape) < p
            if p == 0.02:
                m[shape[0] // 4: shape[0] // 2, 3:] = True
            masks.append(m)
        m = np.zeros(shape, bool); m[0, 0] = True; masks.append(m)
        m = np.ones(shape, bool); m[-1, -1] = False; masks.append(m)
    for m in masks:
        ref = mask_to_rle_pytorch(torch.from_numpy(m)[None])[0]
        assert ref["size"] == list(m.shape)
        assert rle.mask_to_counts(m) == ref["counts"], m.shape
        # and the vectorised string coder agrees with the scalar restatement
        assert rle.counts_to_string_np(np.asarray(ref["counts"], dtype=np.int64)) == rle.counts_to_string(ref["counts"])


@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (64, 64), (250, 333)])
def test_roundtrip_random_masks(shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    for p in (0.0, 0.03, 0.5, 0.97, 1.0):
        m = rng.random(shape) < p
        # blobs, not only salt and pepper: long runs exercise multi-group counts and negative deltas
        if shape[0] > 8:
            m[shape[0] // 4: shape[0] // 2] = True
        r = rle.encode(m)
        assert r["size"] == list(shape) and isinstance(r["counts"], str) and r["counts"].isascii()
        assert rle.string_to_counts(r["counts"]) == rle.mask_to_counts(m)
        assert np.array_equal(rle.decode(r), m)
        c = np.asarray(rle.mask_to_counts(m), dtype=np.int64)
        assert rle.counts_to_string_np(c) == rle.counts_to_string(c.tolist())


def test_writer_contract(tmp_path):
    from PIL import Image
    seg = np.full((16, 20), 255, np.uint8)
    seg[2:6, 3:9] = 4
    masks = np.zeros((2, 16, 20), bool)
    masks[0, 2:6, 3:9] = True
    boxes = np.array([[3, 2, 8, 5], [0, 0, 1, 1]], np.float32)
    labels = np.array([4, 7])
    areas = np.array([24, 0])
    names = [f"c{i}" for i in range(18)]
    pal = generate.default_palette(18)
    generate.write_outputs(str(tmp_path), "P0001", seg, masks, boxes, labels, areas, pal, names)
    g = np.array(Image.open(tmp_path / "gray" / "P0001.png"))
    c = np.array(Image.open(tmp_path / "color" / "P0001.png"))
    assert g.dtype == np.uint8 and np.array_equal(g, seg)
    assert (c[0, 0] == 255).all() and (c[3, 4] == pal[4]).all()
    info = pickle.load(open(tmp_path / "ins" / "P0001.pkl", "rb"))
    assert [sorted(d) for d in info] == [["bbox", "category", "label", "mask", "size"]] * 2
    assert info[0]["label"] == 4 and info[0]["category"] == "c4" and info[0]["size"] == 24
    assert np.array_equal(rle.decode(info[0]["mask"]), masks[0]) and info[1]["size"] == 0
    # what Generate Dataset/statistic.py:15-21 does with these files
    pix = {i: 0 for i in range(18)}
    for d in info:
        if d["size"] > 0:
            pix[d["label"]] += d["size"]
    assert pix[4] == 24 and pix[7] == 0


def test_gray_labels_satisfy_the_training_consumer(tmp_path):
    """N4 (label-format half): what `Pretraining and Finetuning/End_to_End/datasets.py:251-254` does with a generated
    label -- `np.array(Image.open(lbl_path))` fed as the mask of the augmentation pipeline -- and what
    `main_finetune.py:263` requires of it (class ids < n_classes, 255 = ignored by CrossEntropyLoss)."""
    import torch
    from PIL import Image
    n_classes = 18
    rng = np.random.default_rng(0)
    seg = np.full((64, 80), 255, dtype=np.uint8)                      # main_sam_hbox_semantic.py:162
    seg[10:30, 5:50] = 3
    seg[25:60, 40:70] = 17                                            # later box wins (:195-199)
    masks = np.stack([seg == 3, seg == 17])
    boxes = np.array([[5, 10, 49, 29], [40, 25, 69, 59]], dtype=np.float32)
    labels = np.array([3, 17])
    areas = masks.reshape(2, -1).sum(1)
    generate.write_outputs(str(tmp_path), "t0", seg, masks, boxes, labels, areas, generate.default_palette(n_classes),
                           [str(i) for i in range(n_classes)])
    label = np.array(Image.open(os.path.join(tmp_path, "gray", "t0.png")))          # datasets.py:252
    assert label.dtype == np.uint8 and label.shape == seg.shape and np.array_equal(label, seg)
    vals = set(np.unique(label).tolist())
    assert vals <= set(range(n_classes)) | {255}
    logits = torch.from_numpy(rng.standard_normal((1, n_classes, *seg.shape)).astype(np.float32))
    loss = torch.nn.CrossEntropyLoss(ignore_index=255)(logits, torch.from_numpy(label.astype(np.int64))[None])   # main_finetune.py:263
    assert torch.isfinite(loss)
    # statistic.py:12-21 reads the pickles back: size == mask area, label is the class id
    info = pickle.load(open(os.path.join(tmp_path, "ins", "t0.pkl"), "rb"))
    assert [e["label"] for e in info] == [3, 17] and [e["size"] for e in info] == areas.tolist()
    assert np.array_equal(rle.decode(info[1]["mask"]), masks[1])


def test_rle_string_known_answers(golden_dir):
    """cocoapi ``rleToString`` known answers (VERDICT r03 "missing" 5: the string half of samrs_amd/rle.py was pinned to nothing
    committed; pycocotools cannot be installed here).  tests/golden/coco_rle_known_answers.json holds the published example --
    size [9, 10], counts [6,1,40,4,5,4,5,4,21] -> "61X13mN000`0" -- and hand-derived cases, one rule of maskApi.c:rleToString each:
      * a count below 16 is one group: 9 -> chr(9 + 48) = "9"; a mask that starts with a one has a leading zero count: "0..";
      * 40 = 0b01_01000: low group 8, rest 1 != 0 -> continuation bit: chr(8 + 32 + 48) = "X", then chr(1 + 48) = "1";
      * from the 4th count on the value coded is counts[i] - counts[i - 2]: [1, 2, 3, 1] -> 1 - 2 = -1 = ...11111: group 31 has bit 4
        set and the rest is -1 -> no continuation: chr(31 + 48) = "O";
      * the published example's 5th count: 5 - 40 = -35 -> group 29 (bit 4 set), rest -2 != -1 -> continuation: chr(29 + 32 + 48) = "m",
        then group 30, rest -1 -> chr(30 + 48) = "N";
      * 2^20 (an empty 1024^2 mask): four zero groups with continuation ("P" = chr(32 + 48)) and a final 1: "PPPP1".
    Both coders and the decoder of rle.py, and encode() / decode() on the masks the counts describe."""
    import json
    vec = json.load(open(os.path.join(golden_dir, "coco_rle_known_answers.json")))["vectors"]
    assert any(v["name"] == "published" for v in vec)
    for v in vec:
        h, w = v["size"]
        assert sum(v["counts"]) == h * w
        assert rle.counts_to_string(v["counts"]) == v["string"], v["name"]
        assert rle.counts_to_string_np(np.asarray(v["counts"])) == v["string"], v["name"]
        assert rle.string_to_counts(v["string"]) == v["counts"], v["name"]
        mask = rle.decode({"size": v["size"], "counts": v["counts"]})
        assert rle.mask_to_counts(mask) == v["counts"]
        assert rle.encode(mask) == {"size": v["size"], "counts": v["string"]}
        assert np.array_equal(rle.decode({"size": v["size"], "counts": v["string"]}), mask)
