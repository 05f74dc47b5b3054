"""COCO run-length encoding of binary masks (what ``pycocotools.mask.encode`` produces and the
reference stores in ``ins/*.pkl``: Generate Dataset/main_sam_hbox_semantic.py:201-205).

pycocotools is not a dependency here, so the two pieces are restated:
  * counts: column-major (Fortran order) run lengths, starting with the run of zeros -- the in-tree
    statement of the same thing is Generate Dataset/segment_anything/utils/amg.py:107-135
    (``mask_to_rle_pytorch``);
  * the compressed ASCII string (cocoapi ``rleToString`` / ``rleFrString``): each count is
    delta-coded against the count two positions earlier (from the 4th on), split into 5-bit groups,
    bit 5 = "more groups follow", +48 to land in printable ASCII.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def mask_to_counts(mask: np.ndarray) -> List[int]:
    """Uncompressed RLE counts of a 2-D boolean mask (column-major, first run = zeros)."""
    flat = np.asfortranarray(mask.astype(bool)).reshape(-1, order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(bounds).tolist()
    return counts if not flat[0] else [0] + counts


def counts_to_string(counts: List[int]) -> str:
    out = []
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def counts_to_string_np(counts: np.ndarray) -> str:
    """`counts_to_string` vectorised over the whole count array (numpy): the same delta coding and 5-bit groups, a few
    passes over small arrays instead of a Python loop per count (the device path, samrs_rle_encode, makes even this
    unnecessary in the generation loop; this is the host fallback)."""
    c = np.asarray(counts, dtype=np.int64)
    if c.size == 0:
        return ""
    x = c.copy()
    x[3:] -= c[1:-2]
    groups = []          # per pass: (chars, still-active index array)
    idx = np.arange(x.size)
    pos_len = np.zeros(x.size, dtype=np.int64)
    cur = x
    while idx.size:
        g = cur & 0x1F
        cur = cur >> 5
        more = np.where((g & 0x10) != 0, cur != -1, cur != 0)
        groups.append((idx, (g | (more.astype(np.int64) << 5)) + 48))
        pos_len[idx] += 1
        idx, cur = idx[more], cur[more]
    start = np.concatenate([[0], np.cumsum(pos_len)[:-1]])
    out = np.empty(int(pos_len.sum()), dtype=np.uint8)
    for k, (ix, ch) in enumerate(groups):
        out[start[ix] + k] = ch
    return out.tobytes().decode("ascii")


def string_to_counts(s: str) -> List[int]:
    counts: List[int] = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def encode(mask: np.ndarray) -> Dict:
    """Same dict as ``maskUtils.encode(np.asfortranarray(mask))`` with ``counts`` already decoded to str
    (the reference does ``rle['counts'].decode('ascii')``, main_sam_hbox_semantic.py:202)."""
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": counts_to_string_np(np.asarray(mask_to_counts(mask), dtype=np.int64))}


def decode(rle: Dict) -> np.ndarray:
    h, w = rle["size"]
    counts = string_to_counts(rle["counts"]) if isinstance(rle["counts"], str) else list(rle["counts"])
    flat = np.zeros(h * w, dtype=bool)
    pos, val = 0, False
    for c in counts:
        if val:
            flat[pos:pos + c] = True
        pos += c
        val = not val
    return flat.reshape((h, w), order="F")
