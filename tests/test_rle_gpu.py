"""GPU: COCO run-length encoding on the device (samrs_rle_encode, replacing maskUtils.encode + .decode('ascii') of
Generate Dataset/main_sam_hbox_semantic.py:201-202) against the host restatement samrs_amd/rle.py, whose counts half is pinned
to the reference's own mask_to_rle_pytorch (tests/test_rle_and_writers.py).  Byte / integer work: every comparison is exact."""
import numpy as np
import pytest
import torch

from samrs_amd import rle, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import samrs_amd
    sam = samrs_amd.sam_model_registry["vit_tiny"](max_prompts=4, max_points=1).to("cuda")
    return sam.engine


def _encode(eng, masks, cap=None, cursor0=0):
    n, h, w = masks.shape
    cap = cap or int(n * (h * w * 1.2 + 64)) + 64
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    cur = torch.full((1,), cursor0, dtype=torch.int64, device="cuda")
    tab = torch.zeros(n, 3, dtype=torch.int64, device="cuda")
    eng.rle_encode(torch.as_tensor(masks).cuda(), out, cur, tab)
    torch.cuda.synchronize()
    return out.cpu().numpy(), int(cur.item()), tab.cpu().numpy()


def _blobs(rng, h, w, p):
    m = rng.random((h, w)) < p
    if h > 8 and w > 8:
        m[h // 4: h // 2, w // 5: w // 2] = True
        m[h // 3, :] = False
    return m


@pytest.mark.parametrize("h,w", [(1024, 1024), (600, 800), (517, 803), (37, 91), (1, 1), (33, 1), (1, 70), (32, 64), (2048, 1100)])
def test_rle_strings_are_bit_exact(eng, h, w):
    rng = np.random.default_rng(h * 7 + w)
    masks = [np.zeros((h, w), bool), np.ones((h, w), bool), _blobs(rng, h, w, 0.0), _blobs(rng, h, w, 0.5), _blobs(rng, h, w, 0.02)]
    m = np.zeros((h, w), bool); m[0, 0] = True; masks.append(m)
    m = np.ones((h, w), bool); m[-1, -1] = False; masks.append(m)
    m = (np.add.outer(np.arange(h), np.arange(w)) & 1).astype(bool); masks.append(m)      # checkerboard: one count per pixel
    masks = np.stack(masks)
    out, cur, tab = _encode(eng, masks.astype(np.uint8))
    prev_end = 0
    for j in range(len(masks)):
        off, n, nc = (int(v) for v in tab[j])
        want = rle.encode(masks[j])
        assert n >= 0 and off % 16 == 0 and off >= prev_end
        got = out[off:off + n].tobytes().decode("ascii")
        assert got == want["counts"], f"mask {j} ({h}x{w}): {n} bytes vs {len(want['counts'])}"
        assert nc == len(rle.mask_to_counts(masks[j]))
        prev_end = off + n
    assert cur >= prev_end and cur % 16 == 0


def test_rle_nonzero_bytes_count_as_set_and_calls_append(eng):
    rng = np.random.default_rng(3)
    a = (rng.random((3, 200, 256)) < 0.3)
    vals = rng.integers(1, 256, size=a.shape).astype(np.uint8) * a        # any non-zero byte is "set"
    out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    cur = torch.zeros(1, dtype=torch.int64, device="cuda")
    t1 = torch.zeros(3, 3, dtype=torch.int64, device="cuda")
    t2 = torch.zeros(40, 3, dtype=torch.int64, device="cuda")
    eng.rle_encode(torch.as_tensor(vals).cuda(), out, cur, t1)
    b = rng.random((40, 64, 96)) < 0.5                                      # more masks than one internal pass (32)
    eng.rle_encode(torch.as_tensor(b).cuda(), out, cur, t2)
    torch.cuda.synchronize()
    o, t1, t2 = out.cpu().numpy(), t1.cpu().numpy(), t2.cpu().numpy()
    for j in range(3):
        assert o[t1[j, 0]:t1[j, 0] + t1[j, 1]].tobytes().decode() == rle.encode(a[j])["counts"]
    assert t2[0, 0] >= t1[2, 0] + t1[2, 1]
    for j in range(40):
        assert o[t2[j, 0]:t2[j, 0] + t2[j, 1]].tobytes().decode() == rle.encode(b[j])["counts"]
    assert int(cur.item()) >= t2[39, 0] + t2[39, 1]


def test_rle_overflow_is_reported_not_truncated(eng):
    rng = np.random.default_rng(5)
    m = rng.random((2, 256, 256)) < 0.5
    need = [len(rle.encode(x)["counts"]) for x in m]
    out, cur, tab = _encode(eng, m.astype(np.uint8), cap=((need[0] + 15) // 16) * 16 + 16)
    assert tab[0, 1] == need[0] and out[tab[0, 0]:tab[0, 0] + need[0]].tobytes().decode() == rle.encode(m[0])["counts"]
    assert tab[1, 1] == -need[1] - 1                                       # did not fit: the needed size, negated


def test_select_best_matches_torch(eng):
    g = torch.Generator().manual_seed(9)
    for h, w in [(1024, 1024), (333, 517)]:
        m = (torch.rand(7, 3, h, w, generator=g) < 0.4).cuda()
        q = torch.rand(7, 3, generator=g).cuda()
        q[2] = 0.5                                                          # ties: the first maximum wins, like torch.argmax
        best, qual, areas = eng.select_best(m, q)
        torch.cuda.synchronize()
        idx = q.argmax(1)
        rows = torch.arange(7, device="cuda")
        assert torch.equal(best.view(torch.bool), m[rows, idx]) and torch.equal(qual, q[rows, idx])
        assert torch.equal(areas, m[rows, idx].flatten(1).sum(1))


def test_device_rle_known_answers(eng, golden_dir):
    """samrs_rle_encode against the cocoapi known-answer vectors (tests/golden/coco_rle_known_answers.json: the published example
    and hand-derived cases, see tests/test_rle_and_writers.py::test_rle_string_known_answers) -- the device coder pinned to the
    library's published behaviour, not only to this repo's host restatement."""
    import json
    import os
    vec = json.load(open(os.path.join(golden_dir, "coco_rle_known_answers.json")))["vectors"]
    for v in vec:
        mask = rle.decode({"size": v["size"], "counts": v["counts"]})
        out, cur, tab = _encode(eng, mask[None].astype(np.uint8))
        off, n, nc = (int(x) for x in tab[0])
        assert out[off:off + n].tobytes().decode("ascii") == v["string"], v["name"]
        assert nc == len(v["counts"])
