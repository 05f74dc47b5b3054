"""CPU: libsamrs_io.so (include/samrs_io.h) against PIL -- the library the reference itself reads and writes its tiles with
(Generate Dataset/main_sam_hbox_semantic.py:114,212-215).  Files written natively must decode in PIL to the same pixels;
files written by PIL in every 8-bit mode must decode natively to what ``Image.open(..).convert("RGB")`` gives."""
import ctypes
import os
import re
import threading

import numpy as np
import pytest
from PIL import Image

from samrs_amd import tile_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _class_map(h, w, seed=0):
    rng = np.random.default_rng(seed)
    seg = np.full((h, w), 255, np.uint8)
    for _ in range(12):
        y0, x0 = rng.integers(0, h), rng.integers(0, w)
        seg[y0:y0 + rng.integers(1, h // 2 + 2), x0:x0 + rng.integers(1, w // 2 + 2)] = rng.integers(0, 18)
    return seg


def test_library_exports_every_declared_symbol():
    lib = tile_io.load_library()
    header = open(os.path.join(ROOT, "include", "samrs_io.h")).read()
    names = set(re.findall(r"\b(samrs_io_\w+)\s*\(", header))
    assert len(names) == 8
    for n in names:
        assert hasattr(lib, n), n
    assert lib.samrs_io_abi_version() == tile_io.ABI_VERSION == int(re.search(r"SAMRS_IO_ABI_VERSION (\d+)", header).group(1))


@pytest.mark.parametrize("hw", [(1, 1), (1, 7), (7, 1), (16, 20), (64, 80), (257, 300), (1024, 1024)])
@pytest.mark.parametrize("level", [1, 6, tile_io.LEVEL_RUNS, tile_io.LEVEL_LABELS])
def test_written_files_decode_in_pil(tmp_path, hw, level):
    seg = _class_map(*hw, seed=hw[0])
    pal = np.random.default_rng(1).integers(0, 256, (18, 3), dtype=np.uint8)
    lut = tile_io.class_lut(pal)
    g, c, r = str(tmp_path / "g.png"), str(tmp_path / "c.png"), str(tmp_path / "r.png")
    tile_io.write_gray(g, seg, level)
    tile_io.write_lut_rgb(c, seg, lut, level)
    noise = np.random.default_rng(2).integers(0, 256, (*hw, 3), dtype=np.uint8)
    tile_io.write_rgb(r, noise, level)
    gi = Image.open(g)
    assert gi.mode == "L" and np.array_equal(np.array(gi), seg)
    ci = Image.open(c)
    assert ci.mode == "RGB" and np.array_equal(np.array(ci), lut[seg])
    assert np.array_equal(np.array(Image.open(r)), noise)
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]
    # and back through the native decoder
    assert np.array_equal(tile_io.read_rgb(c), lut[seg])
    assert np.array_equal(tile_io.read_rgb(g), np.repeat(seg[:, :, None], 3, axis=2))
    assert np.array_equal(tile_io.read_rgb(r), noise)
    assert tile_io.png_size(c) == hw


def test_strided_sources(tmp_path):
    big = _class_map(64, 96, 3)
    view = big[8:40, 16:80]                                     # row stride 96, not contiguous
    tile_io.write_gray(str(tmp_path / "v.png"), view)
    assert np.array_equal(np.array(Image.open(tmp_path / "v.png")), view)
    tile_io.write_gray(str(tmp_path / "t.png"), big.T)          # column stride != 1: copied
    assert np.array_equal(np.array(Image.open(tmp_path / "t.png")), big.T)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P"])
@pytest.mark.parametrize("hw", [(5, 3), (128, 96), (600, 800)])
def test_reads_what_pil_writes(tmp_path, mode, hw):
    rng = np.random.default_rng(hw[0] + len(mode))
    # smooth + noise, so that PIL's encoder picks all five row filters
    yy, xx = np.mgrid[:hw[0], :hw[1]]
    base = ((yy * 3 + xx * 2) % 256).astype(np.uint8)
    ch = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2, "P": 1}[mode]
    arr = np.stack([(base + rng.integers(0, 24, hw)).astype(np.uint8) * (1 if c % 2 == 0 else 3) for c in range(ch)], axis=2)
    if mode == "P":
        img = Image.fromarray(arr[:, :, 0], mode="P")
        img.putpalette(rng.integers(0, 256, 768, dtype=np.uint8).tobytes())
    else:
        img = Image.fromarray(arr[:, :, 0] if ch == 1 else arr, mode=mode)
    path = str(tmp_path / "x.png")
    img.save(path)
    want = np.array(Image.open(path).convert("RGB"))
    got = tile_io.read_rgb(path)
    assert got.shape == want.shape and np.array_equal(got, want)
    # into a caller-provided (oversized) buffer: the result is a view of it
    out = np.zeros(want.size + 100, np.uint8)
    got2 = tile_io.read_rgb(path, out=out)
    assert np.shares_memory(got2, out) and np.array_equal(got2, want)


def test_variants_outside_the_subset_go_through_pil(tmp_path):
    rng = np.random.default_rng(0)
    a16 = rng.integers(0, 65536, (40, 30), dtype=np.uint16)
    p16 = str(tmp_path / "a16.png")
    Image.fromarray(a16).save(p16)
    lib = tile_io.load_library()
    buf = np.zeros(40 * 30 * 3, np.uint8)
    h, w = ctypes.c_int(), ctypes.c_int()
    assert lib.samrs_io_png_read_rgb(os.fsencode(p16), buf.ctypes.data, buf.size, ctypes.byref(h), ctypes.byref(w)) == tile_io.UNSUPPORTED
    assert np.array_equal(tile_io.read_rgb(p16), np.array(Image.open(p16).convert("RGB")))
    # a 1-bit PNG and a JPEG
    bw = Image.fromarray(rng.integers(0, 2, (33, 47)).astype(bool))
    pbw = str(tmp_path / "bw.png")
    bw.save(pbw)
    assert np.array_equal(tile_io.read_rgb(pbw), np.array(Image.open(pbw).convert("RGB")))
    rgb = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    pj = str(tmp_path / "x.jpg")
    Image.fromarray(rgb).save(pj)
    assert np.array_equal(tile_io.read_rgb(pj), np.array(Image.open(pj).convert("RGB")))


def test_errors_are_codes_not_crashes(tmp_path):
    seg = _class_map(32, 32)
    path = str(tmp_path / "ok.png")
    tile_io.write_gray(path, seg)
    raw = open(path, "rb").read()
    with pytest.raises(FileNotFoundError):
        tile_io.read_rgb(str(tmp_path / "missing.png"))
    # truncated at every interesting boundary, and with flipped bytes (chunk CRC / zlib checks)
    for cut in (0, 7, 20, 33, 40, len(raw) - 13, len(raw) - 1):
        bad = str(tmp_path / f"cut{cut}.png")
        open(bad, "wb").write(raw[:cut])
        with pytest.raises(Exception):
            tile_io.read_rgb(bad)
    for pos in (17, 30, 45, len(raw) - 20):
        b = bytearray(raw)
        b[pos] ^= 0x55
        bad = str(tmp_path / f"flip{pos}.png")
        open(bad, "wb").write(bytes(b))
        with pytest.raises(Exception):
            tile_io.read_rgb(bad)
    with pytest.raises(tile_io.TileIOError) as e:
        tile_io.write_gray(str(tmp_path / "no_such_dir" / "x.png"), seg)
    assert e.value.code == tile_io.EOPEN
    with pytest.raises(ValueError):
        tile_io.write_gray(path, seg.astype(np.int32))
    with pytest.raises(ValueError):
        tile_io.read_rgb(path, out=np.zeros(10, np.uint8))
    lib = tile_io.load_library()
    buf = np.zeros(8, np.uint8)
    h, w = ctypes.c_int(), ctypes.c_int()
    assert lib.samrs_io_png_read_rgb(os.fsencode(path), buf.ctypes.data, buf.size, ctypes.byref(h), ctypes.byref(w)) == tile_io.ESIZE
    assert (h.value, w.value) == (32, 32)


def test_threads_run_concurrently(tmp_path):
    """The point of the native path: many tiles in flight, each call independent (no shared state in the library)."""
    segs = [_class_map(256, 256, s) for s in range(16)]
    lut = tile_io.class_lut(np.random.default_rng(5).integers(0, 256, (18, 3), dtype=np.uint8))
    errs = []

    def work(i):
        try:
            p = str(tmp_path / f"c{i}.png")
            for _ in range(3):
                tile_io.write_lut_rgb(p, segs[i], lut)
                if not np.array_equal(tile_io.read_rgb(p), lut[segs[i]]):
                    errs.append(i)
        except Exception as e:                                  # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(16)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs


@pytest.mark.parametrize("kind", ["noise", "blobs", "constant", "columns", "rows", "wide"])
def test_label_aware_encoder_is_a_plain_png(tmp_path, kind):
    """SAMRS_IO_LEVEL_LABELS (the generation CLI's default): the deflate tokens of gray/*.png and color/*.png are derived from the
    1-byte class map (runs equal to the row above / to the left neighbour, literal pixels otherwise; filter 0, dynamic Huffman).
    Whatever the map looks like -- noise-like (the synthetic weights' worst case: one deflate block of ~10^6 tokens and more),
    blobs, constant (one symbol: the degenerate Huffman tree), vertical / horizontal stripes (only "up" / only "left" matches),
    a row too wide for the "up" distance to fit deflate's 32 KiB window -- PIL and the native decoder must return exactly the
    class map / its palette expansion (main_sam_hbox_semantic.py:199,208-216: seg_color = MAPPING[seg_mask])."""
    rng = np.random.default_rng(7)
    h, w = 256, 384
    if kind == "noise":
        seg = rng.integers(0, 18, (1024, 1100)).astype(np.uint8)             # > 2^20 tokens: two deflate blocks
    elif kind == "blobs":
        seg = _class_map(h, w, seed=3)
    elif kind == "constant":
        seg = np.full((h, w), 255, np.uint8)
    elif kind == "columns":
        seg = np.repeat(rng.integers(0, 18, (1, w)).astype(np.uint8), h, axis=0)
    elif kind == "rows":
        seg = np.repeat(rng.integers(0, 18, (h, 1)).astype(np.uint8), w, axis=1)
    else:
        seg = np.repeat(rng.integers(0, 18, (3, 1200)).astype(np.uint8), 10, axis=1)      # 12000 px wide: 3 w + 1 > 32768
    lut = tile_io.class_lut(rng.integers(0, 256, (18, 3), dtype=np.uint8))
    g, c = str(tmp_path / "g.png"), str(tmp_path / "c.png")
    tile_io.write_gray(g, seg, tile_io.LEVEL_LABELS)
    tile_io.write_lut_rgb(c, seg, lut, tile_io.LEVEL_LABELS)
    gi, ci = Image.open(g), Image.open(c)
    assert gi.mode == "L" and ci.mode == "RGB"
    assert np.array_equal(np.array(gi), seg) and np.array_equal(np.array(ci), lut[seg])
    assert np.array_equal(tile_io.read_rgb(c), lut[seg])
    # a strided source (a column slice of a wider map) encodes the same image
    wide = np.concatenate([seg, seg[:, ::-1]], axis=1)
    tile_io.write_lut_rgb(c, wide[:, :seg.shape[1]], lut, tile_io.LEVEL_LABELS)
    assert np.array_equal(np.array(Image.open(c)), lut[seg])


@pytest.mark.parametrize("kind", ["noise", "blobs", "constant", "columns", "rows", "wide", "pairs", "fixture"])
def test_label_pair_writer_is_one_parse_two_plain_pngs(tmp_path, kind, golden_dir):
    """samrs_io_png_write_label_pair (what `generate` calls per tile): gray/*.png and color/*.png of one class map from ONE parse
    of its label runs.  The parse is tuned for the 3-byte stream; the gray stream turns runs shorter than deflate's 3-byte minimum
    into literals read back from the map -- "pairs" (every label twice in a row: nothing but 2-pixel runs) is the map that lives on
    that branch, "noise" crosses the 2^20-token block boundary with the walk position carried between blocks, "wide" has rows whose
    RGB "up" distance does not fit deflate's window (so neither stream may use it), "fixture" is the reference's own class map
    of the ViT-H C2 fixture.  Both files must decode (PIL and the native decoder) to exactly the map / its palette expansion
    (main_sam_hbox_semantic.py:199,212-215), and stay within 3 % of the size of the two single-stream calls."""
    rng = np.random.default_rng(11)
    h, w = 256, 384
    if kind == "noise":
        seg = rng.integers(0, 18, (1024, 1100)).astype(np.uint8)
    elif kind == "blobs":
        seg = _class_map(h, w, seed=5)
    elif kind == "constant":
        seg = np.full((h, w), 255, np.uint8)
    elif kind == "columns":
        seg = np.repeat(rng.integers(0, 18, (1, w)).astype(np.uint8), h, axis=0)
    elif kind == "rows":
        seg = np.repeat(rng.integers(0, 18, (h, 1)).astype(np.uint8), w, axis=1)
    elif kind == "wide":
        seg = np.repeat(rng.integers(0, 18, (3, 1200)).astype(np.uint8), 10, axis=1)
    elif kind == "pairs":
        seg = np.repeat(rng.integers(0, 200, (h, w // 2)).astype(np.uint8), 2, axis=1)
    else:
        seg = np.ascontiguousarray(np.load(os.path.join(golden_dir, "vit_h_c2c4.npz"))["c2_seg"])
    lut = tile_io.class_lut(rng.integers(0, 256, (200, 3), dtype=np.uint8))
    g, c = str(tmp_path / "g.png"), str(tmp_path / "c.png")
    tile_io.write_label_pair(g, c, seg, lut)
    gi, ci = Image.open(g), Image.open(c)
    assert gi.mode == "L" and ci.mode == "RGB"
    assert np.array_equal(np.array(gi), seg) and np.array_equal(np.array(ci), lut[seg])
    assert np.array_equal(tile_io.read_rgb(c), lut[seg]) and np.array_equal(tile_io.read_rgb(g)[..., 0], seg)
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]
    g1, c1 = str(tmp_path / "g1.png"), str(tmp_path / "c1.png")
    tile_io.write_gray(g1, seg, tile_io.LEVEL_LABELS)
    tile_io.write_lut_rgb(c1, seg, lut, tile_io.LEVEL_LABELS)
    assert open(c, "rb").read() == open(c1, "rb").read()                      # the colour stream IS the single call's
    if kind != "pairs":                                                       # (there the unit-1 parse finds 3-byte matches the shared one cannot)
        assert os.path.getsize(g) <= 1.03 * os.path.getsize(g1) + 64, (os.path.getsize(g), os.path.getsize(g1))
    # a strided source
    wide = np.concatenate([seg, seg[:, ::-1]], axis=1)
    tile_io.write_label_pair(g, c, wide[:, :seg.shape[1]], lut)
    assert np.array_equal(np.array(Image.open(g)), seg) and np.array_equal(np.array(Image.open(c)), lut[seg])
    with pytest.raises(ValueError):
        tile_io.write_label_pair(g, c, seg, lut[:100])


def test_label_pair_writer_on_small_and_odd_shapes(tmp_path):
    """One-pixel images, widths that are no multiple of the 8-label compare of the parse, maps made of 2-pixel runs (the gray
    stream's literal branch) and sparse maps: both files decode to the map / its palette expansion."""
    rng = np.random.default_rng(0)
    lut = tile_io.class_lut(rng.integers(0, 256, (255, 3), dtype=np.uint8))
    g, c = str(tmp_path / "g.png"), str(tmp_path / "c.png")
    shapes = [(1, 1), (1, 2), (2, 1), (1, 3), (3, 7), (5, 8), (17, 9), (2, 258), (2, 259), (3, 87)]
    shapes += [(int(rng.integers(1, 40)), int(rng.integers(1, 70))) for _ in range(60)]
    for it, (h, w) in enumerate(shapes):
        mode = it % 4
        if mode == 0:
            seg = rng.integers(0, int(rng.integers(1, 6)), (h, w))
        elif mode == 1:
            seg = np.repeat(np.repeat(rng.integers(0, 255, (h // 3 + 1, w // 4 + 1)), 3, 0), 4, 1)[:h, :w]
        elif mode == 2:
            seg = np.repeat(rng.integers(0, 255, (h, w // 2 + 1)), 2, 1)[:, :w]
        else:
            seg = (rng.random((h, w)) < 0.1) * int(rng.integers(1, 255))
        seg = np.ascontiguousarray(seg.astype(np.uint8))
        tile_io.write_label_pair(g, c, seg, lut)
        assert np.array_equal(np.array(Image.open(g)).reshape(h, w), seg), (h, w, mode)
        assert np.array_equal(np.array(Image.open(c)).reshape(h, w, 3), lut[seg]), (h, w, mode)
