"""GPU: the production loop (driver.TilePipeline -- batched encoder, three HIP streams, two slot sets) against the
one-image-at-a-time loop that restates the reference driver (driver.SemanticGenerator), plus the boundary features
that came with it: ragged encoder batches, prompt batches beyond max_prompts, prompt-shape validation.
Integer / index work and identical kernels on both sides: every comparison is bit-exact."""
import os

import numpy as np
import pytest
import torch

from samrs_amd import synth

pytestmark = pytest.mark.gpu


def _sam(name="vit_tiny", **kw):
    import samrs_amd
    return samrs_amd.sam_model_registry[name](**kw).to("cuda")


def _stream_items(driver, sizes, counts):
    items = []
    for i, ((h, w), n) in enumerate(zip(sizes, counts)):
        img = synth.make_image(20 + i, h, w)
        boxes, labels = synth.make_boxes(20 + i, n, h, w)
        items.append(driver.WorkItem(f"img{i}", img, boxes, labels))
    return items


@pytest.mark.parametrize("batch", [2, 3])
def test_pipeline_equals_serial_loop(batch):
    import samrs_amd
    from samrs_amd import driver
    sam = _sam(max_images=2 * batch, max_prompts=20, precision="f16")
    sizes = [(1024, 1024), (1024, 1024), (600, 800), (1024, 1024), (1024, 1024), (1024, 1024), (1024, 1024)]
    counts = [3, 23, 5, 41, 1, 20, 7]
    items = _stream_items(driver, sizes, counts)
    # serial: the reference's loop shape, one stream
    gen = driver.SemanticGenerator(samrs_amd.SamPredictor(sam), 18, box_batch=20)
    serial = []
    for it in items:
        r = gen.process_image(it.image, it.boxes, it.labels, keep_masks=True)
        serial.append((r.seg_mask.cpu().numpy(), r.areas.cpu().numpy(), r.masks.cpu().numpy()))
    # pipelined
    pipe = driver.TilePipeline(sam, 18, batch=batch, box_batch=20, keep_masks=True, max_boxes=64)
    got = {}

    def sink(results, release):
        for r in results:
            got[r.key] = (r.seg_mask.copy(), r.areas.copy(), r.masks.copy())
        release()

    n = pipe.run(driver.batched(items, batch), sink)
    assert n == len(items) and len(got) == len(items)
    for i, it in enumerate(items):
        seg, areas, masks = got[it.key]
        assert seg.shape == sizes[i]
        assert np.array_equal(seg, serial[i][0]), f"{it.key}: class map differs from the serial loop"
        assert np.array_equal(areas, serial[i][1])
        assert np.array_equal(masks.astype(bool), serial[i][2].astype(bool))
    assert torch.equal(pipe.class_pixels, gen.class_pixels) and torch.equal(pipe.class_instances, gen.class_instances)
    # a second run through the same pipeline object (ring buffers, events and slot sets are reused)
    got.clear()
    pipe.run(driver.batched(items[:3], batch), sink)
    for i in range(3):
        assert np.array_equal(got[items[i].key][0], serial[i][0])


def test_pipeline_output_does_not_depend_on_the_box_chunking():
    """The reference decodes 20 boxes per call (main_sam_hbox_semantic.py:91); `samrs_amd.generate` defaults to 64 because it is
    faster and the result must not depend on it: class maps, areas and masks of the same stream decoded in chunks of 20, 7 and
    64 boxes are bit-identical (no cross-prompt arithmetic, K-only split of the token GEMMs)."""
    from samrs_amd import driver
    sizes = [(1024, 1024), (600, 800), (1024, 1024), (1024, 1024)]
    counts = [41, 5, 64, 23]
    items = _stream_items(driver, sizes, counts)
    outs = []
    for bb in (20, 7, 64):
        sam = _sam(max_images=4, max_prompts=bb, precision="f16")
        pipe = driver.TilePipeline(sam, 18, batch=2, box_batch=bb, keep_masks=True, max_boxes=64)
        got = {}

        def sink(results, release):
            for r in results:
                got[r.key] = (r.seg_mask.copy(), r.areas.copy(), r.masks.copy())
            release()

        assert pipe.run(driver.batched(items, 2), sink) == len(items)
        outs.append((got, pipe.class_pixels.clone(), pipe.class_instances.clone()))
        del pipe, sam
    for got, cp, ci in outs[1:]:
        for it in items:
            for a, b in zip(got[it.key], outs[0][0][it.key]):
                assert np.array_equal(a, b), f"{it.key}: output depends on the box chunking"
        assert torch.equal(cp, outs[0][1]) and torch.equal(ci, outs[0][2])


def test_pipeline_device_resident_and_pinned_inputs():
    from samrs_amd import driver
    sam = _sam(max_images=4, max_prompts=8)
    tiles = torch.stack([torch.from_numpy(synth.make_noise_image(70 + i)) for i in range(4)])
    anns = [synth.make_boxes(70 + i, 6) for i in range(4)]
    outs = []
    for kind in ("numpy", "pinned", "device"):
        src = {"numpy": [t.numpy() for t in tiles], "pinned": list(tiles.pin_memory()), "device": list(tiles.cuda())}[kind]
        pipe = driver.TilePipeline(sam, 18, batch=2, box_batch=8, max_boxes=8, device_inputs=(kind == "device"))
        res = {}

        def sink(results, release):
            for r in results:
                res[r.key] = (r.seg_mask.copy(), r.areas.copy())
            release()

        pipe.run(driver.batched([driver.WorkItem(i, src[i], anns[i][0], anns[i][1]) for i in range(4)], 2), sink)
        outs.append(res)
    for i in range(4):
        for o in outs[1:]:
            assert np.array_equal(o[i][0], outs[0][i][0]) and np.array_equal(o[i][1], outs[0][i][1])


def test_ragged_encoder_batch_equals_single_tiles():
    import samrs_amd
    sam = _sam(max_images=4)
    eng = sam.engine
    tr = samrs_amd.ResizeLongestSide(1024)
    tiles = []
    for i, (h, w) in enumerate([(1024, 1024), (600, 800), (1024, 683), (1024, 1024)]):
        t = torch.as_tensor(synth.make_image(30 + i, h, w)).cuda()
        tiles.append(tr.apply_image_device(t).contiguous())
    assert [tuple(t.shape[:2]) for t in tiles] == [(1024, 1024), (768, 1024), (1024, 683), (1024, 1024)]
    eng.set_images_ragged(tiles, 0)
    batch = [eng.get_embedding(i).clone() for i in range(4)]
    for i, t in enumerate(tiles):
        eng.set_images(t[None].contiguous(), 0)
        assert torch.equal(eng.get_embedding(0), batch[i]), f"tile {i} ({tuple(t.shape)}): ragged batch differs from single encode"
    with pytest.raises(AssertionError):
        eng.set_images_ragged([tiles[0][:512, :512].contiguous()], 0)        # long side != 1024


def test_predict_batches_beyond_max_prompts():
    """The reference takes any B (its instance drivers pass every object of an image at once); the engine decodes
    max_prompts per pass and chunks inside samrs_predict -- bit-identical to manual chunking."""
    import samrs_amd
    sam = _sam(max_prompts=8, max_points=2)
    pred = samrs_amd.SamPredictor(sam)
    img = synth.make_image(3)
    pred.set_image(img)
    boxes, _ = synth.make_boxes(3, 19)
    tb = pred.transform.apply_boxes_torch(torch.from_numpy(boxes).cuda(), img.shape[:2])
    m, q, l = pred.predict_torch(None, None, tb, None, multimask_output=True)
    assert m.shape == (19, 3, 1024, 1024) and q.shape == (19, 3) and l.shape == (19, 3, 256, 256)
    parts = [pred.predict_torch(None, None, tb[s:e], None, multimask_output=True) for s, e in [(0, 8), (8, 16), (16, 19)]]
    assert torch.equal(m, torch.cat([p[0] for p in parts])) and torch.equal(q, torch.cat([p[1] for p in parts]))
    assert torch.equal(l, torch.cat([p[2] for p in parts]))
    # every prompt kind at once: points + boxes + mask prompts, logits out
    g = torch.Generator().manual_seed(5)
    pc = (torch.rand(19, 2, 2, generator=g) * 1000).cuda()
    pl = torch.randint(0, 2, (19, 2), generator=g).cuda()
    mi = torch.where(torch.rand(19, 1, 256, 256, generator=g) > 0.5, 1000.0, -1000.0).cuda()
    m2, q2, l2 = pred.predict_torch(pc, pl, tb, mi, multimask_output=False, return_logits=True)
    p2 = [pred.predict_torch(pc[s:e], pl[s:e], tb[s:e], mi[s:e], multimask_output=False, return_logits=True)
          for s, e in [(0, 8), (8, 16), (16, 19)]]
    assert m2.dtype == torch.float32 and torch.equal(m2, torch.cat([p[0] for p in p2])) and torch.equal(l2, torch.cat([p[2] for p in p2]))


def test_prompt_shape_validation():
    import samrs_amd
    sam = _sam(max_prompts=8, max_points=2)
    pred = samrs_amd.SamPredictor(sam)
    pred.set_image(synth.make_image(3))
    b = torch.tensor([[10.0, 10.0, 200.0, 300.0]] * 3).cuda()
    with pytest.raises(ValueError, match="mask_input must be"):
        pred.predict_torch(None, None, b, torch.zeros(3, 1, 1024, 1024).cuda())
    with pytest.raises(ValueError, match="mask_input has batch"):
        pred.predict_torch(None, None, b, torch.zeros(2, 1, 256, 256).cuda())
    with pytest.raises(ValueError, match="boxes has 3 rows"):
        pred.predict_torch(torch.zeros(2, 1, 2).cuda(), torch.ones(2, 1).cuda(), b, None)
    with pytest.raises(ValueError, match="point_labels must be"):
        pred.predict_torch(torch.zeros(3, 2, 2).cuda(), torch.ones(3, 1).cuda(), None, None)
    with pytest.raises(ValueError, match="max_points"):
        pred.predict_torch(torch.zeros(3, 3, 2).cuda(), torch.ones(3, 3).cuda(), None, None)
    with pytest.raises(AssertionError):
        pred.set_torch_image(torch.zeros(2, 3, 1024, 1024), (1024, 1024))        # one image per predictor
    with pytest.raises(AssertionError):
        pred.set_torch_image(torch.full((1, 3, 1024, 1024), 0.5), (1024, 1024))  # non-integer pixel values


@pytest.mark.parametrize("prompt,multimask", [("box", True), ("rbox_mask", True), ("point", False), ("point", True)])
def test_instance_pipeline_matches_prompter(prompt, multimask):
    """BASELINE.json configs[3] recipe (rbox -> enclosing hbox / rbox -> mask prompt, multimask_output=True, best of 3) and the
    point recipe of main_sam_hbox_mask_instance.py:160-165 (one foreground point per object, multimask_output=False there)
    through the three-stream pipeline == the one-image-at-a-time InstancePrompter; the best-of-3 selection, quality and area
    come from samrs_select_best on the device, the per-instance RLE from samrs_rle_encode."""
    import samrs_amd
    from samrs_amd import driver, rle
    sam = _sam(max_images=4, max_prompts=6, max_points=1)
    prm = driver.InstancePrompter(samrs_amd.SamPredictor(sam))
    items, ref = [], []
    for i in range(3):
        img = synth.make_image(40 + i)
        polys, labels = synth.make_rboxes(40 + i, 9)
        pts = polys.mean(1).astype(np.float32)                              # object centres
        items.append(driver.WorkItem(i, img, pts if prompt == "point" else polys, labels))
        if prompt == "box":
            m, q = prm.predict(img, "box", hboxes=synth.enclosing_hboxes(polys), multimask_output=multimask)
        elif prompt == "rbox_mask":
            m, q = prm.predict(img, "rbox_mask", rboxes=polys, multimask_output=multimask)
        else:
            m, q = prm.predict(img, "point", points=pts, multimask_output=multimask)
        ref.append((m.cpu().numpy(), q.cpu().numpy()))
    pipe = driver.InstancePipeline(sam, 37, prompt=prompt, multimask=multimask, batch=2, box_batch=6, max_boxes=16, keep_masks=True,
                                   rle=True, rle_buffer_mb=16)
    got = {}

    def sink(results, release):
        for r in results:
            assert r.seg_mask is None                                        # instance pipelines paint no class map
            got[r.key] = (r.masks.copy(), r.quality.copy(), r.areas.copy(), [r.rle(j) for j in range(len(r.labels))])
        release()

    pipe.run(driver.batched(items, 2), sink)
    for i in range(3):
        assert np.array_equal(got[i][0].astype(bool), ref[i][0]) and np.array_equal(got[i][1], ref[i][1])
        assert np.array_equal(got[i][2], ref[i][0].reshape(9, -1).sum(1))
        for j in range(9):
            assert got[i][3][j] == rle.encode(ref[i][0][j]), f"tile {i} object {j}: device RLE differs from the host restatement"


def test_pipeline_rle_mode_equals_host_rle():
    """rle=True: the per-instance COCO RLE of main_sam_hbox_semantic.py:201-202 comes from the device (samrs_rle_encode) while
    the masks stay in HBM; every string must equal samrs_amd.rle.encode of the mask the keep_masks mode hands over, for native,
    DIOR-shaped and ragged tiles, box counts beyond one chunk, and on a second run through the same pipeline."""
    from samrs_amd import driver, rle
    sam = _sam(max_images=4, max_prompts=20, precision="f16")
    sizes = [(1024, 1024), (600, 800), (1024, 1024), (517, 803), (1024, 1024)]
    counts = [3, 5, 41, 2, 20]
    items = _stream_items(driver, sizes, counts)
    pipe = driver.TilePipeline(sam, 18, batch=2, box_batch=20, keep_masks=True, max_boxes=64, rle=True, rle_buffer_mb=64)
    for rnd in range(2):
        got = {}

        def sink(results, release):
            for r in results:
                got[r.key] = (r.masks.copy(), [r.rle(j) for j in range(len(r.labels))], r.areas.copy(), r.rle_table.copy())
            release()

        assert pipe.run(driver.batched(items, 2), sink) == len(items)
        for i, it in enumerate(items):
            masks, rles, areas, tab = got[it.key]
            assert len(rles) == counts[i] and (tab[:, 0] % 16 == 0).all()
            for j in range(counts[i]):
                want = rle.encode(masks[j].astype(bool))
                assert rles[j] == want, f"{it.key} box {j} (round {rnd}): device RLE differs"
                assert rles[j]["size"] == list(sizes[i]) and int(tab[j, 2]) == len(rle.mask_to_counts(masks[j].astype(bool)))
                assert int(rle.decode(rles[j]).sum()) == int(areas[j])
    # a buffer that cannot hold the batch's strings fails loudly, it does not truncate
    small = driver.TilePipeline(sam, 18, batch=2, box_batch=20, max_boxes=64, rle=True, rle_buffer_mb=1)
    small.rle_dev = [t[:4096] for t in small.rle_dev]
    with pytest.raises(RuntimeError, match="RLE buffer too small"):
        small.run(driver.batched(items[:2], 2), lambda res, rel: rel())


def test_precision_follows_the_output_contract():
    """ViT-H engines default to the multimask-safe operand split (79: IoU >= 0.999 on the C4 fixtures); a single-mask pipeline
    runs ITS OWN calls in the 1x-rate mode (15: IoU >= 0.9995 on the C2 fixtures), a multimask pipeline in the model's default; an
    explicit choice (builder options / precision=) is never overridden.  The engine's option is left alone (round 3: whoever
    built the last pipeline decided everybody's precision); the mode an image was encoded in is recorded with its slot.  Smaller
    models have one mode (15)."""
    import samrs_amd
    from samrs_amd import driver
    sam = samrs_amd.sam_model_registry["vit_h"](precision="f16", max_images=2, max_prompts=8, max_points=1).to("cuda")
    eng = sam.engine
    assert sam.default_split == eng.get_option("split") == 79 and eng.get_option("split_depth") == 0
    assert eng.get_option("grade_multimask") == 16 | 64 and eng.get_option("allow_reduced") == 0
    p = driver.TilePipeline(sam, 18, batch=1, box_batch=8, max_boxes=8)
    assert p.split_mode == 15 and not p.allow_reduced
    p = driver.InstancePipeline(sam, 18, prompt="box", multimask=True, batch=1, box_batch=8, max_boxes=8)
    assert p.split_mode == 79
    p = driver.InstancePipeline(sam, 18, prompt="point", multimask=False, batch=1, box_batch=8, max_boxes=8)
    assert p.split_mode == 15
    p = driver.TilePipeline(sam, 18, batch=1, box_batch=8, max_boxes=8, precision="engine")
    assert p.split_mode is None
    p = driver.InstancePipeline(sam, 18, prompt="box", multimask=True, batch=1, box_batch=8, max_boxes=8, precision=15)
    assert p.split_mode == 15 and p.allow_reduced                  # an explicit reduced mode for a multimask pipeline = consent
    assert eng.get_option("split") == 79                           # building pipelines changed nothing on the engine
    # a single-mask pipeline's run leaves its slots marked with ITS mode and the engine's option untouched
    p = driver.TilePipeline(sam, 18, batch=1, box_batch=8, max_boxes=8)
    bx, lb = synth.make_boxes(0, 4)
    p.run([[driver.WorkItem("a", synth.make_image(0), bx, lb)]], lambda res, rel: rel())
    assert eng.get_option("split") == 79
    assert eng.get_slot_info(0) == {"is_set": 1, "split": 15, "split_depth": 0}
    eng.close()
    sam = samrs_amd.sam_model_registry["vit_h"](precision="f16", max_images=2, max_prompts=8, max_points=1, options={"split": 31}).to("cuda")
    p = driver.TilePipeline(sam, 18, batch=1, box_batch=8, max_boxes=8)
    assert p.split_mode is None and sam.engine.get_option("split") == 31 and sam.engine.get_option("allow_reduced") == 1
    sam.engine.close()
    tiny = samrs_amd.sam_model_registry["vit_tiny"](max_images=2, max_prompts=8).to("cuda")
    assert tiny.default_split == 15 and tiny.engine.get_option("grade_multimask") == 0
    p = driver.InstancePipeline(tiny, 18, prompt="box", multimask=True, batch=1, box_batch=8, max_boxes=8)
    assert p.split_mode == 15
    tiny.engine.close()


def test_multimask_after_a_single_mask_pipeline_keeps_its_precision(golden_dir):
    """a TilePipeline (single mask, 1x-rate mode) followed by the drop-in
    ``SamPredictor.predict_torch(multimask_output=True)`` (the reference's default, predictor.py:92-100) on the SAME model still
    meets the C4 floor -- the predictor encodes in the engine's default mode, which no pipeline changes any more -- and a
    multimask predict on an embedding the pipeline encoded is refused (PrecisionError) unless the caller allows the reduced
    mode; the refusal costs nothing on the single-mask path."""
    import samrs_amd
    from samrs_amd import driver
    from samrs_amd.engine import PrecisionError
    from oracle.make_golden import extended_inputs
    g = np.load(os.path.join(golden_dir, "vit_h_c2c4.npz"))
    cfg = synth.CONFIGS["vit_h"]
    sd = synth.make_state_dict(cfg, 0, logit_scale=float(g["logit_scale"]))
    sam = samrs_amd.sam_model_registry["vit_h"](state_dict=sd, precision="f16", max_images=2, max_prompts=32, max_points=1).to("cuda")
    eng = sam.engine
    inp = extended_inputs(0)
    img = synth.make_image(inp["image_index"])
    hw = img.shape[:2]
    pipe = driver.TilePipeline(sam, 18, batch=1, box_batch=20, max_boxes=32)
    pipe.run([[driver.WorkItem("t", img, inp["boxes"], inp["labels"])]], lambda res, rel: rel())
    assert eng.get_slot_info(0)["split"] == 15
    tb_dev = samrs_amd.ResizeLongestSide(1024).apply_boxes_torch(torch.from_numpy(inp["hboxes"]).cuda(), hw)
    with pytest.raises(PrecisionError, match="multimask"):
        eng.predict(0, tb_dev, None, None, None, True, False, hw, hw)          # the pipeline's slot, multimask: refused
    m1, _, _ = eng.predict(0, tb_dev, None, None, None, False, False, hw, hw)  # single mask on it: fine
    with eng.options(allow_reduced=1):
        m15, _, _ = eng.predict(0, tb_dev, None, None, None, True, False, hw, hw)
    pred = samrs_amd.SamPredictor(sam)
    pred.set_image(img)
    assert eng.get_slot_info(pred.slot) == {"is_set": 1, "split": 79, "split_depth": 24}
    m, q, l = pred.predict_torch(None, None, tb_dev, None, multimask_output=True)
    gm = torch.from_numpy(np.unpackbits(g["c4box_masks"], axis=-1).reshape(-1, 3, *hw).astype(bool))

    def ious(a):
        a = a.cpu().flatten(0, 1).flatten(1)
        b = gm.flatten(0, 1).flatten(1)
        return ((a & b).sum(1).double() / (a | b).sum(1).double().clamp(min=1))
    i79, i15 = ious(m), ious(m15)
    print(f"C4 hbox after a single-mask pipeline: predictor (slot split 79) IoU min {i79.min():.5f}; the pipeline's slot with allow_reduced "
          f"(split 15) {i15.min():.5f}")
    assert i79.min() >= 0.999
    assert i15.min() >= 0.998 and m1.shape[1] == 1
    eng.close()
