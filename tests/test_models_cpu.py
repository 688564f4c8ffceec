"""CPU-only: host-side model quantisations of the drop-in module (constriction_amd.stream.model; the perfect quantisation
runs in the library's host code) -- no GPU, no oracle.
Properties are the reference's own test properties (src/stream/model/categorical/contiguous.rs:700-830,
src/stream/model/uniform.rs); see the module docstring for what pins which family."""
import math

import numpy as np
import pytest

from constriction_amd.stream import model as M


def cross_entropy(probs, cdf):
    w = np.diff(cdf.astype(np.int64)) / float(cdf[-1])
    p = np.asarray(probs, dtype=np.float64)
    p = p / p.sum()
    return float(-(p[p > 0] * np.log2(w[p > 0])).sum())


@pytest.mark.parametrize("seed", range(12))
def test_perfect_quantisation_properties(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 7, 50, 300]))
    probs = rng.dirichlet(np.ones(n) * rng.choice([0.05, 0.5, 5.0]))
    if seed % 3 == 0:
        probs[rng.integers(n)] = 0.0                  # zero-probability symbols still get weight 1
    for P in (12, 24):
        if n >= (1 << P):
            continue
        perfect = M.perfect_quantized_cdf(probs, P)
        fast = M.fast_quantized_cdf(probs, P)
        assert perfect[0] == 0 and int(perfect[-1]) == 1 << P
        assert (np.diff(perfect.astype(np.int64)) >= 1).all()
        # contiguous.rs:754-780: the perfect quantisation is never worse than the fast one
        assert cross_entropy(probs, perfect) <= cross_entropy(probs, fast) + 1e-12
        # local optimality: moving one unit between any two symbols does not lower the cross entropy
        w = np.diff(perfect.astype(np.int64))
        p = probs / probs.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            win = p * np.log1p(1.0 / w)
            loss = np.where(w > 1, -p * np.log1p(-1.0 / np.maximum(w, 2)), np.inf)
        assert win.max() <= loss.min() + 1e-15 or np.argmax(win) == np.argmin(loss)


def test_perfect_quantisation_small_cases():
    assert M.perfect_quantized_cdf(np.array([0.5, 0.5]), 24).tolist() == [0, 1 << 23, 1 << 24]
    assert M.perfect_quantized_cdf(np.array([1.0, 0.0]), 8).tolist() == [0, 255, 256]
    assert M.perfect_quantized_cdf(np.array([0.25, 0.5, 0.25], dtype=np.float32), 12).tolist() == [0, 1024, 3072, 4096]
    for bad in ([1.0], [], [0.5, -0.1], [float("nan"), 0.5], [float("inf"), 1.0], [0.0, 0.0]):
        with pytest.raises(ValueError):
            M.perfect_quantized_cdf(np.array(bad, dtype=np.float64), 24)


def test_uniform_table_is_uniform_rs():
    """uniform.rs:120-137: probability_per_bin = 2^24 / size (integer), the last symbol takes the rest"""
    for size in (2, 3, 10, 1000, 65536):
        cdf = M.Uniform(size).cdf.astype(np.int64)
        per = (1 << 24) // size
        assert cdf[0] == 0 and cdf[-1] == 1 << 24 and len(cdf) == size + 1
        assert (np.diff(cdf)[:-1] == per).all() and cdf[-1] - cdf[-2] == (1 << 24) - (size - 1) * per
    for bad in (0, 1, (1 << 24) + 1):
        with pytest.raises(ValueError):
            M.Uniform(bad)
    rows = M.Uniform().family_rows((np.array([2, 5, 3], dtype=np.int32),))
    assert rows.shape == (3, 6) and rows[0].tolist() == [0, 1 << 23, 1 << 24, 1 << 24, 1 << 24, 1 << 24]


def test_leaky_families_argument_checks():
    """constructors validate like the reference (model.rs:736-900 assert scale > 0; quantize.rs:292-294); the tables
    themselves are built on the GPU (tests/test_gpu_model_families.py compares every entry with the oracle)"""
    with pytest.raises(ValueError):
        M.QuantizedLaplace(-5, 5, 0.0, 0.0)
    with pytest.raises(ValueError):
        M.QuantizedCauchy(3, 3, 0.0, 1.0)
    with pytest.raises(ValueError):
        M.QuantizedLaplace(-5, 5, 1.0)                 # only one of the two parameters
    assert not M.QuantizedCauchy(-5, 5).is_concrete() and M.QuantizedLaplace(-5, 5, 0.0, 1.0).is_concrete()
    with pytest.raises(ValueError):
        M.Binomial(None, 0.5)


def test_categorical_flags_like_the_reference(capsys):
    with pytest.raises(ValueError):
        M.Categorical(np.array([0.5, 0.5]), lazy=True, perfect=True)
    M._warned.discard("categorical")
    m = M.Categorical(np.array([0.1, 0.6, 0.3]))               # legacy default: perfect, with a warning (printed once)
    M.Categorical(np.array([0.1, 0.6, 0.3]))
    out = capsys.readouterr().out
    assert out.count("WARNING: Neither argument `perfect` nor `lazy`") == 1 and m.perfect
    assert M.Categorical(np.array([0.1, 0.6, 0.3]), lazy=True).cdf.tolist() == M.Categorical(np.array([0.1, 0.6, 0.3]), perfect=False).cdf.tolist()
    assert M.Bernoulli(0.25, perfect=False).cdf.tolist() == M.fast_quantized_cdf(np.array([0.75, 0.25])).tolist()
    with pytest.raises(ValueError):
        M.Bernoulli(1.5, perfect=True)       # (the fast quantisation does not look at signs: categorical.rs:16-54)


def test_packed_batch_container_roundtrip(tmp_path):
    """the wire format of a packed batch (counterpart of `compressed.tofile` in the reference's docs): little-endian on disk,
    every stream's slice unchanged, malformed files rejected"""
    import numpy as np
    from constriction_amd import container
    rng = np.random.default_rng(4)
    lens = rng.integers(0, 9, 13)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    words = rng.integers(0, 2 ** 32, int(offsets[-1]) + 5, dtype=np.uint64).astype(np.uint32)   # (capacity may exceed the total)
    path = tmp_path / "batch.cst"
    container.save(path, words, offsets, (32, 64, 24))
    got_words, got_offsets, cfg = container.load(path)
    assert cfg == (32, 64, 24) and got_offsets.tolist() == offsets.tolist()
    assert got_words.tolist() == words[: int(offsets[-1])].tolist()
    raw = path.read_bytes()
    assert raw[:8] == b"CSTPACK1" and raw[40 + 8 * 14: 40 + 8 * 14 + 4] == int(words[0]).to_bytes(4, "little")
    import pytest
    (tmp_path / "short.cst").write_bytes(raw[:-3])
    with pytest.raises(ValueError):
        container.load(tmp_path / "short.cst")
    (tmp_path / "magic.cst").write_bytes(b"X" + raw[1:])
    with pytest.raises(ValueError):
        container.load(tmp_path / "magic.cst")
    with pytest.raises(ValueError):
        container.save(tmp_path / "bad.cst", words, offsets[::-1].copy(), (32, 64, 24))
    # a hostile header: huge sizes must be rejected from the file size alone (nothing of that size is read or allocated)
    import struct
    (tmp_path / "huge.cst").write_bytes(struct.pack("<8sQQIIII", b"CSTPACK1", 1 << 60, 1 << 61, 32, 64, 24, 0) + raw[40:])
    with pytest.raises(ValueError):
        container.load(tmp_path / "huge.cst")
    # offsets that decrease / do not start at 0 / a preset the coders do not have: what `save` refuses, `load` refuses too
    body = bytearray(raw)
    body[40 + 8 * 3: 40 + 8 * 4] = struct.pack("<Q", int(offsets[-1]))        # offsets[3] > offsets[4]
    (tmp_path / "order.cst").write_bytes(bytes(body))
    with pytest.raises(ValueError):
        container.load(tmp_path / "order.cst")
    body = bytearray(raw)
    body[40: 48] = struct.pack("<Q", 1)
    (tmp_path / "start.cst").write_bytes(bytes(body))
    with pytest.raises(ValueError):
        container.load(tmp_path / "start.cst")
    body = bytearray(raw)
    body[24: 28] = struct.pack("<I", 64)
    (tmp_path / "preset.cst").write_bytes(bytes(body))
    with pytest.raises(ValueError):
        container.load(tmp_path / "preset.cst")



def test_packed_batch_container_with_a_jump_table(tmp_path):
    """round 5: the batch's jump table (Pos / Seek side information) behind the words -- a reader that ignores it gets the plain batch,
    the old format (field 36..39 = 0) reads as a batch without one, truncated and hostile tables are rejected from the file size"""
    import struct
    from types import SimpleNamespace
    import numpy as np
    import pytest
    from constriction_amd import container
    rng = np.random.default_rng(5)
    n_streams, k = 11, 3
    lens = rng.integers(1, 9, n_streams)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    words = rng.integers(0, 2 ** 32, int(offsets[-1]), dtype=np.uint64).astype(np.uint32)
    jump = SimpleNamespace(interval=64, pos=rng.integers(0, lens[:, None] + 1, (n_streams, k)).astype(np.int32),      # (points lie within their streams)
                           state=rng.integers(0, 2 ** 63, (n_streams, k), dtype=np.int64))
    path = tmp_path / "jump.cst"
    container.save(path, words, offsets, (32, 64, 12), jump_points=jump)
    w, o, cfg = container.load(path)                                   # the table is skipped
    assert cfg == (32, 64, 12) and w.tolist() == words.tolist() and o.tolist() == offsets.tolist()
    w, o, cfg, got = container.load_with_jump_points(path)
    assert got[0] == 64 and got[1].dtype == np.uint32 and got[2].dtype == np.uint64
    assert got[1].tolist() == jump.pos.tolist() and got[2].tolist() == jump.state.view(np.uint64).tolist()
    plain = tmp_path / "plain.cst"
    container.save(plain, words, offsets, (32, 64, 12))
    assert container.load_with_jump_points(plain)[3] is None
    raw = path.read_bytes()
    assert len(raw) == len(plain.read_bytes()) + 8 + 12 * n_streams * k
    (tmp_path / "cut.cst").write_bytes(raw[:-8])
    with pytest.raises(ValueError):
        container.load(tmp_path / "cut.cst")
    body = bytearray(raw)
    body[36:40] = struct.pack("<I", 1 << 30)                           # a hostile n_chunks: the size check, nothing allocated
    (tmp_path / "huge.cst").write_bytes(bytes(body))
    with pytest.raises(ValueError):
        container.load_with_jump_points(tmp_path / "huge.cst")
    with pytest.raises(ValueError):
        container.save(tmp_path / "bad.cst", words, offsets, (32, 64, 12), jump_points=SimpleNamespace(interval=64, pos=jump.pos[:-1], state=jump.state[:-1]))
    # a jump point beyond its stream's words would make a packed decoder read the neighbour's: refused when the file is read
    beyond = jump.pos.copy()
    beyond[4, 1] = lens[4] + 1
    container.save(tmp_path / "beyond.cst", words, offsets, (32, 64, 12), jump_points=SimpleNamespace(interval=64, pos=beyond, state=jump.state))
    with pytest.raises(ValueError):
        container.load_with_jump_points(tmp_path / "beyond.cst")
