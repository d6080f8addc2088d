"""Bit-packed vectors (SURVEY 8 f2 "optional bit-packed output"; include/word2bits_corpus.h).

not gpu: the host side -- pack / unpack twins against a numpy restatement of the layout, the packed model file, and
its expansion back into the reference's output format, byte for byte the committed files of the unmodified reference
(tests/golden/*.vec).
gpu: the device-side producer (w2b_export_packed) against the host twin applied to w2b_export_quantized, the command
line's -packed flag, and the evaluator reading a packed file (same transcript as from the reference's format)."""
import os
import subprocess

import numpy as np
import pytest

import word2bits_amd as w2b
from word2bits_amd import _lib
from w2b_testlib import GOLDEN, ROOT, oracle, fptr, read_vectors

CORPUS = os.path.join(GOLDEN, "corpus_small.txt")
LEVELS = {1: np.float32(1) / np.float32(3), 2: None}


def quantized(rng, rows, dim, bitlevel):
    x = rng.standard_normal((rows, dim)).astype(np.float32)
    x[rng.random((rows, dim)) < 0.05] = 0.0                  # +0 quantizes to the positive level (ref :80)
    q = np.empty_like(x)
    oracle().w2bo_quantize_array(fptr(x), fptr(q), x.size, bitlevel)
    return q


def numpy_pack(q, bitlevel):
    """the layout of include/word2bits_corpus.h, restated: per block of 64 columns a word of sign bits and, at
    bitlevel 2, a word of magnitude bits (set = 0.75)"""
    rows, dim = q.shape
    nb = (dim + 63) // 64
    pad = np.zeros((rows, nb * 64), np.float32)
    pad[:, :dim] = q
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    sign = (np.signbit(pad) & (np.arange(nb * 64) < dim)).reshape(rows, nb, 64)
    planes = [(sign * weights).sum(-1, dtype=np.uint64)]
    if bitlevel == 2:
        mag = (np.abs(pad) > 0.5).reshape(rows, nb, 64)
        planes.append((mag * weights).sum(-1, dtype=np.uint64))
    return np.stack(planes, -1).reshape(rows, nb * bitlevel)


@pytest.mark.parametrize("bitlevel", [1, 2])
@pytest.mark.parametrize("dim", [1, 8, 63, 64, 65, 200, 400, 1000])
def test_pack_unpack_twins(bitlevel, dim):
    rng = np.random.default_rng(dim * 10 + bitlevel)
    q = quantized(rng, 37, dim, bitlevel)
    assert w2b.packed_words_per_row(dim, bitlevel) == (dim + 63) // 64 * bitlevel
    p = w2b.pack_quantized(q, bitlevel)
    assert p.dtype == np.uint64 and np.array_equal(p, numpy_pack(q, bitlevel))
    back = w2b.unpack_quantized(p, dim, bitlevel)
    assert np.array_equal(back.view(np.uint32), q.view(np.uint32))          # lossless: the exact bit patterns


def test_pack_rejects_what_is_not_quantized():
    L = w2b.lib()
    assert L.w2b_packed_words_per_row(100, 0) == -1 and L.w2b_packed_words_per_row(100, 4) == -1
    with pytest.raises(w2b.W2bError) as e:
        w2b.packed_words_per_row(100, 3)
    assert e.value.code == _lib.W2B_EUNSUPPORTED
    for bitlevel, bad in ((1, 0.25), (2, 1.0 / 3.0), (2, 0.0)):
        x = np.full((2, 5), bad, np.float32)
        with pytest.raises(w2b.W2bError) as e:
            w2b.pack_quantized(x, bitlevel)
        assert e.value.code == _lib.W2B_EINVAL


@pytest.mark.parametrize("name,bitlevel,min_count", [("b1_d8", 1, 2), ("b1_iter0", 1, 1), ("b2_d10_text", 2, 1)])
def test_packed_file_expands_to_the_reference_file(name, bitlevel, min_count, tmp_path):
    """the unmodified reference's output file -> packed file -> back: the same bytes, binary and text"""
    golden = os.path.join(GOLDEN, name + ".vec")
    binary = 0 if name.endswith("_text") else 1
    words, M = read_vectors(golden, binary)
    c = w2b.Corpus(CORPUS, min_count)
    assert c.words() == words
    pk = str(tmp_path / "m.w2bp")
    c.save_vectors_packed(pk, w2b.pack_quantized(M, bitlevel), M.shape[1], bitlevel)
    ref_bytes = os.path.getsize(golden) if binary else None
    out = str(tmp_path / "o.vec")
    w2b.unpack_vectors_file(pk, out, binary)
    assert open(out, "rb").read() == open(golden, "rb").read()
    # and into the other format: what the library's own writer produces from the same values
    other, want = str(tmp_path / "o2.vec"), str(tmp_path / "w2.vec")
    w2b.unpack_vectors_file(pk, other, 1 - binary)
    c.save_vectors(want, M, 1 - binary)
    assert open(other, "rb").read() == open(want, "rb").read()
    if ref_bytes:                                              # the point of the exercise
        names = sum(len(w) + 1 for w in words)
        assert os.path.getsize(pk) - names < (ref_bytes - names) / (32 / bitlevel) + 8 * bitlevel * len(words) + 64
    c.close()


def test_damaged_packed_files_are_io_errors(tmp_path):
    p, o = str(tmp_path / "x.w2bp"), str(tmp_path / "o.vec")
    for blob in (b"", b"W2BP1 3 8 1\na\nb\n", b"W2BP1 2 8 1\na\nb\n" + b"\0" * 15, b"W2BP1 2 8 5\na\nb\n" + b"\0" * 64,
                 b"12 8\nfoo "):
        open(p, "wb").write(blob)
        with pytest.raises(w2b.W2bError):
            w2b.unpack_vectors_file(p, o, 1)
    open(p, "wb").write(b"W2BP1 2 8 1\na\nb\n" + b"\0" * 16)            # complete: all signs positive
    w2b.unpack_vectors_file(p, o, 1)
    words, M = read_vectors(o, True)
    assert words == ["a", "b"] and np.all(M.view(np.uint32) == 0x3EAAAAAB)


def test_packed_file_reader_survives_damage(tmp_path):
    """random truncations, byte flips and header edits of a valid packed file: an error code or a file, never a crash
    or an allocation sized by a damaged header"""
    rng = np.random.default_rng(11)
    words, M = read_vectors(os.path.join(GOLDEN, "b1_d8.vec"), True)
    c = w2b.Corpus(CORPUS, 2)
    good, out = str(tmp_path / "g.w2bp"), str(tmp_path / "o.vec")
    c.save_vectors_packed(good, w2b.pack_quantized(M, 1), M.shape[1], 1)
    c.close()
    blob = open(good, "rb").read()
    bad = str(tmp_path / "b.w2bp")
    outcomes = set()
    cases = [blob[:k] for k in rng.integers(0, len(blob), 40)]
    for _ in range(60):
        b = bytearray(blob)
        for k in rng.integers(0, len(b), int(rng.integers(1, 4))):
            b[k] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    for hdr in (b"W2BP1 2147483000 8 1\n", b"W2BP1 99999999999999999999 8 1\n", b"W2BP1 5 16777216 2\n", b"W2BP1 -3 8 1\n",
                b"W2BP1 5 0 1\n", b"W2BP1 5 8\n", b"W2BP1 " + b"9" * 200 + b"\n"):
        cases.append(hdr + blob[blob.index(b"\n") + 1:])
    for data in cases:
        open(bad, "wb").write(data)
        try:
            w2b.unpack_vectors_file(bad, out, 1)
            outcomes.add("ok")
        except w2b.W2bError as e:
            assert e.code in (_lib.W2B_EIO, _lib.W2B_EINVAL, _lib.W2B_EUNSUPPORTED)
            outcomes.add("error")
    assert outcomes == {"ok", "error"}          # (a flipped sign bit still is a valid file)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("bitlevel", [1, 2])
@pytest.mark.parametrize("V,D", [(50, 8), (301, 200), (130, 65), (1000, 1000), (3, 4100)])
def test_device_export_packed_equals_host_twin(gpu, bitlevel, V, D):
    t = w2b.Trainer(V, D, 5, 5, bitlevel, num_threads=1, train_words=1000)
    rng = np.random.default_rng(V + D)
    u = rng.standard_normal((V, D)).astype(np.float32)
    v = rng.standard_normal((V, D)).astype(np.float32)
    u[0, :] = 0.0
    v[0, :] = -0.0                                            # u + v = +0 -> positive level
    v[1, :] = -u[1, :]
    t.set_model(u, v)
    q = t.export_quantized()
    p = t.export_packed()
    assert np.array_equal(p, w2b.pack_quantized(q, bitlevel))
    assert np.array_equal(w2b.unpack_quantized(p, D, bitlevel).view(np.uint32), q.view(np.uint32))
    t.close()


@pytest.mark.gpu
def test_device_export_packed_needs_bitlevel_1_or_2(gpu):
    t = w2b.Trainer(20, 8, 5, 5, 0, num_threads=1, train_words=1000)
    with pytest.raises(w2b.W2bError) as e:
        t.export_packed()
    assert e.value.code == _lib.W2B_EUNSUPPORTED
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bitlevel", [1, 2])
def test_cli_packed_output_and_evaluator_on_it(gpu, bitlevel, tmp_path):
    """./word2bits -packed FILE writes the model of -output at 1 / 2 bits per value; expanding it gives the -output
    file back byte for byte, and ./compute_accuracy prints the same transcript from either file."""
    out, pk = str(tmp_path / "o.bin"), str(tmp_path / "o.w2bp")
    r = subprocess.run([os.path.join(ROOT, "word2bits"), "-train", CORPUS, "-output", out, "-packed", pk, "-threads", "1",
                        "-bitlevel", str(bitlevel), "-size", "72", "-window", "4", "-negative", "5", "-iter", "2",
                        "-min-count", "2", "-binary", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    back = str(tmp_path / "back.bin")
    w2b.unpack_vectors_file(pk, back, 1)
    assert open(back, "rb").read() == open(out, "rb").read()
    assert os.path.getsize(pk) < os.path.getsize(out) / 8
    words, _ = read_vectors(out, True)
    qs = ": s\n" + "".join("%s %s %s %s\n" % tuple(words[(7 * i + k) % len(words)] for k in (1, 2, 3, 4)) for i in range(40))
    ca = os.path.join(ROOT, "compute_accuracy")
    a = subprocess.run([ca, out, "0", "0"], input=qs.encode(), capture_output=True)
    b = subprocess.run([ca, pk, "0", "0"], input=qs.encode(), capture_output=True)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and b"ACCURACY" in a.stdout
