"""The Keras-HDF5 checkpoint container without h5py (reference modeling.py:22-27,76-84).

`tests/golden/keras_layout_libhdf5.h5` was written by the REAL HDF5 library (libhdf5 1.10.6 through ctypes, in the build
container: tests/golden/make_h5_fixture.py) in the layout Keras' `save_weights("tf_model.h5")` uses, with all 213 TF variable
names of wav2vec2-base at toy widths.  wav2vec2/h5lite.py must read it bit-exactly, and its own writer must produce files its
reader -- and, checked by that script where the library exists, libhdf5's h5ls / h5dump -- read back identically."""

import os
import struct

import numpy as np
import pytest

import helpers as H
from wav2vec2 import h5lite
from wav2vec2 import variables as V
from wav2vec2.config import Wav2Vec2Config

SMALL_12L = dict(hidden_size=16, num_heads=2, num_layers=12, intermediate_size=32, filter_sizes=[8] * 7,
                 num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
FIXTURE = os.path.join(H.GOLDEN, "keras_layout_libhdf5.h5")


def _weights():
    cfg = Wav2Vec2Config(**SMALL_12L)
    return cfg, V.seeded_weights(cfg, seed=3)


def test_reads_a_file_written_by_libhdf5_bit_exactly():
    cfg, weights = _weights()
    assert len(weights) == 213                                   # the 213 variables of wav2vec2-base
    f = h5lite.File(FIXTURE)
    assert [n.decode() for n in f.root.attrs["layer_names"]] == ["wav2vec2", "dropout", "lm_head"]
    assert f.root.attrs["backend"].tobytes().rstrip(b"\0") == b"tensorflow"
    assert f["dropout"].attrs["weight_names"].size == 0
    assert len(f["wav2vec2"].attrs["weight_names"]) == 211 and len(f["lm_head"].attrs["weight_names"]) == 2
    got = h5lite.load_keras_weights(FIXTURE)
    assert len(got) == 213
    for n, a in weights.items():
        tfn = V.tf_variable_name(n)
        assert got[tfn].dtype == np.float32 and got[tfn].shape == a.shape
        assert np.array_equal(got[tfn], a), n
    # a dataset by path: weight names nest as groups
    k = f["wav2vec2/wav2vec2-ctc/wav2vec2/encoder/layers/11/feed_forward/output_dense/kernel:0"].read()
    assert np.array_equal(k, weights["encoder/layers/11/feed_forward/output_dense/kernel"])


def test_writer_roundtrip_and_structure(tmp_path):
    cfg, weights = _weights()
    named = [(V.tf_variable_name(n), a) for n, a in weights.items()]
    layers = [("wav2vec2", [x for x in named if "/lm_head/" not in x[0]]), ("dropout", []),
              ("lm_head", [x for x in named if "/lm_head/" in x[0]])]
    path = str(tmp_path / "tf_model.h5")
    h5lite.save_keras_weights(path, layers)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0               # version-0 superblock, like h5py's default
    assert struct.unpack_from("<Q", raw, 40)[0] == len(raw)               # end-of-file address
    got = h5lite.load_keras_weights(path)
    assert set(got) == {n for n, _ in named}
    for n, a in named:
        assert np.array_equal(got[n], a)
    # the encoder's 12 transformer layers need two symbol-table nodes (8 entries each) under one B-tree node
    layers_group = h5lite.File(path)["wav2vec2/wav2vec2-ctc/wav2vec2/encoder/layers"]
    assert sorted(layers_group.members(), key=int) == [str(i) for i in range(12)]


def test_large_name_lists_are_chunked_like_keras(tmp_path):
    names = [f"layer/some/rather/long/variable/name/number/{i:05d}/kernel:0" for i in range(2000)]
    weights = [(n, np.full((2,), i, np.float32)) for i, n in enumerate(names)]
    path = str(tmp_path / "big.h5")
    h5lite.save_keras_weights(path, [("layer", weights)])
    f = h5lite.File(path)
    assert "weight_names" not in f["layer"].attrs and "weight_names0" in f["layer"].attrs and "weight_names1" in f["layer"].attrs
    got = h5lite.load_keras_weights(path)
    assert len(got) == 2000 and got[names[1234]][0] == 1234


def test_other_dtypes_scalars_and_errors(tmp_path):
    path = str(tmp_path / "misc.h5")
    w = h5lite.Writer(path)
    w.create_dataset("a/ints", np.arange(6, dtype=np.int32).reshape(2, 3))
    w.create_dataset("a/doubles", np.linspace(0, 1, 5))
    w.create_dataset("scalar", np.float32(2.5))
    w.create_dataset("empty", np.zeros((0, 4), np.float32))
    w.set_attr(w.group("a"), "note", np.array(b"hello"))
    w.set_attr(w.group("a"), "values", np.array([1.5, 2.5]))
    w.close()
    f = h5lite.File(path)
    assert np.array_equal(f["a/ints"].read(), np.arange(6, dtype=np.int32).reshape(2, 3))
    assert np.array_equal(f["a/doubles"].read(), np.linspace(0, 1, 5))
    assert f["scalar"].read() == np.float32(2.5) and f["empty"].read().shape == (0, 4)
    assert f["a"].attrs["note"].tobytes() == b"hello" and list(f["a"].attrs["values"]) == [1.5, 2.5]
    with pytest.raises(KeyError):
        f["a/missing"]
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(h5lite.H5FormatError):
        h5lite.File(str(bad))
