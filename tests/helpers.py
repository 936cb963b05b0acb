"""Shared test helpers: golden fixtures, seeded configs, tolerances."""

import os

import numpy as np

from wav2vec2 import variables as V
from wav2vec2.config import RobustWav2Vec2Config, Wav2Vec2Config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = dict(hidden_size=64, num_heads=2, num_layers=2, intermediate_size=128,
            filter_sizes=[32] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)

# fp32 bar: max-abs logit error <= 1e-3 vs the oracle fixture (the reference's own
# TF-vs-HF bar, tests/test_wav2vec2.py:77-79).  We aim an order of magnitude inside it.
ATOL_BAR = 1e-3
ATOL_AIM = 2e-4


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def case_config(name):
    if name == "tiny_base":
        return Wav2Vec2Config(**TINY)
    if name == "tiny_robust":
        return RobustWav2Vec2Config(**TINY)
    if name.startswith("base"):
        return Wav2Vec2Config()
    if name.startswith("robust"):
        return RobustWav2Vec2Config()
    raise KeyError(name)


def case_weights(name, with_lm_head=True):
    return V.seeded_weights(case_config(name), seed=0, with_lm_head=with_lm_head)


# time-strides the fixture generator used for the strided stage taps (tests/golden/make_golden.py)
TAP_STRIDE = {"conv0": 997, "conv1": 499, "conv2": 251, "conv3": 127, "conv4": 61, "conv5": 31,
              "conv6": 7, "encoder_in": 13, "layer0": 13, "last_hidden": 13}


def tap_view(name, arr, full):
    return arr if full else arr[:, ::TAP_STRIDE[name]]


def max_err(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


class oracle_operands:
    """with H.oracle_operands("bf16"): run both oracles with bf16-rounded Dense / Conv1D operands."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from oracle import w2v2_oracle as O
        from oracle import w2v2_torch_train as TT
        self.prev = (O.GEMM_OPERANDS, TT.GEMM_OPERANDS)
        O.GEMM_OPERANDS = TT.GEMM_OPERANDS = self.mode

    def __exit__(self, *exc):
        from oracle import w2v2_oracle as O
        from oracle import w2v2_torch_train as TT
        O.GEMM_OPERANDS, TT.GEMM_OPERANDS = self.prev
        return False


def train_golden_grads(g):
    """{local_name: (stride, flat fp64 slice)} of a tests/golden/train_*.npz fixture (make_train_golden.py):
    whole tensors are stored as `grad:<name>`, flat strided slices as `grad:<name>@<stride>`."""
    out = {}
    for k, v in g.items():
        if not k.startswith("grad:"):
            continue
        name, _, st = k[5:].partition("@")
        out[name] = (int(st) if st else 1, np.asarray(v, dtype=np.float64).reshape(-1))
    return out


def grad_slice_errors(get_grad, g):
    """Worst relative error max|got - ref| / max|ref| over the fixture's gradient slices; `get_grad(name)` returns
    the full gradient in the TF variable layout."""
    worst = ("", 0.0)
    for name, (st, ref) in train_golden_grads(g).items():
        got = np.asarray(get_grad(name), dtype=np.float64).reshape(-1)[::st]
        assert got.shape == ref.shape, name
        scale = max(1e-3, float(np.abs(ref).max()))
        e = float(np.abs(got - ref).max()) / scale
        if e > worst[1]:
            worst = (name, e)
    return worst
