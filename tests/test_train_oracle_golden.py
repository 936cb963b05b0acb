"""Pin the TRAINING oracle (oracle/w2v2_torch_train.py) to HuggingFace-PyTorch.

The fixtures tests/golden/train_*.npz hold the CTC loss and gradient slices of the HF import on fixed
(waveform, labels), conv stack frozen (tests/golden/make_train_golden.py; the comparator of the reference's
tests/test_wav2vec2.py:191-237 and the freeze set of src/main.py:234-237).  The oracle is the checker of the
HIP training step, so its loss and gradients are checked here, on CPU, against that external pin."""

import numpy as np
import pytest

import helpers as H
from oracle import w2v2_torch_train as TT
from wav2vec2 import variables as V

CASES = {"train_tiny_base": "tiny_base", "train_tiny_robust": "tiny_robust", "train_base_sample": "base_sample_unpadded"}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_loss_and_gradients_match_hf(name):
    g = H.golden(name)
    cfg, w = H.case_config(CASES[name]), H.case_weights(CASES[name])
    mask = g.get("attention_mask")
    div = float(g["division_factor"])
    loss, nll, logits, grads = TT.loss_and_grads(cfg, w, g["wave"], g["labels"], attention_mask=mask, p=0.0,
                                                 division_factor=div)
    assert H.max_err(logits, g["logits_f64"]) < 2e-5          # the fixture stores fp64 logits as fp32
    assert np.allclose(nll, g["nll"], rtol=1e-6, atol=1e-5)
    assert abs(loss - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))       # reference bar: 1e-3 absolute
    name_w, worst = H.grad_slice_errors(lambda n: grads[n], g)
    print(f"{name}: loss {loss:.6f}; worst gradient slice {name_w}: {worst:.2e}")
    assert worst < 1e-6
    # freeze set: exactly the conv stack (4,200,448 elements for base, SURVEY a-16) has no gradient
    frozen = sum(int(np.prod(s)) for n, (s, _) in V.variable_specs(cfg).items() if n not in grads)
    assert frozen == int(g["frozen_elements"])
    if name == "train_base_sample":
        assert frozen == 4200448
        # 90,195,104 trainable elements (SURVEY a-16) + the 768 of masked_spec_embed (trainable, unused without spec-augment)
        assert sum(int(np.prod(s)) for n, (s, _) in V.variable_specs(cfg).items() if n in grads) == 90195104 + 768
        assert grads["masked_spec_embed"] is None


def test_fixture_covers_the_gradient_kinds_survey_asks_for():
    """>= 6 gradient slices incl. lm_head, an FFN kernel, a q_proj, pos-conv weight_v / weight_g, the projection and
    a LayerNorm gamma."""
    names = set(H.train_golden_grads(H.golden("train_base_sample")))
    for want in ("lm_head/kernel", "feed_forward/intermediate_dense/kernel", "attention/q_proj/kernel",
                 "pos_conv_embed/conv/weight_v", "pos_conv_embed/conv/weight_g", "feature_projection/projection/kernel",
                 "layer_norm/gamma"):
        assert any(want in n for n in names), want
