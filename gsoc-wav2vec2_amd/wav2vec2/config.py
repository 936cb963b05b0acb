"""Hyper-parameter dataclasses for the Wav2Vec2 forward / CTC path.

Mirrors the interface of the reference's ``src/wav2vec2/config.py:6-73``: the
same 22 field names (including the mis-spelt ``kernal_sizes``, which is a JSON
key in every published ``config.json`` -- reference ``config.py:27,54``), the
same defaults, the same validation and error types, and the same JSON
round-trip (``save_pretrained`` writes ``config.json``; ``from_json`` does
``cls(**dict)`` so an unknown key raises ``TypeError``).
"""

import json
import os
from dataclasses import asdict, dataclass, field


def _default_filters():
    return [512, 512, 512, 512, 512, 512, 512]


def _default_kernels():
    return [10, 3, 3, 3, 3, 2, 2]


def _default_strides():
    return [5, 2, 2, 2, 2, 2, 2]


@dataclass
class Wav2Vec2Config:
    # transformer (reference config.py:8-17)
    vocab_size: int = 32
    dropout: float = 0.1
    hidden_size: int = 768
    num_heads: int = 12
    num_layers: int = 12
    intermediate_size: int = 3072
    is_gelu_approx: bool = False
    layer_norm_eps: float = 1e-5
    survival_prob: float = 1.0
    pad_id: int = 0

    # relative positional convolution (reference config.py:19-21)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16

    # strided conv feature extractor (reference config.py:23-30)
    filter_sizes: list = field(default_factory=_default_filters)
    kernal_sizes: list = field(default_factory=_default_kernels)
    strides: list = field(default_factory=_default_strides)
    conv_bias: bool = False

    # train-time span masking (reference config.py:32-35)
    apply_spec_augment: bool = True
    mask_time_prob: float = 0.05
    mask_time_length: int = 10

    attention_norm_type: str = "postnorm"
    feature_extractor_norm_type: str = "group"
    is_robust: bool = False

    def __post_init__(self):
        # same checks, same exception types as reference config.py:40-49
        if not (len(self.filter_sizes) == len(self.kernal_sizes) == len(self.strides)):
            raise ValueError("Length of filter_sizes, kernal_sizes, strides must match.")
        if self.hidden_size % self.num_heads != 0:
            raise ValueError("Hidden size must be perfect multiple of num_heads.")
        assert self.feature_extractor_norm_type in ["group", "layer"], \
            "Only `group` / `layer` are supported"
        assert self.attention_norm_type in ["prenorm", "postnorm"], \
            "Only `prenorm` / `postnorm` are supported"

    def save_pretrained(self, save_dir):
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "config.json"), "w") as f:
            json.dump(asdict(self), f)

    @classmethod
    def from_json(cls, path: str):
        with open(path, "r") as f:
            config_dict = json.load(f)
        return cls(**config_dict)

    # ---- helpers that are not part of the reference surface -------------
    @classmethod
    def from_hf_config(cls, hf):
        """Build the config from a HuggingFace ``Wav2Vec2Config`` dict (or the path of its ``config.json``): the
        field correspondence the reference relies on implicitly when it converts checkpoints
        (src/convert_torch_to_tf.py:128-153 picks ``Wav2Vec2Config()`` / ``RobustWav2Vec2Config()`` by hand;
        SURVEY 8c lists the verified field-for-field equality for base).  ``do_stable_layer_norm`` selects the
        prenorm (robust / xlsr) transformer, ``feat_extract_norm`` the conv-stack norm."""
        if isinstance(hf, (str, os.PathLike)):
            with open(hf, "r") as f:
                hf = json.load(f)
        act = hf.get("hidden_act", "gelu")
        if act not in ("gelu", "gelu_new", "gelu_fast", "gelu_pytorch_tanh"):
            raise NotImplementedError(f"hidden_act `{act}` has no counterpart in the reference (GELU only)")
        if hf.get("feat_extract_activation", "gelu") != "gelu":
            raise NotImplementedError("the reference's feature extractor is GELU only")
        robust = bool(hf.get("do_stable_layer_norm", False))
        fields = dict(
            vocab_size=hf.get("vocab_size", 32),
            dropout=hf.get("hidden_dropout", 0.1),
            hidden_size=hf["hidden_size"],
            num_heads=hf["num_attention_heads"],
            num_layers=hf["num_hidden_layers"],
            intermediate_size=hf["intermediate_size"],
            is_gelu_approx=act != "gelu",
            layer_norm_eps=hf.get("layer_norm_eps", 1e-5),
            survival_prob=1.0 - hf.get("layerdrop", 0.0) * 0.0,      # the reference never enables stochastic depth (config.py:16)
            pad_id=hf.get("pad_token_id", 0) or 0,
            num_conv_pos_embeddings=hf.get("num_conv_pos_embeddings", 128),
            num_conv_pos_embedding_groups=hf.get("num_conv_pos_embedding_groups", 16),
            filter_sizes=list(hf.get("conv_dim", _default_filters())),
            kernal_sizes=list(hf.get("conv_kernel", _default_kernels())),
            strides=list(hf.get("conv_stride", _default_strides())),
            conv_bias=bool(hf.get("conv_bias", False)),
            apply_spec_augment=bool(hf.get("apply_spec_augment", True)),
            mask_time_prob=hf.get("mask_time_prob", 0.05),
            mask_time_length=hf.get("mask_time_length", 10),
            attention_norm_type="prenorm" if robust else "postnorm",
            feature_extractor_norm_type=hf.get("feat_extract_norm", "group"),
            is_robust=robust,
        )
        return (RobustWav2Vec2Config if robust else Wav2Vec2Config)(**fields)

    def num_frames(self, num_samples: int) -> int:
        """Frames out of the conv stack: ``1 + (len - k) // s`` per layer
        (reference modeling.py:202-204, losses.py:47-56)."""
        n = int(num_samples)
        for k, s in zip(self.kernal_sizes, self.strides):
            n = 1 + (n - k) // s
        return n


@dataclass
class RobustWav2Vec2Config(Wav2Vec2Config):
    # overrides of reference config.py:63-73
    attention_norm_type: str = "prenorm"
    feature_extractor_norm_type: str = "layer"
    is_robust: bool = True
    conv_bias: bool = True

    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_heads: int = 16
    num_layers: int = 24
