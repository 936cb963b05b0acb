"""Wav2Vec2Model / Wav2Vec2ForCTC -- the drop-in model surface.

Host-side mirror of the reference's ``src/wav2vec2/modeling.py``: same class
names, constructor arguments, call signature ``model(batch, attention_mask=
None, training=False)``, ``.config``, ``from_pretrained`` / ``save_pretrained``,
``freeze_feature_extractor`` and a Keras-like ``.variables`` list carrying the
reference's TF variable names (convert_torch_to_tf.py:24-44).  All arithmetic
runs in the hand-written HIP library behind include/w2v2.h; torch-ROCm tensors
only carry the device buffers and the stream.

What is deliberately different from Keras:
  * outputs are torch CUDA tensors (a thin subclass whose ``.numpy()`` copies to
    host, so reference-style ``model(x).numpy()`` keeps working);
  * weights are stored as ``tf_model.npz`` (TF variable names as keys) -- the
    Keras-HDF5 container needs h5py, which this image lacks; ``tf_model.h5`` is
    read when h5py is importable;
  * ``training=True`` (dropout / spec-augment / stochastic depth) is not built
    yet and raises instead of silently running the inference graph.
"""

import ctypes as C
import logging
import os
from dataclasses import replace

import numpy as np

from . import _native as N
from . import variables as V
from .config import Wav2Vec2Config

logger = logging.getLogger(__name__)


def _torch():
    import torch
    return torch


def _require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "wav2vec2 (MI355X build): no HIP device is visible.  The forward / CTC path runs only "
            "on the GPU through lib/libw2v2.so; there is no CPU fallback.")
    return torch


class DeviceTensor:
    """Factory for the returned tensor type: a torch.Tensor subclass whose
    ``.numpy()`` first copies to host (TF eager tensors expose ``.numpy()``)."""
    _cls = None

    @classmethod
    def wrap(cls, t):
        torch = _torch()
        if cls._cls is None:
            class W2V2Tensor(torch.Tensor):
                def numpy(self):
                    return self.detach().as_subclass(torch.Tensor).cpu().numpy()
            cls._cls = W2V2Tensor
        return t.as_subclass(cls._cls)


class Variable:
    """Minimal stand-in for tf.Variable: ``.name``, ``.shape``, ``.numpy()``,
    ``.assign()`` -- what convert_torch_to_tf.py:79-121 and callers touch."""

    def __init__(self, model, local_name, shape, trainable=True):
        self._model = model
        self.local_name = local_name
        self.name = V.tf_variable_name(local_name, with_lm_head=model._prefix_with_head)
        self.shape = tuple(shape)
        self.trainable = trainable

    def numpy(self):
        return self._model._get_param(self.local_name, self.shape)

    def assign(self, value):
        self._model._set_param(self.local_name, np.asarray(value, dtype=np.float32))
        return self

    def __repr__(self):
        return f"<Variable {self.name} shape={self.shape}>"


def hf_checkpoint_file(save_dir):
    """Path of the HuggingFace weight file in `save_dir` (`model.safetensors` preferred, else `pytorch_model.bin`)."""
    for f in ("model.safetensors", "pytorch_model.bin"):
        p = os.path.join(save_dir, f)
        if os.path.exists(p):
            return p
    return None


def read_hf_state_dict(save_dir):
    """{hf_key: numpy array} from a HuggingFace-PyTorch Wav2Vec2 checkpoint directory."""
    path = hf_checkpoint_file(save_dir)
    if path is None:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {save_dir}")
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return dict(load_file(path))
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def convert_hf_checkpoint(hf_dir, save_dir, with_lm_head=True):
    """The job of the reference's src/convert_torch_to_tf.py without TensorFlow and without a GPU: read a
    HuggingFace-PyTorch checkpoint directory, apply the HF -> TF name / layout map (file:12-44,110-117) and write
    the package's own checkpoint (`config.json` with the reference's 22 fields + weights keyed by TF variable names).
    Returns the config."""
    config = Wav2Vec2Config.from_hf_config(os.path.join(hf_dir, "config.json"))
    weights = V.from_hf_state_dict(read_hf_state_dict(hf_dir), config, with_lm_head=with_lm_head)
    config.save_pretrained(save_dir)
    np.savez(os.path.join(save_dir, "tf_model.npz"), **{V.tf_variable_name(n, with_lm_head): a for n, a in weights.items()})
    return config


class TFKerasModel:
    """Shared plumbing (reference modeling.py:21-102 ``TFKerasModel``)."""

    _with_lm_head = False        # native model computes the LM head
    _prefix_with_head = False    # variable names carry the "wav2vec2-ctc/" prefix

    def _build_native(self, config, seed=0):
        _require_gpu()
        lib = N.load()
        self._lib = lib
        self._handle = C.c_void_p()
        cfg = N.make_config(config, self._with_lm_head)
        N.check(lib.w2v2_create(C.byref(cfg), C.byref(self._handle)), "w2v2_create")
        self._specs = V.variable_specs(config, with_lm_head=self._with_lm_head)
        self._dirty = True
        # random initialisation (the reference builds variables with Keras initialisers
        # by running a dummy forward, modeling.py:86-102)
        self.set_weights(V.seeded_weights(config, seed=seed, with_lm_head=self._with_lm_head))
        self._variables = [Variable(self, n, s) for n, (s, _) in self._specs.items()]

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.w2v2_destroy(self._handle)
                self._handle = None
        except Exception:  # noqa: BLE001
            pass

    # ---- variables ---------------------------------------------------------
    @property
    def variables(self):
        return list(self._variables)

    @property
    def trainable_variables(self):
        return [v for v in self._variables if v.trainable]

    def _set_param(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if name not in self._specs:
            raise KeyError(f"unknown variable `{name}`")
        if tuple(arr.shape) != tuple(self._specs[name][0]):
            raise ValueError(f"`{name}`: expected shape {self._specs[name][0]}, got {arr.shape}")
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        N.check(self._lib.w2v2_set_param(self._handle, name.encode(), N.ptr(arr), shape, arr.ndim),
                f"w2v2_set_param({name})")
        self._dirty = True

    def _get_param(self, name, shape):
        out = np.empty(shape, dtype=np.float32)
        N.check(self._lib.w2v2_get_param(self._handle, name.encode(), N.ptr(out), out.size),
                f"w2v2_get_param({name})")
        return out

    def set_weights(self, weights):
        """``weights``: mapping local-name or TF-name -> array (TF layout)."""
        for k, v in weights.items():
            self._set_param(V.local_name_from_tf(k), v)

    def load_hf_state_dict(self, state_dict):
        """Load a HuggingFace-PyTorch Wav2Vec2 ``state_dict`` through the reference's HF -> TF name / layout
        map (src/convert_torch_to_tf.py:12-44,110-117) -- the counterpart of ``get_tf_pretrained_model``
        without TensorFlow.  Keys missing from the dict raise ``KeyError``."""
        self.set_weights(V.from_hf_state_dict(state_dict, self.config, with_lm_head=self._with_lm_head))

    def get_weights(self):
        return {n: self._get_param(n, s) for n, (s, _) in self._specs.items()}

    def _finalize(self):
        if self._dirty:
            N.check(self._lib.w2v2_finalize(self._handle, N.current_stream()), "w2v2_finalize")
            self._dirty = False

    def set_trainable(self, name_prefix, trainable):
        """Keras `.trainable` for every variable whose local name starts with `name_prefix`
        (main.py:210,234-237 toggle whole sub-layers this way)."""
        for v in self._variables:
            if v.local_name.startswith(name_prefix):
                v.trainable = bool(trainable)
        N.check(self._lib.w2v2_set_trainable(self._handle, name_prefix.encode(), int(bool(trainable))), "w2v2_set_trainable")

    # ---- arithmetic of the dense contractions --------------------------------
    PRECISIONS = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "mixed_bfloat16": 1, "bf16x3": 2}

    def set_precision(self, precision):
        """"fp32" (default: the reference's arithmetic) or "bf16" (Conv1D layers 1..6 and every Dense take
        bf16-rounded operands with fp32 accumulation, forward and backward -- the mixed-precision policy the
        bf16 fine-tune configurations ask for; variables, activations and optimizer state stay fp32), or "bf16x3"
        (fp32 operands split exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32
        accumulation -- fp32-level results at the bf16 matrix cores' rate; forward GEMMs and attention, and the
        data-gradient GEMMs of the training step; csrc/gemm_split.hip, attention_split.hip)."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(set(self.PRECISIONS))}, got {precision!r}")
        N.check(self._lib.w2v2_set_precision(self._handle, self.PRECISIONS[precision]), "w2v2_set_precision")

    @property
    def precision(self):
        return {0: "fp32", 1: "bf16", 2: "bf16x3"}[self._lib.w2v2_get_precision(self._handle)]

    # ---- persistence (reference modeling.py:22-27, 41-84) -------------------
    def save_weights(self, path):
        arrays = {V.tf_variable_name(n, self._prefix_with_head): a for n, a in self.get_weights().items()}
        if path.endswith(".h5"):
            path = path[:-3] + ".npz"
        np.savez(path, **arrays)

    def load_weights(self, path):
        if path.endswith(".h5") and not os.path.exists(path) and os.path.exists(path[:-3] + ".npz"):
            path = path[:-3] + ".npz"
        if path.endswith(".npz"):
            with np.load(path) as z:
                self.set_weights({k: z[k] for k in z.files})
            return
        try:
            import h5py  # noqa: F401
        except ImportError as e:
            raise NotImplementedError(
                "reading Keras-HDF5 `tf_model.h5` needs h5py, which is not installed; "
                "use the `tf_model.npz` written by save_pretrained") from e
        import h5py
        found = {}
        with h5py.File(path, "r") as f:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    found[name] = np.asarray(obj)
            f.visititems(visit)
        ours = {}
        for n in self._specs:
            tfn = V.tf_variable_name(n, self._prefix_with_head)
            hits = [k for k in found if k.endswith(tfn)]
            if not hits:
                raise KeyError(f"`{tfn}` not found in {path}")
            ours[n] = found[hits[0]]
        self.set_weights(ours)

    def save_pretrained(self, save_dir):
        """config.json + weights, as reference modeling.py:22-27 (weights container: npz)."""
        self.config.save_pretrained(save_dir)
        self.save_weights(os.path.join(save_dir, "tf_model.h5"))

    @classmethod
    def from_pretrained(cls, model_id, **config_kwargs):
        """Load from a local directory (reference modeling.py:41-84).  The
        reference downloads from the HuggingFace Hub when the directory does not
        exist; this build has no network path and raises the same ValueError the
        reference raises on a failed download."""
        save_dir = model_id
        if not os.path.isdir(save_dir):
            raise ValueError(f"Couldn't download model weights from https://huggingface.co/{model_id}")
        print(f"Loading weights locally from `{save_dir}`")
        input_shape = config_kwargs.pop("input_shape", (1, 2048))
        has_own = any(os.path.exists(os.path.join(save_dir, f)) for f in ("tf_model.h5", "tf_model.npz"))
        if not has_own and hf_checkpoint_file(save_dir):
            # a HuggingFace-PyTorch checkpoint directory: what src/convert_torch_to_tf.py converts (file:92-123)
            config = replace(Wav2Vec2Config.from_hf_config(os.path.join(save_dir, "config.json")), **config_kwargs)
            model = cls(config, input_shape=input_shape)
            model.load_hf_state_dict(read_hf_state_dict(save_dir))
            print("Total number of loaded variables:", len(model.variables))
            return model
        config = Wav2Vec2Config.from_json(os.path.join(save_dir, "config.json"))
        config = replace(config, **config_kwargs)
        model = cls(config, input_shape=input_shape)
        model.load_weights(os.path.join(save_dir, "tf_model.h5"))
        print("Total number of loaded variables:", len(model.variables))
        return model

    # ---- forward -------------------------------------------------------------
    def _prepare(self, batch, attention_mask):
        torch = _require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
        if not isinstance(batch, torch.Tensor):
            batch = torch.as_tensor(np.asarray(batch, dtype=np.float32))
        batch = batch.to(device=dev, dtype=torch.float32).contiguous()
        if batch.dim() == 1:
            batch = batch[None, :]
        if batch.dim() != 2:
            raise ValueError(f"`batch` must be (batch_size, seqlen), got {tuple(batch.shape)}")
        if attention_mask is not None:
            if not isinstance(attention_mask, torch.Tensor):
                attention_mask = torch.as_tensor(np.asarray(attention_mask))
            attention_mask = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
            if tuple(attention_mask.shape) != tuple(batch.shape):
                raise ValueError("`attention_mask` must have the shape of `batch`")
        return batch, attention_mask

    def _forward(self, batch, attention_mask, training, out_channels):
        if training:
            raise NotImplementedError(
                "the training-mode forward (dropout, spec-augment, stochastic depth) needs a seed and host-drawn "
                "masks: use wav2vec2.Trainer(model, loss).forward / .step instead of model(x, training=True)")
        # the tuple call convention of the reference's fixed-length export wrappers (export2hub.py:40-57):
        # model((speech, attention_mask))
        if isinstance(batch, tuple) and len(batch) == 2 and attention_mask is None:
            batch, attention_mask = batch
        # same (non-fatal) warnings as reference modeling.py:183-186
        if self.config.is_robust and attention_mask is None:
            logger.warning("You should pass `attention_mask` when working with Wav2Vec2 new checkpoints")
        elif not self.config.is_robust and attention_mask is not None:
            logger.warning("You should not pass `attention_mask` when working with checkpoints based on `wav2vec2-base`")
        torch = _torch()
        batch, attention_mask = self._prepare(batch, attention_mask)
        B, L = batch.shape
        T = int(self._lib.w2v2_num_frames(self._handle, L))
        if T < 1:
            raise ValueError(f"input of {L} samples is shorter than the feature extractor's receptive field")
        self._finalize()
        out = torch.empty((B, T, out_channels), device=batch.device, dtype=torch.float32)
        N.check(self._lib.w2v2_forward(self._handle, N.ptr(batch), B, L, N.ptr(attention_mask), N.ptr(out),
                                       N.current_stream()), "w2v2_forward")
        return DeviceTensor.wrap(out)

    def predict(self, batch, attention_mask=None):
        return self(batch, attention_mask=attention_mask, training=False)

    # ---- introspection used by the parity tests / bench -----------------------
    def activation(self, name):
        shape = (C.c_int64 * 3)()
        N.check(self._lib.w2v2_activation_info(self._handle, name.encode(), shape), "w2v2_activation_info")
        out = np.empty(tuple(shape), dtype=np.float32)
        N.check(self._lib.w2v2_copy_activation(self._handle, name.encode(), N.ptr(out), out.size,
                                               N.current_stream()), "w2v2_copy_activation")
        return out

    def profile(self, enable=True, families=None):
        """Bracket kernel launches with HIP events; `families` (names as in profile_read) limits the
        instrumentation to those kernel families, None = all."""
        mask = 0
        if families:
            names = []
            for i in range(self._lib.w2v2_profile_num_families()):
                nm = C.c_char_p()
                n, a, b, c = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
                N.check(self._lib.w2v2_profile_read(self._handle, i, C.byref(nm), C.byref(n), C.byref(a), C.byref(b), C.byref(c)))
                names.append(nm.value.decode())
            for f in families:
                mask |= 1 << names.index(f)
        N.check(self._lib.w2v2_profile_families(self._handle, mask), "w2v2_profile_families")
        N.check(self._lib.w2v2_profile_enable(self._handle, int(enable)), "w2v2_profile_enable")

    def profile_reset(self):
        N.check(self._lib.w2v2_profile_reset(self._handle), "w2v2_profile_reset")

    def profile_read(self):
        """{family: dict(launches, ms, flops, bytes)} over the recorded launches."""
        out = {}
        for i in range(self._lib.w2v2_profile_num_families()):
            name = C.c_char_p()
            n = C.c_int64()
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            N.check(self._lib.w2v2_profile_read(self._handle, i, C.byref(name), C.byref(n), C.byref(ms),
                                                C.byref(fl), C.byref(by)), "w2v2_profile_read")
            out[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out

    def num_frames(self, num_samples):
        return int(self._lib.w2v2_num_frames(self._handle, int(num_samples)))


class Wav2Vec2Model(TFKerasModel):
    """Backbone: waveform (B, L) -> hidden states (B, T, hidden)
    (reference modeling.py:105-214)."""

    def __init__(self, config: Wav2Vec2Config, input_shape=(1, 246000), name="wav2vec2"):
        if not isinstance(config, Wav2Vec2Config):
            raise ValueError("`config` must be an instace of `Wave2Vec2Config`")
        self.name = name
        self.config = config
        self.hidden_size = config.hidden_size
        self.is_robust = config.is_robust
        self.kernal_sizes = config.kernal_sizes
        self.strides = config.strides
        self.input_shape = input_shape
        self._build_native(config)

    def __call__(self, batch, attention_mask=None, training=False):
        return self._forward(batch, attention_mask, training, self.config.hidden_size)

    call = __call__

    def freeze_feature_extractor(self):
        """Marks the 7 conv layers non-trainable (reference modeling.py:211-214)."""
        self.set_trainable("feature_extractor/", False)


class Wav2Vec2ForCTC(TFKerasModel):
    """Backbone + dropout (identity at inference) + ``lm_head`` Dense(H -> vocab)
    (reference modeling.py:217-255).  Returns logits (B, T, vocab)."""

    _with_lm_head = True
    _prefix_with_head = True

    def __init__(self, config: Wav2Vec2Config, input_shape=(1, 246000), name="wav2vec2-ctc"):
        if not isinstance(config, Wav2Vec2Config):
            raise ValueError("`config` must be an instace of `Wave2Vec2Config`.")
        self.name = name
        self.config = config
        self.pad_id = config.pad_id
        self.input_shape = input_shape
        self._build_native(config)

    def __call__(self, batch, attention_mask=None, training=False):
        return self._forward(batch, attention_mask, training, self.config.vocab_size)

    call = __call__

    def freeze_feature_extractor(self):
        self.set_trainable("feature_extractor/", False)
