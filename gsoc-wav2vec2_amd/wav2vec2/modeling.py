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
  * ``model(x, training=True)`` runs the training-mode forward (dropout, spec-augment,
    stochastic depth -- reference modeling.py:169-209,239-255) with randomness drawn from a
    model-level seeded generator (``model.set_seed``); gradients, the optimizer and the
    data-parallel all-reduce live in ``wav2vec2.Trainer``;
  * the Keras object graph callers touch -- ``model.layers``, ``layer.trainable``,
    ``model.trainable`` (src/main.py:210,232-237) -- is rebuilt over the flat variable
    inventory by ``wav2vec2/layers.py``;
  * weights are written as a Keras-layout HDF5 file ``tf_model.h5`` by the package's own
    dependency-free HDF5 reader / writer (``wav2vec2/h5lite.py``; this image has no h5py);
    ``tf_model.npz`` (TF variable names as keys) is still read.
"""

import ctypes as C
import logging
import os
from dataclasses import replace

import numpy as np

from . import _native as N
from . import variables as V
from .config import Wav2Vec2Config
from .layers import Layer, attach, build_backbone_layers
from .spec_augment import compute_mask_indices

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
    ``.assign()``, ``.trainable`` -- what convert_torch_to_tf.py:79-121 and callers touch.
    A variable trains only if its own flag and every layer above it are trainable (Keras)."""

    def __init__(self, model, local_name, shape, trainable=True):
        self._model = model
        self.local_name = local_name
        self.name = V.tf_variable_name(local_name, with_lm_head=model._prefix_with_head)
        self.shape = tuple(shape)
        self._own_trainable = bool(trainable)
        self._gates = []                    # the chain of layers above this variable (layers.attach)

    @property
    def trainable(self):
        return self._own_trainable and all(g._trainable for g in self._gates)

    @trainable.setter
    def trainable(self, value):
        self._own_trainable = bool(value)
        self._model._sync_trainable()

    def numpy(self):
        return self._model._get_param(self.local_name, self.shape)

    def assign(self, value):
        self._model._set_param(self.local_name, np.asarray(value, dtype=np.float32))
        return self

    def __repr__(self):
        return f"<Variable {self.name} shape={self.shape}>"


class BackboneLayer(Layer):
    """The ``Wav2Vec2Model`` inside ``Wav2Vec2ForCTC`` (``self.model``, reference modeling.py:227)."""

    def freeze_feature_extractor(self):
        for layer in self.feature_extractor:                      # reference modeling.py:211-214
            layer.trainable = False


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
    the reference's checkpoint layout (`config.json` with its 22 fields + `tf_model.h5`, a Keras-layout HDF5 weight file).
    Returns the config."""
    config = Wav2Vec2Config.from_hf_config(os.path.join(hf_dir, "config.json"))
    weights = V.from_hf_state_dict(read_hf_state_dict(hf_dir), config, with_lm_head=with_lm_head)
    config.save_pretrained(save_dir)
    from . import h5lite
    h5lite.save_keras_weights(os.path.join(save_dir, "tf_model.h5"), V.keras_layers(config, weights, with_lm_head))
    return config


class TFKerasModel(Layer):
    """Shared plumbing (reference modeling.py:21-102 ``TFKerasModel``).  The model is itself the root ``Layer`` of the
    Keras-like object graph (``.layers``, ``.trainable``, ``.variables``; wav2vec2/layers.py)."""

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
        self._pushed_trainable = {}
        self._build_layers()
        self.set_seed(seed)

    def _build_layers(self):
        """The reference's Keras layer tree over the flat inventory (wav2vec2/layers.py)."""
        fe, proj, encoder = build_backbone_layers(self.config, self._sync_trainable)
        backbone_attrs = dict(feature_extractor=fe, feature_projection=proj, encoder=encoder)
        if self._with_lm_head:
            backbone = BackboneLayer("wav2vec2", ["masked_spec_embed"], fe + [proj, encoder], self._sync_trainable,
                                     config=self.config, **backbone_attrs)
            dropout = Layer("dropout", on_change=self._sync_trainable, rate=self.config.dropout)
            lm_head = Layer("lm_head", ["lm_head/"], on_change=self._sync_trainable)
            Layer.__init__(self, self.name, (), [backbone, dropout, lm_head], self._sync_trainable)
            self.model, self.dropout, self.lm_head = backbone, dropout, lm_head
        else:
            Layer.__init__(self, self.name, ["masked_spec_embed"], fe + [proj, encoder], self._sync_trainable)
            self.__dict__.update(backbone_attrs)
        attach(self, self._variables)

    def _native_inventory(self):
        """Variable names in the native library's own inventory order (w2v2_param_info)."""
        if getattr(self, "_native_names", None) is None:
            names = []
            for i in range(self._lib.w2v2_num_params(self._handle)):
                name, shape, rank = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
                N.check(self._lib.w2v2_param_info(self._handle, i, C.byref(name), shape, C.byref(rank)), "w2v2_param_info")
                names.append(name.value.decode())
            if sorted(names) != sorted(self._specs):
                raise RuntimeError("native variable inventory differs from wav2vec2.variables.variable_specs")
            self._native_names = names
        return self._native_names

    def _sync_trainable(self):
        """Push the effective per-variable flags (own flag AND every layer above) to the native training state: the whole
        vector in ONE call (w2v2_set_trainable_flags), and only when some flag differs from what was pushed last -- an
        inference-only model whose flags never leave the all-trainable default never creates the training state."""
        flags = {v.local_name: bool(v.trainable) for v in self._variables}
        if all(self._pushed_trainable.get(n, True) == f for n, f in flags.items()):
            return
        vec = (C.c_uint8 * len(flags))(*[int(flags[n]) for n in self._native_inventory()])
        N.check(self._lib.w2v2_set_trainable_flags(self._handle, vec, len(flags)), "w2v2_set_trainable_flags")
        self._pushed_trainable = flags

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
        """Set the own `.trainable` flag of every variable whose local name starts with `name_prefix` (a flat alternative to
        the layer objects; the layers' own switches still gate the result)."""
        hits = [v for v in self._variables if v.local_name.startswith(name_prefix)]
        if not hits:
            raise KeyError(f"set_trainable: no variable starts with `{name_prefix}`")
        for v in hits:
            v._own_trainable = bool(trainable)
        self._sync_trainable()

    def freeze_feature_extractor(self):
        """Marks the 7 conv layers non-trainable (reference modeling.py:211-214)."""
        for layer in self.feature_extractor if not self._with_lm_head else self.model.feature_extractor:
            layer._trainable = False
        self._sync_trainable()

    def summary(self, print_fn=print):
        """Keras-like one-line-per-layer summary (main.py:211,238 call it after changing the trainable set)."""
        rows = [(l.name, l.count_params(), sum(int(np.prod(v.shape)) for v in l.trainable_variables)) for l in self._walk() if l is not self]
        print_fn(f'Model: "{self.name}"')
        for name, n, nt in rows:
            print_fn(f"  {name:<48s} params {n:>12,d}   trainable {nt:>12,d}")
        total = self.count_params()
        tr = sum(int(np.prod(v.shape)) for v in self.trainable_variables)
        print_fn(f"Total params: {total:,d}\nTrainable params: {tr:,d}\nNon-trainable params: {total - tr:,d}")

    # ---- randomness of the training-mode forward ------------------------------------------------
    def set_seed(self, seed):
        """Seed of the model-level generator behind `model(x, training=True)`: dropout masks are a counter-based hash of
        (seed, call number, site, element); spec-augment spans and stochastic-depth draws come from a host RandomState --
        the reference draws both with numpy / TF global generators at call time (spec_augment.py:14,53;
        tensorflow_addons.py:381)."""
        self._seed = int(seed)
        self._train_calls = 0
        self._train_rng = np.random.RandomState(self._seed & 0xFFFFFFFF)

    # ---- arithmetic of the dense contractions --------------------------------
    PRECISIONS = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "mixed_bfloat16": 1, "bf16x3": 2, "f16x2": 3}

    def set_precision(self, precision):
        """"fp32" (default: the reference's arithmetic) or "bf16" (Conv1D layers 1..6 and every Dense take
        bf16-rounded operands with fp32 accumulation, forward and backward -- the mixed-precision policy the
        bf16 fine-tune configurations ask for; variables, activations and optimizer state stay fp32), or "bf16x3"
        (fp32 operands split exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32
        accumulation -- fp32-level results at the bf16 matrix cores' rate; forward GEMMs and attention, and the
        data-gradient GEMMs of the training step; csrc/gemm_split.hip, gemm_split_sw.hip, attention_split.hip), or "f16x2"
        (inference forward: every GEMM operand as TWO fp16 terms, three MFMA products per fp32 product -- half the matrix work
        of "bf16x3" at a measured error at or below the fp32 kernel's).

        CONTRACT of "f16x2": activations must stay below 4094 in magnitude.  A forward that meets a larger one saturates it, sets a
        sticky device flag and STILL RETURNS logits -- they are then not fp32-grade.  `model(x)` does not read the flag (that would
        synchronise every forward); a caller that cannot bound its activations must poll `range_overflow()` after the forwards
        it cares about and rerun those in "bf16x3" / "fp32".  Training in "f16x2" is refused (w2v2_train_forward)."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(set(self.PRECISIONS))}, got {precision!r}")
        N.check(self._lib.w2v2_set_precision(self._handle, self.PRECISIONS[precision]), "w2v2_set_precision")

    OPTIONS = {"bf16_shadows": 0, "keep_activations": 1, "split_planes": 2, "wgrad_stream": 3, "defer_folds": 4}      # W2V2_OPT_* of include/w2v2.h

    def set_option(self, name, value):
        """Per-model switches of the bf16 precision mode (include/w2v2.h: w2v2_set_option): "bf16_shadows" (default on; off =
        every GEMM rounds its fp32 operands itself, same bits, slower) and "keep_activations" (default off; on = stage outputs
        that are normally written only as bf16 -- or, in "bf16x3" / "f16x2", only as operand planes -- keep their fp32 copy so
        `activation(name)` can tap them); "wgrad_stream" (default off; bf16 training backward: the layers' weight-gradient GEMMs run on a second HIP stream
        of the model, same bits); "split_planes" (default on; off = the "bf16x3" / "f16x2" forward GEMMs split fp32 rows
        in registers instead of streaming planes written by their producers: the round-4 path, for A/B measurements);
        "defer_folds" (default on; training backward: the nine small reductions that finish an encoder layer's gradients run as one
        launch in front of the layer's bucket event; off = one launch behind each producer, same bits)."""
        if name not in self.OPTIONS:
            raise KeyError(f"unknown option {name!r}; one of {sorted(self.OPTIONS)}")
        N.check(self._lib.w2v2_set_option(self._handle, self.OPTIONS[name], int(bool(value))), "w2v2_set_option")

    def get_option(self, name):
        return bool(self._lib.w2v2_get_option(self._handle, self.OPTIONS[name]))

    @property
    def precision(self):
        return {0: "fp32", 1: "bf16", 2: "bf16x3", 3: "f16x2"}[self._lib.w2v2_get_precision(self._handle)]

    def range_overflow(self):
        """precision "f16x2": True if a forward since the last call met an activation beyond fp16's scaled range (|x| >= 4094) --
        its logits are then not fp32-grade; rerun in "bf16x3" or "fp32".  Clears the flag; synchronises the stream."""
        import ctypes
        flag = ctypes.c_int32(0)
        N.check(self._lib.w2v2_range_overflow(self._handle, ctypes.byref(flag), N.current_stream()), "w2v2_range_overflow")
        return bool(flag.value)

    # ---- persistence (reference modeling.py:22-27, 41-84) -------------------
    def _keras_layers(self, weights):
        """[(layer_name, [(TF variable name, array)])]: groups in the order of the reference model's `.layers`, weights inside a
        group in the order Keras lists them (wav2vec2/variables.py::keras_weight_order) -- what `load_weights` of a Keras-layout
        HDF5 file zips against, position by position."""
        return V.keras_layers(self.config, weights, self._prefix_with_head)

    def save_weights(self, path):
        """Keras' rule (`Model.save_weights`): a path ending in `.h5` / `.hdf5` / `.keras` is a Keras-layout HDF5 weight file,
        written by the package's own HDF5 writer (wav2vec2/h5lite.py) -- the container of the reference's `tf_model.h5`
        (modeling.py:26); any other path is the PREFIX of a TensorFlow checkpoint (`<path>.index` + `<path>.data-00000-of-00001`,
        wav2vec2/tfckpt.py) -- what the reference's ModelCheckpoint writes to `.../tf_model` (training_utils.py:32-45).
        `*.npz`: numpy archive keyed by TF variable names."""
        weights = self.get_weights()
        if path.endswith(".npz"):
            np.savez(path, **{V.tf_variable_name(n, self._prefix_with_head): a for n, a in weights.items()})
            return
        if path.endswith((".h5", ".hdf5", ".keras")):
            from . import h5lite
            h5lite.save_keras_weights(path, self._keras_layers(weights))
            return
        from . import tfckpt
        # (checkpoint keys are variable names without the `:0` output suffix)
        tfckpt.write_checkpoint(path, {V.tf_variable_name(n, self._prefix_with_head).rsplit(":", 1)[0]: a for n, a in weights.items()})

    def _match_by_name(self, found, where):
        """{our name: array} out of {TF variable name (possibly under extra leading scopes, possibly `:0`-suffixed): array}."""
        bare = lambda k: k[:-2] if k.endswith(":0") else k        # HDF5 weight names carry the suffix, checkpoint keys do not
        found = {bare(k): v for k, v in found.items()}
        ours = {}
        for n in self._specs:
            tfn = bare(V.tf_variable_name(n, self._prefix_with_head))
            hits = [k for k in found if k == tfn or k.endswith("/" + tfn)]
            if not hits:
                # a backbone file loaded into the CTC model (or the reverse): same variables under the other prefix
                alt = bare(V.tf_variable_name(n, not self._prefix_with_head))
                hits = [k for k in found if k == alt or k.endswith("/" + alt)]
            if not hits:
                raise KeyError(f"`{tfn}` not found in {where}")
            ours[n] = found[hits[0]]
        return ours

    def load_weights(self, path):
        """Reads `tf_model.h5` (Keras HDF5 weight file; h5lite, no h5py needed), a TensorFlow checkpoint prefix such as
        `.../tf_model` (src/main.py:132; name-based or object-based, wav2vec2/tfckpt.py) or a `.npz` of TF variable names.
        Variables are matched BY NAME (the TF variable names of convert_torch_to_tf.py:24-44), a missing one raises KeyError."""
        if path.endswith(".h5") and not os.path.exists(path) and os.path.exists(path[:-3] + ".npz"):
            path = path[:-3] + ".npz"
        if path.endswith(".npz"):
            with np.load(path) as z:
                self.set_weights({k: z[k] for k in z.files})
            return
        from . import tfckpt
        if not os.path.isfile(path) and tfckpt.is_checkpoint(path):
            self.set_weights(self._match_by_name(tfckpt.read_checkpoint(path), path + ".index"))
            return
        from . import h5lite
        self.set_weights(self._match_by_name(h5lite.load_keras_weights(path), path))

    def save_pretrained(self, save_dir):
        """config.json + `tf_model.h5`, as reference modeling.py:22-27."""
        self.config.save_pretrained(save_dir)
        self.save_weights(os.path.join(save_dir, "tf_model.h5"))

    def push_to_hub(self, directory: str, model_id: str):
        """Upload a `save_pretrained` directory to the HuggingFace Hub (reference modeling.py:29-39, which calls
        `ModelHubMixin.push_to_hub(directory, model_id=...)`).  Needs `huggingface_hub` and network access; whatever the
        client raises (no token, no network) is passed on."""
        from huggingface_hub import HfApi
        return HfApi().upload_folder(folder_path=directory, repo_id=model_id)

    @staticmethod
    def _download(model_id):
        """`config.json` + `tf_model.h5` of a Hub repository into the local cache (reference modeling.py:57-74 shells out to
        wget); returns the directory.  Any failure -- there is no network on the build / GPU boxes -- becomes the reference's
        ValueError."""
        try:
            from huggingface_hub import snapshot_download
            return snapshot_download(model_id, allow_patterns=["config.json", "tf_model.h5"])
        except Exception as e:  # noqa: BLE001
            raise ValueError(f"Couldn't download model weights from https://huggingface.co/{model_id}") from e

    @classmethod
    def from_pretrained(cls, model_id, **config_kwargs):
        """Load from a local directory, or from the HuggingFace Hub when `model_id` is not a directory (reference
        modeling.py:41-84).  A failed download raises the same ValueError the reference raises."""
        save_dir = model_id
        if not os.path.isdir(save_dir):
            print(f"Downloading model weights from https://huggingface.co/{model_id} ... ", end="")
            save_dir = cls._download(model_id)
            print("Done")
        else:
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

    def _train_forward(self, batch, attention_mask=None, dropout=None, apply_spec_augment=None, spec_mask=None, sd_keep=None,
                       step_seed=None, rng=None):
        """`call(training=True)` (reference modeling.py:169-209,239-255): Dropout at every Dropout layer, spec-augment on
        the projected features when `config.apply_spec_augment`, StochasticDepth on the FFN branch.  Explicit `spec_mask`
        / `sd_keep` / `step_seed` override the draws (parity tests feed the same masks to the oracle).  Saves what
        `Trainer.backward` needs.  Returns (output, dict of the randomness used)."""
        torch = _torch()
        cfg = self.config
        batch, attention_mask = self._prepare(batch, attention_mask)
        B, L = batch.shape
        T = self.num_frames(L)
        if T < 1:
            raise ValueError(f"input of {L} samples is shorter than the feature extractor's receptive field")
        self._finalize()
        rng = self._train_rng if rng is None else rng
        p = cfg.dropout if dropout is None else dropout
        spec_on = cfg.apply_spec_augment if apply_spec_augment is None else apply_spec_augment
        if spec_mask is None and spec_on:
            spec_mask = compute_mask_indices((B, T), cfg.mask_time_prob, cfg.mask_time_length, min_masks=2, rng=rng)
        if sd_keep is None and cfg.survival_prob < 1.0:
            # one Bernoulli scalar per StochasticDepth call (tensorflow_addons.py:381)
            sd_keep = (rng.uniform(size=cfg.num_layers) < cfg.survival_prob).astype(np.float32)
        sm = None if spec_mask is None else np.ascontiguousarray(spec_mask, dtype=np.uint8).reshape(-1)
        sd = None if sd_keep is None else np.ascontiguousarray(sd_keep, dtype=np.float32)
        if step_seed is None:
            step_seed = self._seed * 1000003 + self._train_calls
            self._train_calls += 1
        seed = int(step_seed) & 0xFFFFFFFFFFFFFFFF
        width = cfg.vocab_size if self._with_lm_head else cfg.hidden_size
        out = torch.empty((B, T, width), device=batch.device, dtype=torch.float32)
        N.check(self._lib.w2v2_train_forward(self._handle, N.ptr(batch), B, L, N.ptr(attention_mask), N.ptr(sm), N.ptr(sd),
                                             float(p), C.c_uint64(seed), N.ptr(out), N.current_stream()), "w2v2_train_forward")
        return out, dict(spec_mask=spec_mask, sd_keep=sd_keep, seed=seed)

    def _forward(self, batch, attention_mask, training, out_channels):
        # the tuple call convention of the reference's fixed-length export wrappers (export2hub.py:40-57):
        # model((speech, attention_mask))
        if isinstance(batch, tuple) and len(batch) == 2 and attention_mask is None:
            batch, attention_mask = batch
        # same (non-fatal) warnings as reference modeling.py:183-186
        if self.config.is_robust and attention_mask is None:
            logger.warning("You should pass `attention_mask` when working with Wav2Vec2 new checkpoints")
        elif not self.config.is_robust and attention_mask is not None:
            logger.warning("You should not pass `attention_mask` when working with checkpoints based on `wav2vec2-base`")
        if training:
            out, self.last_training_call = self._train_forward(batch, attention_mask)
            return DeviceTensor.wrap(out)
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

    def profile(self, enable=True, families=None, stride=1):
        """Bracket kernel launches with HIP events; `families` (names as in profile_read) limits the
        instrumentation to those kernel families, None = all; `stride` > 1 samples every stride-th launch of a family."""
        N.check(self._lib.w2v2_profile_sampling(self._handle, int(stride)), "w2v2_profile_sampling")
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
        """{family: dict(launches, ms, flops, bytes, issued, kernels)}: event-bracketed launches with their time and algorithmic
        work; `issued` = op-level calls since the last reset (sampled or not); `kernels` = kernel launches actually enqueued for
        the family since the last reset (process-wide; what a rocprofv3 kernel trace counts)."""
        out = {}
        for i in range(self._lib.w2v2_profile_num_families()):
            name = C.c_char_p()
            n = C.c_int64()
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            N.check(self._lib.w2v2_profile_read(self._handle, i, C.byref(name), C.byref(n), C.byref(ms),
                                                C.byref(fl), C.byref(by)), "w2v2_profile_read")
            seen = C.c_int64()
            N.check(self._lib.w2v2_profile_seen(self._handle, i, C.byref(seen)), "w2v2_profile_seen")
            kern = C.c_int64()
            N.check(self._lib.w2v2_profile_kernel_launches(self._handle, i, C.byref(kern)), "w2v2_profile_kernel_launches")
            out[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value, issued=seen.value,
                                            kernels=kern.value)
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
