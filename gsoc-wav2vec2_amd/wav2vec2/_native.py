"""ctypes binding of the C ABI in include/w2v2.h (lib/libw2v2.so).

This is the whole Python<->HIP boundary: plain pointers and sizes, no torch
types.  torch-ROCm is only the carrier of device buffers (``data_ptr()``) and
of the current stream.  There is NO fallback: if the library cannot be loaded
the product path raises -- it never routes through the CPU oracle.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libw2v2.so")

MAX_CONV_LAYERS = 16


class W2V2Config(C.Structure):
    """struct w2v2_config (include/w2v2.h)."""
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("hidden_size", C.c_int32),
        ("num_heads", C.c_int32),
        ("num_layers", C.c_int32),
        ("intermediate_size", C.c_int32),
        ("num_conv_pos_embeddings", C.c_int32),
        ("num_conv_pos_embedding_groups", C.c_int32),
        ("num_conv_layers", C.c_int32),
        ("filter_sizes", C.c_int32 * MAX_CONV_LAYERS),
        ("kernal_sizes", C.c_int32 * MAX_CONV_LAYERS),
        ("strides", C.c_int32 * MAX_CONV_LAYERS),
        ("conv_bias", C.c_int32),
        ("feature_extractor_norm_type", C.c_int32),
        ("attention_norm_type", C.c_int32),
        ("is_gelu_approx", C.c_int32),
        ("with_lm_head", C.c_int32),
        ("pad_id", C.c_int32),
        ("layer_norm_eps", C.c_float),
    ]


# name -> (restype, argtypes); every symbol declared in include/w2v2.h
_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
PROTOTYPES = {
    "w2v2_last_error": (C.c_char_p, []),
    "w2v2_version": (C.c_char_p, []),
    "w2v2_release_scratch": (C.c_int, []),
    "w2v2_create": (C.c_int, [C.POINTER(W2V2Config), C.POINTER(_P)]),
    "w2v2_destroy": (None, [_P]),
    "w2v2_num_params": (C.c_int, [_P]),
    "w2v2_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(C.c_int)]),
    "w2v2_set_param": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(_I64), C.c_int]),
    "w2v2_get_param": (C.c_int, [_P, C.c_char_p, _P, _I64]),
    "w2v2_finalize": (C.c_int, [_P, _P]),
    "w2v2_num_frames": (_I64, [_P, _I64]),
    "w2v2_set_precision": (C.c_int, [_P, _I32]),
    "w2v2_get_precision": (C.c_int, [_P]),
    "w2v2_forward": (C.c_int, [_P, _P, _I32, _I64, _P, _P, _P]),
    "w2v2_ctc_loss": (C.c_int, [_P, _I32, _I32, _I32, _P, _I32, _P, _P, _I32, _P, _P, _P]),
    "w2v2_ctc_loss_fused": (C.c_int, [_P, _I32, _I32, _I32, _P, _I32, _I32, _I32, C.c_float, _P, _P, _P, _P]),
    "w2v2_set_trainable": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "w2v2_set_trainable_flags": (C.c_int, [_P, _P, _I32]),
    "w2v2_set_option": (C.c_int, [_P, _I32, _I32]),
    "w2v2_get_option": (C.c_int, [_P, _I32]),
    "w2v2_range_overflow": (C.c_int, [_P, C.POINTER(_I32), _P]),
    "w2v2_train_forward": (C.c_int, [_P, _P, _I32, _I64, _P, _P, _P, C.c_float, C.c_uint64, _P, _P]),
    "w2v2_train_backward": (C.c_int, [_P, _P, _P]),
    "w2v2_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "w2v2_adam_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "w2v2_adam_reset": (C.c_int, [_P, _P]),
    "w2v2_train_num_buckets": (C.c_int, [_P]),
    "w2v2_train_bucket": (C.c_int, [_P, _I32, C.POINTER(_I64), C.POINTER(_I64)]),
    "w2v2_train_bucket_wait": (C.c_int, [_P, _I32, _P]),
    "w2v2_grad_slot": (C.c_int, [_P, C.c_char_p, C.POINTER(_I64), C.POINTER(_I64)]),
    "w2v2_comm_unique_id": (C.c_int, [_P, _I32]),
    "w2v2_comm_init": (C.c_int, [_P, _P, _I32, _I32, _I32]),
    "w2v2_comm_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "w2v2_comm_destroy": (C.c_int, [_P]),
    "w2v2_allreduce_num_runs": (C.c_int, [_P, _I32, C.POINTER(_I32)]),
    "w2v2_allreduce_run": (C.c_int, [_P, _I32, _I32, C.POINTER(_I64), C.POINTER(_I64)]),
    "w2v2_allreduce_bucket": (C.c_int, [_P, _I32, _I32]),
    "w2v2_allreduce_finish": (C.c_int, [_P, _P, C.POINTER(_I64)]),
    "w2v2_get_grad": (C.c_int, [_P, C.c_char_p, _P, _I64, _P]),
    "w2v2_train_storage": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I64)]),
    "w2v2_adam_step": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, _I64, _P]),
    "w2v2_ln_bwd_ws_floats": (_I64, [_I64, _I32]),
    "w2v2_op_layer_norm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, C.c_float, _P, _P]),
    "w2v2_op_attention_train": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, C.c_float, C.c_uint64, C.c_uint32, _P]),
    "w2v2_op_attention_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, C.c_float, C.c_uint64, C.c_uint32, _P]),
    "w2v2_op_dropout": (C.c_int, [_P, _P, _P, _I64, _I32, C.c_float, C.c_uint64, C.c_uint32, _P]),
    "w2v2_op_layer_norm_dropout": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, C.c_float, C.c_float, C.c_uint64, C.c_uint32, _P]),
    "w2v2_activation_info": (C.c_int, [_P, C.c_char_p, C.POINTER(_I64)]),
    "w2v2_copy_activation": (C.c_int, [_P, C.c_char_p, _P, _I64, _P]),
    "w2v2_profile_enable": (C.c_int, [_P, C.c_int]),
    "w2v2_profile_families": (C.c_int, [_P, C.c_uint32]),
    "w2v2_profile_sampling": (C.c_int, [_P, C.c_int32]),
    "w2v2_profile_seen": (C.c_int, [_P, C.c_int, C.POINTER(_I64)]),
    "w2v2_profile_kernel_launches": (C.c_int, [_P, C.c_int, C.POINTER(_I64)]),
    "w2v2_profile_num_families": (C.c_int, []),
    "w2v2_clock_probe": (C.c_int, [_P, _I32, _P, C.POINTER(_I32)]),
    "w2v2_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_I64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "w2v2_profile_reset": (C.c_int, [_P]),
    "w2v2_op_gemm": (C.c_int, [_P, _I64, _I64, _P, _I64, _P, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_set_precision": (C.c_int, [_I32]),
    "w2v2_op_gemm_split": (C.c_int, [_P, _I64, _I64, _P, _P, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_gemm_bf16_at": (C.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _P]),
    "w2v2_crc32c_extend": (C.c_uint32, [C.c_uint32, _P, C.c_uint64]),
    "w2v2_op_weight_grad_bf16": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_check_select_forms": (C.c_int, [_P, _P]),
    "w2v2_op_split_planes": (C.c_int, [_P, _P, _I64, _I64, _I32, _P, _P]),
    "w2v2_op_split_weight": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "w2v2_op_gemm_split_planes": (C.c_int, [_I32, _P, _I64, _I64, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "w2v2_op_gemm_bf16_shadows": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_gemm_bf16": (C.c_int, [_P, _I64, _I64, _P, _I64, _P, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_layer_norm": (C.c_int, [_P, _P, _P, _P, _I64, _I32, C.c_float, _I32, _P]),
    "w2v2_conv0_ws_floats": (_I64, [_I32, _I64, _I32, _I32, _I32]),
    "w2v2_op_conv0": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _I32, C.c_float, _I32, _I32, _P]),
    "w2v2_op_weight_norm_regroup": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_pos_conv": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_attention": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "w2v2_op_frame_lengths": (C.c_int, [_P, _P, _I32, _I64, C.POINTER(_I32), C.POINTER(_I32), _I32, _P]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load(build_if_missing=True):
    """Load lib/libw2v2.so, declaring every prototype.  Raises
    NativeLibraryError (never falls back) when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    # torch-ROCm bundles its own HIP runtime (soname libamdhip64.so.7).  It must be the one already
    # mapped when libw2v2.so is loaded, so that both sides share ONE runtime (device pointers,
    # streams); loading libw2v2.so first would pull in /opt/rocm's copy as a second runtime.
    import torch  # noqa: F401
    override = os.environ.get("W2V2_NATIVE_LIB")      # tools/ only: the -DW2V2_TUNING build (build.py --tuning) for sweeps
    if override:
        path = override
    else:
        path = LIB_PATH
    if not os.path.exists(LIB_PATH) and build_if_missing and not override:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("w2v2_build", os.path.join(_PKG_ROOT, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise NativeLibraryError(
                f"libw2v2.so is missing at {LIB_PATH} and could not be built with hipcc: {e}") from e
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise NativeLibraryError(
            f"cannot load the HIP library {path}: {e}.  The MI355X path has no CPU fallback; "
            "build it with `python gsoc-wav2vec2_amd/build.py`.") from e
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    return (load().w2v2_last_error() or b"").decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"{what or 'w2v2'} failed (code {rc}): {last_error()}")


def make_config(config, with_lm_head):
    """Wav2Vec2Config dataclass -> struct w2v2_config."""
    n = len(config.filter_sizes)
    if n > MAX_CONV_LAYERS:
        raise ValueError(f"at most {MAX_CONV_LAYERS} conv layers are supported")
    c = W2V2Config()
    c.vocab_size = config.vocab_size
    c.hidden_size = config.hidden_size
    c.num_heads = config.num_heads
    c.num_layers = config.num_layers
    c.intermediate_size = config.intermediate_size
    c.num_conv_pos_embeddings = config.num_conv_pos_embeddings
    c.num_conv_pos_embedding_groups = config.num_conv_pos_embedding_groups
    c.num_conv_layers = n
    for i in range(n):
        c.filter_sizes[i] = config.filter_sizes[i]
        c.kernal_sizes[i] = config.kernal_sizes[i]
        c.strides[i] = config.strides[i]
    c.conv_bias = int(bool(config.conv_bias))
    if config.feature_extractor_norm_type not in ("group", "layer"):
        raise NotImplementedError(config.feature_extractor_norm_type)     # feature_extractor.py:51-52
    c.feature_extractor_norm_type = 0 if config.feature_extractor_norm_type == "group" else 1
    c.attention_norm_type = 0 if config.attention_norm_type == "postnorm" else 1
    c.is_gelu_approx = int(bool(config.is_gelu_approx))
    c.with_lm_head = int(bool(with_lm_head))
    c.pad_id = config.pad_id
    c.layer_norm_eps = config.layer_norm_eps
    return c


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array, or NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
