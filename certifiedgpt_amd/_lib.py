"""ctypes binding of libcgpt.so (include/cgpt.h).  This is the stub a certifiedGPT maintainer would add
(INTEGRATION.md); there is no CPU fallback: a missing library is an ImportError, a missing GPU is a
CGPT_ERR_NO_DEVICE from cgpt_create."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGPT_LIB_PATH") or os.path.join(_HERE, "libcgpt.so")   # CGPT_LIB_PATH: another build of the same ABI (A/B measurements)

CGPT_OK = 0
ERR_NAMES = {1: "CGPT_ERR_INVALID", 2: "CGPT_ERR_NO_DEVICE", 3: "CGPT_ERR_HIP", 4: "CGPT_ERR_NOT_FOUND", 5: "CGPT_ERR_STATE"}
MODE_VIT_HEAD = 0
MODE_ENCODE_IMG = 1
ABSTAIN = -1


class CgptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Config(C.Structure):
    """struct cgpt_config (include/cgpt.h) -- field order is ABI."""
    _fields_ = [
        ("struct_size", C.c_int32), ("mode", C.c_int32), ("device", C.c_int32), ("num_classes", C.c_int32),
        ("max_batch", C.c_int32),
        ("img_size", C.c_int32), ("patch_size", C.c_int32), ("vit_dim", C.c_int32), ("vit_depth", C.c_int32),
        ("vit_heads", C.c_int32), ("vit_mlp", C.c_int32), ("vit_ln_eps", C.c_float), ("ln_vision_eps", C.c_float),
        ("qf_layers", C.c_int32), ("qf_dim", C.c_int32), ("qf_heads", C.c_int32), ("qf_ffn", C.c_int32),
        ("qf_queries", C.c_int32), ("qf_xattn_freq", C.c_int32), ("qf_ln_eps", C.c_float), ("proj_dim", C.c_int32),
    ]


# (name, restype, argtypes) of every symbol include/cgpt.h declares
_P, _I64, _I32, _F, _D, _U64 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_uint64
SIGNATURES = {
    "cgpt_create": (_I32, [C.POINTER(Config), C.POINTER(_P)]),
    "cgpt_destroy": (_I32, [_P]),
    "cgpt_last_error": (C.c_char_p, []),
    "cgpt_version": (C.c_char_p, []),
    "cgpt_load_weight": (_I32, [_P, C.c_char_p, _P, _I64]),
    "cgpt_get_weight": (_I32, [_P, C.c_char_p, _P, _I64]),
    "cgpt_weight_numel": (_I64, [_P, C.c_char_p]),
    "cgpt_weight_name": (C.c_char_p, [_P, _I32]),
    "cgpt_init_synthetic_weights": (_I32, [_P, _U64, _P]),
    "cgpt_sample_counts": (_I32, [_P, _P, _I64, _I64, _I64, _F, _U64, _P, _P]),
    "cgpt_sample_counts2": (_I32, [_P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _F, _U64, _P]),
    "cgpt_sample_counts_images": (_I32, [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _F, _U64, _P]),
    "cgpt_forward_logits": (_I32, [_P, _P, _I64, _I64, _F, _U64, _P, _P]),
    "cgpt_classify": (_I32, [_P, _P, _I64, _P, _P]),
    "cgpt_encode_img": (_I32, [_P, _P, _I64, _P, _P]),
    "cgpt_encode_img_noisy": (_I32, [_P, _P, _I64, _I64, _F, _U64, _P, _P]),
    "cgpt_get_activation": (_I32, [_P, C.c_char_p, _P, _I64, _P]),
    "cgpt_noise_batch": (_I32, [_P, _I64, _I64, _I64, _F, _U64, _P, _P]),
    "cgpt_rgf_step": (_I32, [_P, _P, _I64, _I64, _I32, _P, _F, _F, _U64, _P, _P]),
    "cgpt_vote": (_I32, [_P, _I64, _I32, _P, _P]),
    "cgpt_allreduce_counts": (_I32, [_P, _P, _I64, _P]),
    "cgpt_allreduce_counts_fn": (_I32, [_P, _P, _P, _I64, _P]),
    "cgpt_certify_from_counts": (_I32, [_P, _P, _I32, _I64, _D, _D, C.POINTER(_I32), C.POINTER(_D)]),
    "cgpt_certify_many_from_counts": (_I32, [_P, _I64, _I32, _I64, _D, _D, _P, _P]),
    "cgpt_predict_from_counts": (_I32, [_P, _I32, _D, C.POINTER(_I32)]),
    "cgpt_certify_device": (_I32, [_P, _P, _I32, _I64, _D, _D, _P, _P]),
    "cgpt_predict_device": (_I32, [_P, _I32, _D, _P, _P]),
    "cgpt_lower_confidence_bound": (_D, [_I64, _I64, _D]),
    "cgpt_binom_test": (_D, [_I64, _I64, _D]),
    "cgpt_norm_ppf": (_D, [_D]),
    "cgpt_set_option": (_I32, [C.c_char_p, _I32]),
    "cgpt_mfma_sustained": (_I32, [_D, C.POINTER(_D), C.POINTER(_D)]),
    "cgpt_profile_enable": (_I32, [_P, _I32]),
    "cgpt_profile_read": (_I32, [_P, _I32, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I64)]),
    "cgpt_profile_clock": (_I32, [_P, _I32, C.POINTER(_D)]),
    "cgpt_profile_batches": (_I32, [_P, C.POINTER(_I32), _I64, C.POINTER(_I64)]),
    "cgpt_gemm_f16": (_I32, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "cgpt_linear_f16": (_I32, [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _I64, _I64, _I64, _I32, _P]),
    "cgpt_attention_f16": (_I32, [_P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "cgpt_layernorm": (_I32, [_P, _I64, _P, _P, _F, _P, _I64, _P, _I64, _I64, _I32, _P]),
}

_lib = None


def lib():
    """Load libcgpt.so once.  Raises ImportError (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C certifiedgpt_amd/csrc). "
                "certifiedgpt_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != CGPT_OK:
        raise CgptError(status, lib().cgpt_last_error().decode())
