"""ctypes binding of libmdil_hip.so (include/mdil_hip.h).  The product path has NO fallback:
if the library is missing or an entry point fails, a RuntimeError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDIL_HIP_LIB", os.path.join(_HERE, "libmdil_hip.so"))  # override: tuning builds
MAX_TAPS = 9


class Geom(C.Structure):
    _fields_ = [("N", C.c_int), ("HO", C.c_int), ("WO", C.c_int), ("HI", C.c_int), ("WI", C.c_int),
                ("ihs", C.c_int), ("iws", C.c_int), ("ntaps", C.c_int),
                ("dh", C.c_int * MAX_TAPS), ("dw", C.c_int * MAX_TAPS), ("src", C.c_int * MAX_TAPS),
                ("in_pitch", C.c_int * 2), ("OH", C.c_int), ("OW", C.c_int),
                ("ohs", C.c_int), ("oho", C.c_int), ("ows", C.c_int), ("owo", C.c_int),
                ("out_pitch", C.c_int), ("out_coff", C.c_int)]


class PackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ntaps", C.c_int), ("M", C.c_int),
                ("K", C.c_int), ("M_P", C.c_int), ("K_P", C.c_int), ("s_m", C.c_int),
                ("s_k", C.c_int), ("stem", C.c_int), ("ktap", C.c_int * MAX_TAPS)]


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("res", C.c_void_p), ("res_gate", C.c_void_p), ("gate", C.c_void_p),
                ("relu", C.c_int), ("bias2", C.c_void_p)]


class NbHalf(C.Structure):                    # == mdil_nb_half
    _fields_ = [(n, C.c_void_p) for n in (
        "wp31", "wp13", "b31", "b13", "pb", "gamma", "beta", "running_mean", "running_var",
        "num_batches_tracked", "coef", "dw31", "db31", "dw13", "db13", "dpw", "dpb", "dgamma",
        "dbeta")]


class BnGrad(C.Structure):                    # == mdil_bn_grad
    _fields_ = [("gamma", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("accumulate", C.c_int), ("coef", C.c_void_p), ("ticket", C.c_void_p)]


class BnTrain(C.Structure):                   # == mdil_bn_train
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
                ("eps", C.c_float), ("momentum", C.c_float), ("coef", C.c_void_p)]


class BnTail(C.Structure):                    # == mdil_bn_tail
    _fields_ = [(n, C.c_void_p) for n in ("gate", "z", "save_mean", "save_invstd", "drop", "partial")] + \
               [("fin", BnGrad)]


class NbBlock(C.Structure):                   # == mdil_nb_block
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
                ("dilation", C.c_int), ("rap", C.c_int), ("train", C.c_int),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("half", NbHalf * 2)] + \
               [(n, C.c_void_p) for n in ("x", "drop", "a1", "z1", "u", "a2", "z2", "out", "gy",
                                          "gz2", "ga", "gu", "gx")] + \
               [("bn_workspace", C.c_void_p), ("bn_workspace_bytes", C.c_size_t),
                ("wgrad_workspace", C.c_void_p), ("wgrad_workspace_bytes", C.c_size_t),
                ("head_partial", C.c_void_p), ("head_nblk", C.c_int), ("tail", BnTail),
                ("head_coef", C.c_void_p), ("ticket", C.c_void_p), ("eval_coef_ready", C.c_int)]


class WgradJob(C.Structure):                  # == mdil_wgrad_job (opaque record of a pending reduction)
    _fields_ = [("opaque", C.c_ubyte * 192)]


class ProfileRecord(C.Structure):             # == mdil_profile_record
    _fields_ = [("kind", C.c_int), ("path", C.c_int), ("cin", C.c_int), ("cout", C.c_int),
                ("ntaps", C.c_int), ("npix", C.c_longlong), ("ms", C.c_float)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_longlong
_F = C.c_float
_D = C.c_double
_Z = C.c_size_t

_SIGNATURES = {
    "mdil_last_error": (C.c_char_p, []),
    "mdil_version": (_I, []),
    "mdil_pack_weights": (_I, [_P, _P, _I, C.POINTER(_I), _I, _I, _I, _I, _I, _I, _I, _P]),
    "mdil_pack_weights_batch": (_I, [_P, _I, _P]),
    "mdil_tapconv": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(Epilogue), _P, _P]),
    "mdil_tapconv_stat_blocks": (_I, [C.POINTER(Geom), _I, _I]),
    "mdil_tapconv_stats": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(Epilogue), _P, _P, _P, _P]),
    "mdil_tapconv_bnred": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(Epilogue), _P, _P, _P, _P,
                                _P, C.POINTER(BnGrad), _P]),
    "mdil_tapconv_bn_train": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(Epilogue), _P,
                                   C.POINTER(BnTrain), _P, _Z, _P, _P]),
    "mdil_bn_backward_apply": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "mdil_tapconv_tail_blocks": (_I, [C.POINTER(Geom), _I, _I]),
    "mdil_tapconv_tail": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(Epilogue), _P,
                               C.POINTER(BnTail), _P]),
    "mdil_nb_block_tail_blocks": (_I, [_I, _I, _I, _I, _I]),
    "mdil_bn_backward_partials": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _Z, _P]),
    "mdil_bn_train_finalize": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mdil_wgrad_workspace": (_Z, [C.POINTER(Geom), _I, _I]),
    "mdil_wgrad": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(_I), _I, _I, _P, _P,
                   _I, _I, _I, _P, _P, _I, _P, _Z, _P]),
    "mdil_bn_workspace": (_Z, [_L, _I]),
    "mdil_bn_train_stats": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "mdil_bn_eval_coeffs": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "mdil_bn_apply": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "mdil_bn_backward": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _Z, _P, _P]),
    "mdil_maxpool_concat_fwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "mdil_maxpool_concat_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mdil_outconv_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mdil_nb_block_wgrad_workspace": (_Z, [_I, _I, _I, _I, _I, _I]),
    "mdil_nb_block_forward": (_I, [C.POINTER(NbBlock), _P]),
    "mdil_nb_block_backward": (_I, [C.POINTER(NbBlock), _P]),
    "mdil_nb_block_backward_deferred": (_I, [C.POINTER(NbBlock), C.POINTER(WgradJob), C.POINTER(_I),
                                             C.POINTER(_Z), _P]),
    "mdil_wgrad_deferred": (_I, [C.POINTER(Geom), _I, _I, _P, _P, _P, C.POINTER(_I), _I, _I, _P, _P,
                            _I, _I, _I, _P, _P, _P, _Z, C.POINTER(WgradJob), _P]),
    "mdil_wgrad_reduce_batch": (_I, [C.POINTER(WgradJob), _I, _P]),
    "mdil_head_workspace": (_Z, []),
    "mdil_head_ce": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "mdil_head_kld": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "mdil_loss_workspace": (_Z, [_L]),
    "mdil_ce_loss": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "mdil_kld_loss": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "mdil_argmax_confusion": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _P]),
    "mdil_adam_step": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _D, _D, _D, _D, _P]),
    "mdil_augment_batch": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mdil_nchw_to_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mdil_dropout_factors": (_I, [_P, _P, _P, _P, _I, _P]),
    "mdil_profile_begin": (_I, [_I]),
    "mdil_profile_end": (_I, [C.POINTER(ProfileRecord), _I]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError when the extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libmdil_hip.so not found at {LIB_PATH}: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU / eager fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mdil_last_error().decode()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
