"""ctypes binding of liborca_hip.so (include/orca_hip.h).

The library is built in-tree (``make -C orca_amd/csrc`` or
``__graft_entry__.build()``).  There is NO fallback: if the shared object is
missing, or an entry point fails, an exception is raised.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liborca_hip.so")

ORCA_NET_ENCODER, ORCA_NET_ENCODER2, ORCA_NET_ENCODER3, ORCA_NET_DECODER, ORCA_NET_DECODER_1M, ORCA_NET_ENCODER2B = 1, 2, 3, 4, 5, 6
ORCA_UPSAMPLE_NEAREST, ORCA_UPSAMPLE_BILINEAR = 0, 1
PRECISIONS = {"f32": 0, "bf16": 1, "bf16x2": 2, "bf16x3": 3, "f16x2": 4, "f16": 5}
ENCODER_FORMS = {"default": 0, "stored_residual": 1, "lconv1_only": 2, "two_conv": 3}      # include/orca_hip.h: ORCA_ENCODER_FORM_*


class OrcaHipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [("weight_host", c_void_p), ("bias_host", c_void_p), ("cout", c_int32), ("cin", c_int32),
                ("ksize", c_int32), ("dilation", c_int32)]


class KernelTime(ctypes.Structure):
    _fields_ = [("cout", c_int32), ("cin", c_int32), ("tile", c_int32), ("batch", c_int32), ("n", c_int64),
                ("ms", ctypes.c_float), ("ksize", c_int32)]


# name -> (restype, argtypes); every symbol declared in include/orca_hip.h
SIGNATURES = {
    "orca_abi_version": (c_int, []),
    "orca_last_error": (c_char_p, []),
    "orca_device_count": (c_int, []),
    "orca_ctx_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "orca_ctx_destroy": (c_int, [c_void_p]),
    "orca_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "orca_ctx_workspace_bytes": (c_int, [c_void_p, POINTER(c_size_t)]),
    "orca_ctx_release_workspace": (c_int, [c_void_p]),
    "orca_ctx_take_overflow": (c_int, [c_void_p, POINTER(c_int)]),
    "orca_ctx_set_timing": (c_int, [c_void_p, c_int]),
    "orca_ctx_get_timing": (c_int, [c_void_p, POINTER(KernelTime), c_int, POINTER(c_int)]),
    "orca_ctx_launch_counts": (c_int, [c_void_p, POINTER(ctypes.c_int64)]),
    "orca_net_create": (c_int, [c_void_p, c_int, POINTER(ConvDesc), c_int, c_int, POINTER(c_void_p)]),
    "orca_net_free": (c_int, [c_void_p]),
    "orca_net_set_precision": (c_int, [c_void_p, c_int]),
    "orca_net_set_encoder_form": (c_int, [c_void_p, c_int]),
    "orca_encoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int64, c_int64,
                                     c_int64, c_void_p, c_int64, c_int64, c_int64]),
    "orca_p16_plane_units": (c_int64, [c_int64]),
    "orca_encoder_stage3_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64]),
    "orca_p16_pool5_into": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64]),
    "orca_encoder_front_snippet": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64]),
    "orca_encoder_back": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64]),
    "orca_encoder_stage4_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "orca_rows_pool5_into": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64]),
    "orca_encoder_front4_snippet": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64]),
    "orca_encoder_front4_ranges": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, POINTER(ctypes.c_int64), c_void_p, c_int64]),
    "orca_encoder_back5": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64]),
    "orca_encoder_forward_2bit": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64]),
    "orca_pack_sequence": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, POINTER(c_int)]),
    "orca_encoder_forward_codes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int64,
                                           c_void_p, c_int64, c_int64, c_int64]),
    "orca_encoder_forward_codes_window": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int64, c_int64, c_int64,
                                                  c_void_p, c_int64, c_int64, c_int64]),
    "orca_encoder_num_bins": (c_int64, [c_int64]),
    "orca_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int,
                                  POINTER(c_void_p), c_int]),
    "orca_decoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                     c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int]),
    "orca_decoder_forward_mt": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                        c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int]),
    "orca_decoder_forward_rows": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int64, c_int64, POINTER(c_void_p), c_int64, c_int64, c_int64,
                                          POINTER(c_void_p), c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int]),
    "orca_decoder1m_forward_rows": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int64, c_int64, c_int, c_int, c_void_p, c_int]),
    "orca_net_num_targets": (c_int, [c_void_p, POINTER(c_int)]),
    "orca_decoder1m_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int]),
    "orca_strand_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "orca_block_mean_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_int]),
    "orca_adaptive_coarsegrain": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, ctypes.c_float, c_int, c_int, c_void_p, c_int64]),
    "orca_genome_unpack_2bit": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "orca_comm_unique_id": (c_int, [c_void_p]),
    "orca_comm_init_rank": (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "orca_comm_destroy": (c_int, [c_void_p]),
    "orca_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "orca_conv1d_forward": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                    c_void_p, c_void_p, c_int, c_int64, c_int, c_int]),
    "orca_conv1d_nlc_forward": (c_int, [c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int]),
    "orca_conv1d_p16_forward": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int]),
    "orca_conv1d_b16_forward": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int]),
    "orca_conv2d_forward": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "orca_conv2d_m16_forward": (c_int, [c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "orca_maxpool1d_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int]),
    "orca_pointwise1d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                          c_int64, c_int, c_int64, c_int]),
}

_lib = None


def load():
    """Load liborca_hip.so (once) and set the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrcaHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C orca_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be
    # in the process BEFORE liborca_hip.so so that our NEEDED libamdhip64.so.7 binds to the same runtime
    # instance that owns torch's tensors and streams (two runtimes in one process do not share devices).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.orca_abi_version() != 1:
        raise OrcaHipError("liborca_hip ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().orca_last_error()
        raise OrcaHipError(f"{what} failed (code {rc}): {msg.decode('utf8', 'replace') if msg else ''}")
