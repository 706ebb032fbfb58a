"""ctypes binding of libvcx.so (include/vcx.h).

The library is the only compute backend of this package: there is no CPU or PyTorch fallback.
`lib()` raises if the shared object is missing, and every op wrapper raises on a non-zero
return code with the library's own error text.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvcx.so")

# every symbol include/vcx.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "vcx_abi_version", "vcx_last_error", "vcx_device_arch", "vcx_gemm_f16", "vcx_gemm_units_f16",
    "vcx_groupnorm_ws_bytes", "vcx_groupnorm_stats_f16", "vcx_groupnorm_apply_f16", "vcx_groupnorm_apply2_f16", "vcx_groupnorm_stats_from_colstats_f32", "vcx_groupnorm_fold_linear_f16", "vcx_layernorm_f16", "vcx_rowstats_f16",
    "vcx_attn_flash_d64_f16", "vcx_attn_flash_d512_f16", "vcx_attn_flash_dual_d64_f16", "vcx_attn_temporal_d64_f16", "vcx_attn_temporal_d64_masked_f16", "vcx_attn_temporal_d64_rel_f16", "vcx_softmax_rows_f16",
    "vcx_silu_f32", "vcx_gelu_f16", "vcx_clip_preprocess_f32", "vcx_add_nchw_f32_to_nhwc_f16", "vcx_timestep_embedding_f32", "vcx_cast_f32_to_f16", "vcx_cast_f16_to_f32",
    "vcx_copy2d_f16", "vcx_avgpool2x2_f16", "vcx_upsample2x_f16", "vcx_ncthw_f32_to_nthwc_f16", "vcx_nthwc_to_ncthw_f32", "vcx_ddim_ws_bytes", "vcx_ddim_step_f32", "vcx_ddim_step3_f32",
    "vcx_profile_begin", "vcx_profile_end", "vcx_tune_set", "vcx_tune_get",
]

ABI_VERSION = 9          # include/vcx.h VCX_ABI_VERSION
# experiment knobs (include/vcx.h VCX_TUNE_*): name -> (index, default)
TUNE = {"GEMM_CFG": (0, -1), "GEMM_DMA": (1, 1), "FLASH_QB": (2, 0), "XATTN_RESIDENT": (3, 1), "FLASH_IMPL": (4, 0), "EXP0": (5, 0),
        "EXP1": (6, 0), "GEMM_WS": (7, 1)}
GEMM_BIAS_N, GEMM_BIAS_M, GEMM_ROWADD, GEMM_RESIDUAL, GEMM_GEGLU, GEMM_OUT_F32, GEMM_CONV_SLABK = 1, 2, 4, 8, 16, 32, 64
GEMM_LNFOLD, GEMM_LNFOLD_T, GEMM_COLSTATS, GEMM_ROWSTATS = 0x80, 0x100, 0x200, 0x400
PROF_FAMILIES = ("gemm", "flash_attn", "temporal_attn", "groupnorm", "layernorm", "elementwise")


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_size_t),
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("rowadd", c_void_p),
        ("residual", c_void_p), ("lda", c_int64), ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("ldw", c_int32), ("ldc", c_int32), ("ldr", c_int32), ("mode", c_int32),
        ("in_h", c_int32), ("in_w", c_int32), ("out_h", c_int32), ("out_w", c_int32), ("cin", c_int32),
        ("kh", c_int32), ("kw", c_int32), ("stride", c_int32), ("pad_h", c_int32), ("pad_w", c_int32),
        ("ups", c_int32), ("rowadd_div", c_int32), ("flags", c_int32), ("alpha", c_float),
        ("ln_stats", c_void_p), ("ln_colsum", c_void_p), ("colstats", c_void_p), ("ldcs", c_int64),
        ("rowstats", c_void_p), ("rowstats_eps", c_float), ("rowadd_ld", c_int32),
        ("tail_a0", c_void_p), ("tail_a1", c_void_p), ("tail_lda0", c_int64), ("tail_lda1", c_int64), ("tail_k0", c_int32), ("tail_k1", c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(GemmDesc)      # vcx_gemm_f16 rejects any other value (include/vcx.h, ABI 5+)


_lib = None


class VcxError(RuntimeError):
    pass


def lib():
    """Load libvcx.so once; fail loudly if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VcxError(
            f"{LIB_PATH} not found: build it with `make -C viewcrafter_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    # Load order matters: PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64.so.1.  libvcx.so must bind to the HIP
    # runtime the host framework uses (its streams and device pointers come from there), so torch is imported first and the
    # dynamic loader resolves libvcx's NEEDED libamdhip64.so.7 to the copy already in the process.  The other way round
    # (/opt/rocm's runtime first, torch's HSA layer second) leaves two half-initialised runtimes: every launch then fails with
    # "no ROCm-capable device is detected".
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.vcx_abi_version.restype = c_int
    L.vcx_last_error.restype = c_char_p
    L.vcx_device_arch.argtypes = [c_char_p, c_int]
    L.vcx_gemm_f16.argtypes = [POINTER(GemmDesc), c_void_p]
    L.vcx_gemm_units_f16.argtypes = [POINTER(GemmDesc), c_int, c_int64, c_int64, c_void_p]
    L.vcx_groupnorm_ws_bytes.argtypes = [c_int, c_int64, c_int]
    L.vcx_groupnorm_stats_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]
    L.vcx_groupnorm_apply_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                          c_int, c_float, c_int, c_void_p]
    L.vcx_groupnorm_apply2_f16.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                           c_int, c_float, c_int, c_void_p]
    L.vcx_layernorm_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]
    L.vcx_rowstats_f16.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]
    L.vcx_groupnorm_stats_from_colstats_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]
    L.vcx_groupnorm_fold_linear_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                c_float, c_void_p]
    L.vcx_attn_flash_d64_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_void_p]
    L.vcx_attn_flash_d512_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_float, c_void_p]
    L.vcx_attn_flash_dual_d64_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                              c_int, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int, c_int64, c_int64,
                                              c_int64, c_int64, c_float, c_int, c_void_p]
    L.vcx_attn_temporal_d64_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                            c_int64, c_float, c_void_p]
    L.vcx_attn_temporal_d64_masked_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                            c_int64, c_float, c_int, c_void_p]
    L.vcx_attn_temporal_d64_rel_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                                c_int64, c_float, c_int, c_void_p]
    L.vcx_softmax_rows_f16.argtypes = [c_void_p, c_int64, c_int, c_int64, c_void_p]
    L.vcx_silu_f32.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    L.vcx_gelu_f16.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    L.vcx_add_nchw_f32_to_nhwc_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]
    L.vcx_clip_preprocess_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float),
                                          c_void_p]
    L.vcx_timestep_embedding_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]
    L.vcx_cast_f32_to_f16.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    L.vcx_cast_f16_to_f32.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    L.vcx_copy2d_f16.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p]
    L.vcx_avgpool2x2_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.vcx_upsample2x_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.vcx_ncthw_f32_to_nthwc_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_float,
                                             c_void_p]
    L.vcx_nthwc_to_ncthw_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p]
    L.vcx_ddim_ws_bytes.restype = ctypes.c_size_t
    L.vcx_ddim_ws_bytes.argtypes = [c_int, c_int64]
    L.vcx_ddim_step_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_int,
                                    c_int64, POINTER(c_float), c_void_p]
    L.vcx_ddim_step3_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t,
                                     c_int, c_int64, POINTER(c_float), c_void_p]
    L.vcx_profile_begin.argtypes = [c_int]
    L.vcx_profile_end.argtypes = [POINTER(c_double)]
    L.vcx_tune_set.argtypes = [c_int, c_int]
    L.vcx_tune_get.argtypes = [c_int]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name == "vcx_groupnorm_ws_bytes":
            fn.restype = ctypes.c_size_t
        elif name != "vcx_last_error":
            fn.restype = c_int
    if L.vcx_abi_version() != ABI_VERSION:
        raise VcxError(f"libvcx ABI version {L.vcx_abi_version()} != {ABI_VERSION}; rebuild the library (make -C viewcrafter_amd/csrc)")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().vcx_last_error().decode("utf-8", "replace")
        raise VcxError(f"libvcx call failed ({rc}) {what}: {msg}")
