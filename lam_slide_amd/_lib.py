"""ctypes binding of liblamslide_hip.so (include/lsl_api.h).

The HIP library is the only compute path of this package: importing works without it (so CPU-only
tooling can read shapes and pack weights), but every call that would run the network raises
``RuntimeError`` if the library is missing or the tensors are not on an AMD GPU.  There is no CPU
fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblamslide_hip.so")
SRC_DIR = os.path.join(_HERE, "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

ABI_VERSION = 6  # LSL_VERSION of include/lsl_api.h this binding was written against
RK_SCRATCH_BYTES = 8192  # LSL_RK_SCRATCH_BYTES

EXPORTED = (
    "lsl_version", "lsl_build_info", "lsl_last_error", "lsl_model_create", "lsl_model_set_weights", "lsl_model_destroy",
    "lsl_model_set_chunk", "lsl_model_set_attention_mode", "lsl_model_set_tail", "lsl_model_tail", "lsl_model_set_ln_fuse", "lsl_model_ln_fuse", "lsl_profile_kernel_name", "lsl_pass_size", "lsl_sampler_path", "lsl_workspace_bytes", "lsl_forward", "lsl_sample", "lsl_sample_ex", "lsl_debug_block", "lsl_debug_taps", "lsl_debug_mods",
    "lsl_profile_enable", "lsl_profile_read", "lsl_randn", "lsl_rk_lincomb", "lsl_rk_dense", "lsl_rk_error_ratio",
    "lsl_decoder_create", "lsl_decoder_destroy", "lsl_decode_workspace_bytes", "lsl_decode",
    "lsl_encoder_create", "lsl_encoder_destroy", "lsl_encode_workspace_bytes", "lsl_encode",
)


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_dim", "hidden", "heads", "head_dim", "head_dim_pad", "mlp_dim", "depth",
                                         "vec_in_dim", "normalize")] + [("theta", C.c_float)]


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "qs", "ks", "w2", "b2")]


class Weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "x_in_w", "x_in_b", "cond_w", "cond_b", "mask_emb", "time_freqs", "time_w1", "time_b1", "time_w2", "time_b2",
        "vec_w1", "vec_b1", "vec_w2", "vec_b2", "mod_w", "mod_b", "out_w", "out_b")] + [("blocks", C.POINTER(BlockWeights))]


class IO(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_cond", C.c_void_p), ("mask", C.c_void_p), ("y", C.c_void_p), ("t", C.c_void_p),
                ("out", C.c_void_p), ("B", C.c_int32), ("T", C.c_int32), ("L", C.c_int32)]


class DecBlock(C.Structure):  # lsl_dec_block
    _fields_ = [(n, C.c_void_p) for n in ("ln_w", "ln_b", "lnc_w", "lnc_b", "w_q", "w_kv", "w_out", "b_out", "q_scale", "k_scale",
                                          "ff_ln_w", "ff_ln_b", "ff_w1", "ff_b1", "ff_w2", "ff_b2")]


class DecoderDesc(C.Structure):  # lsl_decoder_desc
    _fields_ = [(n, C.c_int32) for n in ("in_dim", "dim_latent", "dim_query", "dim_emb", "n_entities", "heads_latent", "dim_head_latent",
                                         "heads_cross", "dim_head_cross", "num_block_attn", "num_block_cross", "act", "out_dim", "num_split")]


class DecoderWeights(C.Structure):  # lsl_decoder_weights
    _fields_ = [("pq_w", C.c_void_p), ("pq_b", C.c_void_p), ("table", C.c_void_p), ("qm_w", C.c_void_p), ("qm_b", C.c_void_p),
                ("self_blocks", C.POINTER(DecBlock)), ("cross_blocks", C.POINTER(DecBlock)), ("out_block", DecBlock),
                ("ext_w", C.c_void_p), ("ext_b", C.c_void_p), ("head_w1", C.c_void_p), ("head_b1", C.c_void_p), ("head_w2", C.c_void_p), ("head_b2", C.c_void_p)]


class EncoderDesc(C.Structure):  # lsl_encoder_desc
    _fields_ = [(n, C.c_int32) for n in ("dim_input", "dim_emb", "n_entities", "dim_latent", "num_latents", "heads_cross", "dim_head_cross",
                                         "heads_latent", "dim_head_latent", "num_block_cross", "num_block_attn", "act")]


class EncoderWeights(C.Structure):  # lsl_encoder_weights
    _fields_ = [("table", C.c_void_p), ("mlp_w1", C.c_void_p), ("mlp_b1", C.c_void_p), ("mlp_w2", C.c_void_p), ("mlp_b2", C.c_void_p),
                ("latents", C.c_void_p), ("cross_blocks", C.POINTER(DecBlock)), ("self_blocks", C.POINTER(DecBlock)),
                ("quant_w", C.c_void_p), ("quant_b", C.c_void_p)]


class Step(C.Structure):
    _fields_ = [("t", C.c_float), ("ax", C.c_float), ("am", C.c_float), ("aw", C.c_float)]


class StepEx(C.Structure):  # lsl_step_ex
    _fields_ = [("t", C.c_float), ("ax", C.c_float), ("am", C.c_float), ("aw", C.c_float), ("as_", C.c_float), ("flags", C.c_int32),
                ("noise_index", C.c_int32), ("trace_index", C.c_int32)]


STEP_NO_NETWORK, STEP_SAVE = 1, 2


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into the in-tree shared library (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(SRC_DIR, f) for f in sorted(os.listdir(SRC_DIR))] + [os.path.join(INCLUDE_DIR, "lsl_api.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value",
           "-Wl,--version-script=" + os.path.join(SRC_DIR, "exports.map"), "-o", LIB_PATH, os.path.join(SRC_DIR, "lsl_api.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed building liblamslide_hip.so")
    return LIB_PATH


class LibraryMissing(RuntimeError):
    """liblamslide_hip.so has not been built (the only failure a caller may treat as "no library": anything else - a HIP error, an
    out-of-memory condition - propagates)."""


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library or fail loudly: the HIP path is the product, there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is the only compute path of lam_slide_amd)")
    lib = C.CDLL(LIB_PATH)
    lib.lsl_version.restype = C.c_int
    lib.lsl_build_info.restype = C.c_char_p
    lib.lsl_last_error.restype = C.c_char_p
    lib.lsl_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]
    lib.lsl_model_set_weights.argtypes = [C.c_void_p, C.POINTER(Weights)]
    lib.lsl_model_destroy.argtypes = [C.c_void_p]
    lib.lsl_model_destroy.restype = None
    lib.lsl_model_set_chunk.argtypes = [C.c_void_p, C.c_int32]
    lib.lsl_model_set_attention_mode.argtypes = [C.c_void_p, C.c_int32]
    lib.lsl_model_set_tail.argtypes = [C.c_void_p, C.c_int32]
    lib.lsl_model_tail.argtypes = [C.c_void_p]
    lib.lsl_model_set_ln_fuse.argtypes = [C.c_void_p, C.c_int32]
    lib.lsl_model_ln_fuse.argtypes = [C.c_void_p]
    lib.lsl_profile_kernel_name.argtypes = [C.c_void_p]
    lib.lsl_profile_kernel_name.restype = C.c_char_p
    lib.lsl_pass_size.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.lsl_sampler_path.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.lsl_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.lsl_workspace_bytes.restype = C.c_size_t
    lib.lsl_forward.argtypes = [C.c_void_p, C.POINTER(IO), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lsl_sample.argtypes = [C.c_void_p, C.POINTER(IO), C.POINTER(Step), C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64,
                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lsl_sample_ex.argtypes = [C.c_void_p, C.POINTER(IO), C.POINTER(StepEx), C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64,
                                  C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lsl_debug_block.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lsl_debug_taps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lsl_debug_mods.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]
    lib.lsl_randn.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.lsl_rk_lincomb.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int32, C.c_uint64, C.c_void_p]
    lib.lsl_rk_dense.argtypes = [C.c_void_p] * 6 + [C.c_float, C.c_uint64, C.c_void_p]
    lib.lsl_rk_error_ratio.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int32, C.c_float, C.c_float,
                                       C.c_uint64, C.c_void_p, C.c_void_p]
    lib.lsl_profile_enable.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.lsl_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.lsl_decoder_create.argtypes = [C.POINTER(DecoderDesc), C.POINTER(DecoderWeights), C.POINTER(C.c_void_p)]
    lib.lsl_decoder_destroy.argtypes = [C.c_void_p]
    lib.lsl_decoder_destroy.restype = None
    lib.lsl_decode_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.lsl_decode_workspace_bytes.restype = C.c_size_t
    lib.lsl_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_void_p]
    lib.lsl_encoder_create.argtypes = [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.POINTER(C.c_void_p)]
    lib.lsl_encoder_destroy.argtypes = [C.c_void_p]
    lib.lsl_encoder_destroy.restype = None
    lib.lsl_encode_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.lsl_encode_workspace_bytes.restype = C.c_size_t
    lib.lsl_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_void_p]
    if lib.lsl_version() != ABI_VERSION:
        raise RuntimeError(f"liblamslide_hip.so reports ABI version {lib.lsl_version()}, this package binds version {ABI_VERSION}: rebuild it "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def last_error() -> str:
    return load().lsl_last_error().decode()


def check(rc: int, shape_error=RuntimeError):
    if rc == 0:
        return
    msg = last_error()
    if rc in (-20, -21):
        raise ValueError(msg)
    if rc == -3:
        raise ValueError(msg)
    raise RuntimeError(f"lamslide_hip error {rc}: {msg}")
