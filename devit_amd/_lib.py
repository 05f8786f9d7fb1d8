"""ctypes binding of libdevit_hip.so (C ABI declared in include/devit_hip.h).

The product path has NO fallback: if the shared library is missing, or the current device is
not an MI355X (gfx950), every op raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEVIT_LIB_PATH") or os.path.join(_HERE, "libdevit_hip.so")   # override: diagnostics only

# devit_epilogue_kind
EPI_STORE_BF16, EPI_GELU_BF16, EPI_RESIDUAL_F32, EPI_PATCH_F32, EPI_DGELU_BF16, EPI_ATOMIC_F32, EPI_STORE_F32 = range(7)


class Epilogue(C.Structure):
    _fields_ = [("kind", C.c_int), ("out", C.c_void_p), ("ldc", C.c_int), ("bias", C.c_void_p),
                ("colscale", C.c_void_p), ("aux", C.c_void_p), ("aux_in", C.c_void_p), ("res", C.c_void_p),
                ("rowscale", C.c_void_p), ("rows_per_scale", C.c_int), ("pos", C.c_void_p),
                ("patch_tokens", C.c_int), ("extra_tokens", C.c_int), ("exact_gelu", C.c_int),
                ("out_batch_stride", C.c_longlong), ("m_valid", C.c_int), ("dtype16", C.c_int)]


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int), ("kmajor", C.c_int), ("row_group", C.c_int),
                ("row_skip", C.c_int), ("batch_stride", C.c_longlong)]


ACT_LN1, ACT_MEAN1, ACT_RSTD1, ACT_QKV, ACT_ATTN_O, ACT_LSE, ACT_X1, ACT_ATT, ACT_LN2, ACT_MEAN2, ACT_RSTD2, ACT_H, ACT_H_PRE, \
    ACT_X2, ACT_COUNT = range(15)
BLK_SAVE, BLK_QKV_PAD, BLK_ATT = 1, 2, 4
BWD_DH_PRE, BWD_DLN2, BWD_DX1, BWD_G1, BWD_DATTN, BWD_DQKV, BWD_DLN1, BWD_LNWS, BWD_COUNT = range(9)


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("n1w", "n1b", "qkv_b", "proj_b", "n2w", "n2b", "fc1_b", "fc2_b", "qkv_w16",
                                          "proj_w16", "fc1_w16", "fc2_w16", "head_gate", "neuron_gate")] + \
               [("num_heads", C.c_int), ("attn_width", C.c_int), ("hidden", C.c_int), ("dtype16", C.c_int), ("fc2_w16t", C.c_void_p)]


class BlockWgrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("n1w", "n1b", "qkv_w", "qkv_b", "proj_w", "proj_b", "n2w", "n2b", "fc1_w",
                                          "fc1_b", "fc2_w", "fc2_b")]


class BlockActs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("buf", C.c_void_p * 14), ("dp1", C.c_void_p), ("dp2", C.c_void_p), ("flags", C.c_int)]


class BlockBwdIO(C.Structure):
    _fields_ = [("dx", C.c_void_p), ("g2", C.c_void_p), ("dx_in", C.c_void_p), ("g_prev", C.c_void_p),
                ("prev_dp2", C.c_void_p), ("prev_fc2_b_grad", C.c_void_p), ("g2_bias_done", C.c_int),
                ("dqkv_add", C.c_void_p), ("ws", C.c_void_p * 8), ("lnws_bytes", C.c_size_t),
                ("defer_jobs", C.c_void_p), ("defer_count", C.c_void_p)]


class IndexJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("idx", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int),
                ("src_ld", C.c_int), ("dst_ld", C.c_int), ("mode", C.c_int), ("elem", C.c_int)]


class WgradJob(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int), ("a_cols", C.c_int), ("b", C.c_void_p), ("ldb", C.c_int), ("out", C.c_void_p),
                ("ldc", C.c_int), ("transposed", C.c_int), ("a_colsum", C.c_void_p)]


WGRAD_MAX_JOBS = 48
ABI_VERSION = 2
# devit_abi_struct_size(which) -> the mirror it must equal (checked at load time: an array of stale mirrors is misread silently)
ABI_STRUCTS = {0: Epilogue, 1: Operand, 2: BlockWeights, 3: BlockWgrads, 4: BlockActs, 5: BlockBwdIO, 6: IndexJob, 7: WgradJob}


class DevitError(RuntimeError):
    pass


_P, _I, _F, _Z, _LL = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong

# name -> (restype, argtypes); must list every symbol of include/devit_hip.h (tests check this)
SIGNATURES = {
    "devit_version": (_I, []),
    "devit_last_error": (C.c_char_p, []),
    "devit_check_device": (_I, [_I]),
    "devit_abi_struct_size": (_Z, [_I]),
    "devit_gemm_full_row_selected": (_I, [_I, _I, _I, _I]),
    "devit_wgrad_grouped": (_I, [C.POINTER(WgradJob), _I, _I, _I, _P]),
    "devit_gemm_bf16": (_I, [C.POINTER(Operand), C.POINTER(Operand), _I, _I, _I, _I, _I, C.POINTER(Epilogue), _P]),
    "devit_set_reserved_cus": (_I, [_I]),
    "devit_get_reserved_cus": (_I, []),
    "devit_layernorm_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _I, _P]),
    "devit_layernorm_bwd_workspace": (_Z, [_I, _I]),
    "devit_layernorm_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _Z, _P]),
    "devit_block_acts_sizes": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(C.c_size_t)]),
    "devit_block_bwd_sizes": (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_size_t)]),
    "devit_encoder_fwd": (_I, [_I, C.POINTER(BlockWeights), C.POINTER(BlockActs), _I, _I, _I, _F, _P]),
    "devit_block_bwd": (_I, [C.POINTER(BlockWeights), C.POINTER(BlockActs), C.POINTER(BlockWgrads), C.POINTER(BlockBwdIO),
                             _I, _I, _I, _F, _P]),
    "devit_attn_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "devit_attn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "devit_attn_fwd_rows": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "devit_attn_bwd_rows": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "devit_index_copy": (_I, [_P, _I, _I, _P]),
    "devit_im2row_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "devit_mix_im2row_bf16": (_I, [_P, _P, _P, _I, _I, C.c_double, _I, _I, _I, _I, _P]),
    "devit_mix_targets": (_I, [_P, _P, _I, _I, C.c_double, C.c_double, _P]),
    "devit_embed_tokens": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "devit_embed_bwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "devit_cast_bf16": (_I, [_P, _P, _Z, _I, _P]),
    "devit_scale_cast_bf16": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "devit_colsum_workspace": (_Z, [_I, _I]),
    "devit_colsum_bf16": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "devit_sgemm_small": (_I, [_P, _LL, _LL, _P, _LL, _LL, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "devit_sumsq_workspace": (_Z, []),
    "devit_sumsq_f32": (_I, [_P, _Z, _P, _P, _Z, _P]),
    "devit_adamw_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _F, _F, _P]),
    "devit_cls_distill_loss": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "devit_token_mse": (_I, [_P, _P, _Z, _P, _P, _I, _P]),
    "devit_relation_stats": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "devit_relation_grad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "devit_gemm_f32": (_I, [_P, _LL, _LL, _LL, _LL, _P, _LL, _LL, _LL, _LL, _I, _I, _I, _I, _I, _LL, _LL, _I, _I, _F, _P, _I,
                            C.POINTER(Epilogue), _P]),
    "devit_softmax_rows_f32": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "devit_softmax_bwd_rows_f32": (_I, [_P, _P, _I, _I, _I, _F, _P]),
    "devit_im2row_f32": (_I, [_P, _P, _I, _P]),
    "devit_scale_rows_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "devit_colsum_f32": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "devit_comm_unique_id": (_I, [_P]),
    "devit_comm_init": (_I, [_P, _I, _I, C.POINTER(C.c_void_p)]),
    "devit_comm_allreduce_f32": (_I, [_P, _P, _Z, _P]),
    "devit_comm_destroy": (_I, [_P]),
}

_lib = None
_checked_devices = set()


def load():
    """Load libdevit_hip.so (no GPU needed to load; kernels need gfx950)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DevitError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or devit_amd/csrc/build.sh).  devit_amd has no CPU / PyTorch fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.devit_version() != ABI_VERSION:
            raise DevitError(f"libdevit_hip.so ABI version {lib.devit_version()} != {ABI_VERSION} of this binding: rebuild (devit_amd/csrc/build.sh)")
        for which, cls in ABI_STRUCTS.items():
            if lib.devit_abi_struct_size(which) != C.sizeof(cls):
                raise DevitError(f"libdevit_hip.so: sizeof struct {which} = {lib.devit_abi_struct_size(which)}, the binding's {cls.__name__} has "
                                 f"{C.sizeof(cls)} bytes: header and binding disagree")
        _lib = lib
    return _lib


def require_device(t: torch.Tensor):
    """Fail loudly unless `t` lives on a gfx950 GPU."""
    if not t.is_cuda:
        raise DevitError("devit_amd ops run on MI355X (gfx950) only; got a %s tensor. There is no CPU fallback "
                         "(the CPU restatement lives in oracle/ and is test infrastructure)." % t.device)
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if idx not in _checked_devices:
        call("devit_check_device", idx)
        _checked_devices.add(idx)


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise DevitError(f"{name} failed ({rc}): {lib.devit_last_error().decode()}")
    return rc


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
