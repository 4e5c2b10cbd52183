"""ctypes binding of the C ABI in include/xumx_slicq_hip.h.

The HIP library is the product: there is NO CPU fallback.  Importing this
module without the built library raises immediately with build instructions,
and every call checks its status code and raises ``XsqError`` with the
library's message.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must come first: it loads the ROCm runtime this library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# XSQ_LIB selects a diagnostic build (tools/ablate.sh); the default is the product library.
LIB_PATH = os.environ.get("XSQ_LIB") or os.path.join(_HERE, "libxumx_slicq_hip.so")


class XsqError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise XsqError(
            f"{LIB_PATH} is missing: the HIP extension is the product path and has no CPU "
            "fallback. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C xumx_slicq_amd/csrc`.")
    # torch has already mapped its own libamdhip64.so / librocfft.so (same SONAMEs as the
    # system ROCm ones this library was linked against), so the loader resolves our
    # NEEDED entries to the copies torch uses and both sides share one HIP runtime.
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_vp = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)

_SIGS = {
    "xsq_abi_version": (C.c_int, []),
    "xsq_last_error": (C.c_char_p, []),
    "xsq_build_info": (C.c_char_p, []),
    "xsq_plan_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "xsq_plan_destroy": (C.c_int, [_vp]),
    "xsq_plan_num_blocks": (C.c_int, [_vp]),
    "xsq_plan_set_fft_backend": (C.c_int, [_vp, C.c_int]),
    "xsq_plan_set_band_radix4": (C.c_int, [_vp, C.c_int]),
    "xsq_plan_set_short_inline": (C.c_int, [_vp, C.c_int]),
    "xsq_plan_set_packed_fft": (C.c_int, [_vp, C.c_int]),
    "xsq_plan_block_table": (C.c_int, [_vp, _vp]),
    "xsq_plan_coefs_per_slice": (C.c_int64, [_vp]),
    "xsq_plan_num_slices": (C.c_int, [_vp, C.c_int64]),
    "xsq_slicqt_forward_workspace": (C.c_size_t, [_vp, C.c_int, C.c_int64]),
    "xsq_slicqt_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp, C.c_size_t, _vp]),
    "xsq_slicqt_forward_xin": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "xsq_slicqt_forward_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "xsq_slicqt_forward_rows_indirect": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "xsq_slicqt_inverse_workspace": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "xsq_slicqt_inverse": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int64, _vp, _vp, C.c_size_t, _vp]),
    "xsq_slicqt_inverse_rows": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp, C.c_size_t, _vp]),
    "xsq_slicqt_inverse_masked": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp, C.c_size_t, _vp]),
    "xsq_model_num_params": (C.c_int64, [C.c_int, _vp, _vp]),
    "xsq_model_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _vp, C.c_int, _vp, C.c_int64]),
    "xsq_model_destroy": (C.c_int, [_vp]),
    "xsq_model_set_precision": (C.c_int, [_vp, C.c_int]),
    "xsq_model_set_winograd": (C.c_int, [_vp, C.c_int]),
    "xsq_cdae_workspace": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "xsq_cdae_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "xsq_cdae_forward_xin": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, C.c_int]),
    "xsq_model_whitening": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int)]),
    "xsq_phasemix": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "xsq_wiener_workspace": (C.c_size_t, [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "xsq_loss_workspace": (C.c_size_t, [C.c_int, _vp, _vp, C.c_int, C.c_int]),
    "xsq_loss_forward": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "xsq_magnitude_stats": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "xsq_train_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, _vp, C.c_int, _vp, C.c_int64]),
    "xsq_train_destroy": (C.c_int, [_vp]),
    "xsq_train_workspace": (C.c_size_t, [_vp, C.c_int, C.c_int, C.c_int]),
    "xsq_train_step": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "xsq_train_ticket": (C.c_int64, [_vp]),
    "xsq_train_loss": (C.c_int, [_vp, C.c_int64, _vp]),
    "xsq_train_read": (C.c_int, [_vp, C.c_int, _vp]),
    "xsq_train_write": (C.c_int, [_vp, C.c_int, _vp]),
    "xsq_train_step_count": (C.c_int64, [_vp, C.c_int64]),
    "xsq_train_set_precision": (C.c_int, [_vp, C.c_int]),
    "xsq_place_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, _vp]),
    "xsq_wiener_num_windows": (C.c_int64, [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "xsq_wiener_window_max": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "xsq_wiener_em_masked_ext": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "xsq_demixer_create": (C.c_int, [C.POINTER(_vp), _vp]),
    "xsq_demixer_destroy": (C.c_int, [_vp]),
    "xsq_demixer_set_max_rows": (C.c_int, [_vp, C.c_int]),
    "xsq_separator_schedule": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, C.c_int]),
    "xsq_demix_pass_workspace": (C.c_size_t, [_vp, _vp, C.c_int, C.c_int64, C.c_int]),
    "xsq_demix_pass": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "xsq_separator_workspace": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "xsq_separator_forward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]),
    "xsq_separator_forward_indirect": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]),
    "xsq_comm_load": (C.c_int, [C.c_char_p]),
    "xsq_comm_version": (C.c_int, []),
    "xsq_comm_unique_id": (C.c_int, [_vp]),
    "xsq_comm_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "xsq_comm_destroy": (C.c_int, [_vp]),
    "xsq_comm_abort": (C.c_int, [_vp]),
    "xsq_exchange_rows": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int, C.c_int, _vp]),
    "xsq_profile_enable": (C.c_int, [C.c_int]),
    "xsq_profile_reset": (C.c_int, []),
    "xsq_profile_filter": (C.c_int, [C.c_char_p]),
    "xsq_profile_read": (C.c_int, [C.c_char_p, C.c_size_t, _vp, _vp, C.c_int]),
    "xsq_wiener_em": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "xsq_wiener_em_masked": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGS)


def build_info() -> dict:
    """What is loaded: path, ABI version, the library's own build string (xsq_build_info)."""
    return {"path": LIB_PATH, "default_path": LIB_PATH == os.path.join(_HERE, "libxumx_slicq_hip.so"),
            "abi": int(lib.xsq_abi_version()), "build": (lib.xsq_build_info() or b"").decode("utf-8", "replace")}


def xsq_environment() -> dict:
    """Every XSQ_* variable set in this process: most of them select kernels, fusions or diagnostic paths (csrc getenv,
    separator.py / model.py / transforms.py switches).  bench.py records them and flags a line measured with any."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("XSQ_")}


def last_error() -> str:
    return (lib.xsq_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise XsqError(f"{what} failed ({rc}): {last_error()}")


def stream_ptr() -> int:
    """The HIP stream torch is currently issuing work on."""
    return torch.cuda.current_stream().cuda_stream


def profile_enable(on: bool = True):
    lib.xsq_profile_enable(1 if on else 0)


def profile_reset():
    lib.xsq_profile_reset()


def profile_filter(name=None):
    """Time only the kernel launched under this event name (None: all kernels)."""
    lib.xsq_profile_filter(name.encode() if name else None)


def profile_read() -> dict:
    """{kernel name: (total ms, launches)} accumulated since the last reset."""
    import numpy as np
    buf = C.create_string_buffer(8192)
    ms = np.zeros(128, dtype=np.float64)
    cnt = np.zeros(128, dtype=np.int64)
    n = lib.xsq_profile_read(buf, len(buf), ms.ctypes.data, cnt.ctypes.data, 128)
    names = buf.value.decode().split("\n")[:n]
    return {nm: (float(ms[i]), int(cnt[i])) for i, nm in enumerate(names)}
