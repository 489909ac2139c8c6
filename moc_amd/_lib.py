"""ctypes binding of libmoc_hip.so (the C ABI in include/moc_hip.h).

No fallback: if the shared library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch first, always: its wheel bundles the HIP runtime (SONAME libamdhip64.so.7).  Loaded
# before libmoc_hip.so, the dynamic linker binds our NEEDED libamdhip64.so.7 to that same
# copy, so our launches share torch's streams and allocations.  Loaded after, a second HIP
# runtime would come up from /opt/rocm and find no usable device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOC_HIP_LIB") or os.path.join(_HERE, "libmoc_hip.so")
ABI_VERSION = 18

MOC_F32, MOC_BF16, MOC_F16 = 0, 1, 2
TICKET_WORDS = (64 + 8) * 64    # MOC_TICKET_WORDS (moc_batch_t.tile_ticket)
MOC_STATS_COMPACT = 1
MOC_SELECT_PER_COLUMN = 2
MOC_CAND_FROM_STATS = 4
MOC_FORWARD_ROWS64 = 8
MOC_FORWARD_FOUR_WAVES = 16
MOC_FORWARD_ROWS16 = 32
SEL_BITS = {"topk": 1, "delta_softmax": 2, "delta_diff": 4, "bottomk": 8}

_p = C.c_void_p


class MocBatch(C.Structure):
    _fields_ = [
        ("X", _p), ("dtype", C.c_int32), ("D", C.c_int32), ("total_rows", C.c_int64),
        ("n_slides", C.c_int32), ("max_rows", C.c_int32), ("row_off", _p), ("row_off_host", _p), ("x_off", _p), ("mask", _p),
        ("C", C.c_int32), ("Ce", C.c_int32), ("topj", C.c_int32), ("topk", C.c_int32),
        ("discard_bits", C.c_uint32), ("flags", C.c_uint32),
        ("kept", _p), ("n_kept", _p), ("stats", _p), ("sel_flag", _p), ("sel_idx", _p),
        ("sel_row", _p), ("n_sel", _p), ("cand", _p),
        ("cu_reserved", _p), ("tile_ticket", _p), ("n_sel_host", _p),
    ]


class MocMeta(C.Structure):
    _fields_ = (
        [(n, _p) for n in ("W1", "b1", "W2", "b2", "m_W1", "m_b1", "m_W2", "m_b2",
                           "v_W1", "v_b1", "v_W2", "v_b2", "g_W1", "g_b1", "g_W2", "g_b2", "W1_image")]
        + [(n, C.c_double) for n in ("lr", "beta1", "beta2", "eps", "weight_decay")]
        + [("H", C.c_int32), ("D", C.c_int32), ("step", C.c_int64)]
    )


class MocMetaWs(C.Structure):
    _fields_ = [(n, _p) for n in ("H1", "gates", "mixed", "pooled", "topk_idx", "topk_cnt",
                                  "loss", "pred", "pair_dh", "W2_alt", "pair_row", "n_pair", "tile_ws")] + [("tile_ws_bytes", C.c_int64)]


class MocRuns(C.Structure):
    _fields_ = [("n_runs", C.c_int32), ("slide_stride", C.c_int32), ("par_stride", C.c_int64), ("image_stride", C.c_int64)]


# name -> (restype, argtypes); every symbol include/moc_hip.h declares
_BP, _MP, _WP = C.POINTER(MocBatch), C.POINTER(MocMeta), C.POINTER(MocMetaWs)
SIGNATURES = {
    "moc_version": (C.c_int, []),
    "moc_last_error": (C.c_char_p, []),
    "moc_bank_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "moc_w1_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "moc_tile_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "moc_prepare_bank": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "moc_host_draw_masks": (C.c_int64, [_p, C.c_int64, C.c_int64, _p]),
    "moc_host_max_kept": (C.c_int64, [_p, _p, C.c_int]),
    "moc_mask_compact": (C.c_int, [_BP, _p]),
    "moc_scores": (C.c_int, [_BP, _p, _p]),
    "moc_scores_timed": (C.c_int, [_BP, _p, _p, _p, _p]),
    "moc_scores_from_cache": (C.c_int, [_BP, _p, C.c_int64, _p]),
    "moc_row_stats": (C.c_int, [_p, C.c_int64, C.c_int, C.c_int, _p, _p]),
    "moc_select": (C.c_int, [_BP, _p]),
    "moc_gather_candidates": (C.c_int, [_BP, _p, _p]),
    "moc_phase_a": (C.c_int, [_BP, _p, _p]),
    "moc_pack_selected": (C.c_int, [_BP, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "moc_pack_selected_rows": (C.c_int, [_BP, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int64, _p]),
    "moc_meta_forward": (C.c_int, [_BP, _MP, _WP, C.c_int, C.c_int, C.c_uint32, _p]),
    "moc_mix_fixed": (C.c_int, [_BP, _WP, C.c_int, C.c_int, C.c_int, _p]),
    "moc_pool_loss": (C.c_int, [_BP, _WP, _p, C.c_int, C.c_int, _p]),
    "moc_ce_loss": (C.c_int, [_p, _p, C.c_int, C.c_int, _p, _p, _p]),
    "moc_train_grad": (C.c_int, [_BP, _MP, _WP, _p, C.c_int, C.c_uint32, _p]),
    "moc_senet_backward": (C.c_int, [_p, C.c_int, C.c_int64, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "moc_adam_step": (C.c_int, [_MP, C.c_float, _p]),
    "moc_train_steps": (C.c_int, [_BP, _MP, _WP, _p, C.c_int, C.c_int, C.c_uint32, _p]),
    "moc_train_steps_runs": (C.c_int, [_BP, _MP, C.POINTER(MocRuns), _WP, _p, C.c_int, C.c_int, C.c_uint32, _p]),
    "moc_train_runs_mode": (C.c_int, [_BP, _WP]),
    "moc_step_graph_workspace_bytes": (C.c_size_t, [C.c_int]),
    "moc_step_graph_create": (C.c_int, [_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "moc_step_graph_destroy": (C.c_int, [_p]),
    "moc_step_graph_stats": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "moc_train_steps_graph": (C.c_int, [_p, _BP, _MP, _WP, _p, C.c_int, C.c_int, C.c_uint32, _p]),
    "moc_train_steps_dp": (C.c_int, [_BP, _MP, _WP, _p, C.c_int, C.c_int, C.c_uint32, _p, C.c_int64, _p, _p,
                                     C.c_int, _p]),
    "moc_p2p_handle_bytes": (C.c_int, []),
    "moc_p2p_create": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "moc_p2p_export": (C.c_int, [_p, _p]),
    "moc_p2p_connect": (C.c_int, [_p, _p]),
    "moc_p2p_allreduce": (C.c_int, [_p, _p, C.c_int64, _p]),
    "moc_p2p_error": (C.c_int, [_p]),
    "moc_p2p_destroy": (C.c_int, [_p]),
    "moc_p2p_step_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "moc_train_steps_p2p": (C.c_int, [_BP, _MP, _WP, _p, C.c_int, C.c_int, C.c_uint32, _p, _p]),
    "moc_gated_attention_workspace": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "moc_gated_attention_pool": (C.c_int, [_p, C.c_int64, C.c_int, _p, _p, _p, _p, C.c_int, _p, _p, C.c_int, _p, _p, _p,
                                           C.c_size_t, _p]),
    "moc_gated_attention_backward_workspace": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "moc_gated_attention_dab_stride": (C.c_int, [C.c_int, C.c_int]),
    "moc_gated_attention_backward": (C.c_int, [_p, C.c_int64, C.c_int, _p, _p, _p, _p, C.c_int, _p, C.c_int, _p, _p, _p,
                                               _p, _p, _p, _p, _p, C.c_size_t, _p]),
    "moc_cu_census": (C.c_int, [_p, C.c_int, C.c_int, _p]),
    "moc_topk_mean": (C.c_int, [_p, C.c_int64, _p, C.c_int64, _p, _p, C.c_int, C.c_int, C.c_int,
                                C.c_int, _p, _p, _p, _p]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C moc_amd/csrc` "
                "(or __graft_entry__.build()).  moc_amd has no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        _assert_single_hip_runtime()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.moc_version() != ABI_VERSION:
            raise RuntimeError(f"libmoc_hip ABI {h.moc_version()} != binding {ABI_VERSION}; rebuild")
        _lib = h
    return _lib


def _assert_single_hip_runtime():
    try:
        maps = open("/proc/self/maps").read()
    except OSError:
        return
    copies = {line.split()[-1] for line in maps.splitlines() if "libamdhip64.so" in line}
    if len(copies) > 1:
        raise RuntimeError(f"two HIP runtimes in one process ({sorted(copies)}): import torch before moc_amd")


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().moc_last_error().decode(errors="replace")
        # shape/contract violations mirror the reference's assert style
        if rc == 1:
            raise AssertionError(msg or what)
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """data pointer of a torch tensor, or None."""
    return None if t is None else t.data_ptr()


def discard_bits(names) -> int:
    bits = 0
    for n in names or ():
        bits |= SEL_BITS.get(n, 0)   # unknown strings are ignored, as `in` tests in the reference do
    return bits
