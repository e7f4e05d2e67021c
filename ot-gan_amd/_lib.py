"""ctypes binding of csrc/libotgan_hip.so (the C ABI declared in include/otgan.h)."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OTGAN_LIB_PATH: a differently built library for A/B timing experiments -- dev only)
LIB_PATH = os.environ.get("OTGAN_LIB_PATH") or os.path.join(_HERE, "csrc", "libotgan_hip.so")
_lock = threading.Lock()
_lib = None

c_fp = ctypes.c_void_p   # device pointers travel as integers (tensor.data_ptr())
c_int, c_long, c_float, c_double, c_size_t = (ctypes.c_int, ctypes.c_long, ctypes.c_float,
                                               ctypes.c_double, ctypes.c_size_t)

# name -> (restype, argtypes); every symbol include/otgan.h declares
SIGNATURES = {
    "otgan_version": (c_int, []),
    "otgan_last_error": (ctypes.c_char_p, []),
    "otgan_prof_enable": (c_int, [c_int]),
    "otgan_sinkhorn_counters": (c_int, [c_int]),
    "otgan_sinkhorn_counters_read": (c_int, [c_fp, c_int]),
    "otgan_prof_reset": (c_int, []),
    "otgan_prof_collect": (c_int, [c_int, ctypes.POINTER(c_double)]),
    "otgan_matching_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "otgan_matching_two_batch_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int, c_int,
                                             c_fp, c_fp, c_fp, c_fp, c_long, c_fp, c_fp, c_fp,
                                             c_fp, c_size_t, c_fp]),
    "otgan_matching_two_batch_rows_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int, c_int,
                                                  c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_long, c_fp, c_fp,
                                                  c_fp, c_fp, c_size_t, c_fp]),
    "otgan_matching_grad_workspace_bytes": (c_size_t, [c_int, c_int]),
    "otgan_matching_two_batch_grad_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int, c_fp, c_fp, c_long,
                                                  c_fp, c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_matching_two_batch_rows_grad_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int, c_int, c_int,
                                                       c_fp, c_fp, c_fp, c_long, c_fp, c_fp, c_fp, c_fp, c_size_t,
                                                       c_fp]),
    "otgan_matching_stack_bytes": (c_size_t, [c_int, c_int]),
    "otgan_matching_stack_split_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_int, c_fp, c_fp, c_fp, c_fp]),
    "otgan_cost_slices_stack_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "otgan_cost_slices_stack_f32": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp, c_int, c_float, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_matching_two_batch_rows_grad_stack_f32": (c_int, [c_fp, c_int, c_int, c_float, c_int, c_int, c_int, c_fp, c_fp, c_fp,
                                                             c_long, c_fp, c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_matching_single_batch_grad_workspace_bytes": (c_size_t, [c_int, c_int]),
    "otgan_matching_single_batch_grad_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int, c_fp, c_fp, c_long,
                                                     c_fp, c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_matching_single_batch_rows_grad_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int, c_int, c_int,
                                                          c_fp, c_fp, c_fp, c_long, c_fp, c_fp, c_fp, c_fp, c_size_t,
                                                          c_fp]),
    "otgan_matching_single_batch_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_long, c_float, c_int,
                                                c_fp, c_fp, c_fp, c_fp, c_long, c_fp, c_fp, c_fp,
                                                c_fp, c_size_t, c_fp]),
    "otgan_cost_matrix_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "otgan_cost_matrix_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_long, c_float, c_int,
                                      c_float, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_cost_matrix_batched_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "otgan_cost_matrix_batched_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_long, c_float, c_int,
                                              c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_sinkhorn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "otgan_sinkhorn_plan_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_float, c_fp, c_fp, c_fp,
                                        c_fp, c_size_t, c_fp]),
    "otgan_plan_apply_f32": (c_int, [c_fp, c_long, c_int, c_int, c_fp, c_long, c_int, c_float,
                                     c_fp, c_long, c_fp]),
    "otgan_calc_distance_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_long, c_int, c_double,
                                        c_fp, c_fp, c_fp]),
}


def _register_layers(sig):
    try:
        from ._lib_layers import SIGNATURES as L
        sig.update(L)
    except ImportError:
        pass


_register_layers(SIGNATURES)


class OtganError(RuntimeError):
    pass


# Every OTGAN_* environment variable this build reads: the library's engine / test switches, the
# host side's, and the names only the test suite and its workers use.  Anything else that starts with OTGAN_ is a switch of an
# earlier round (or a typo) and changes NOTHING: say so once, so that an A/B over a dead knob cannot be read as "no difference"
# (ADVICE r5).
KNOWN_SWITCHES = frozenset("""
OTGAN_WINO_PIECES OTGAN_WINO_FP32 OTGAN_DISABLE_WINOGRAD OTGAN_X3_NARROW OTGAN_IGEMM_X3 OTGAN_WINO_UNFOLD_FUSED
OTGAN_MATCH_FP32 OTGAN_SINKHORN_LINEAR OTGAN_SINKHORN_LIN_RANGE OTGAN_SINKHORN_SETTLE
OTGAN_PANEL_XCD OTGAN_PANEL_MAX_WG OTGAN_DENSE16_CHAIN
OTGAN_LIB_PATH OTGAN_DIST_BACKEND OTGAN_FORCE_COLLECTIVES OTGAN_SINGLE_DEVICE OTGAN_COLLECTIVES OTGAN_SIDE_STREAM OTGAN_STEP_GRAPH
OTGAN_ROOT OTGAN_WORKER_CASES OTGAN_WORKER_DATA OTGAN_TEST_DIST_SEED
""".split())


def unknown_switches(environ=None):
    """OTGAN_* names in the environment that this build does not read."""
    environ = os.environ if environ is None else environ
    return sorted(k for k in environ if k.startswith("OTGAN_") and k not in KNOWN_SWITCHES)


def lib():
    """Load the HIP library once.  Fails loudly -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            dead = unknown_switches()
            if dead:
                import warnings
                warnings.warn("otgan_amd: " + ", ".join(dead) + " set in the environment but not read by this build (a switch of an "
                              "earlier round, or a typo): no effect.  Supported switches: INTEGRATION.md section 4.")
            if not os.path.exists(LIB_PATH):
                raise OtganError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                    f"g.build()'` (or `make -C ot-gan_amd/csrc`). There is no CPU fallback.")
            import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)  # AttributeError here == ABI/header drift: fail loudly
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().otgan_last_error().decode("utf-8", "replace")
        raise OtganError(f"{what} failed with code {rc}: {msg}")


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


# ---- profiler helpers -------------------------------------------------------------------
PROF_CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "cost_gemm", "sinkhorn", "plan_apply",
                "pointwise", "wino_gemm", "wino_gemm_bf16x3")


_prof_on = False


def prof_enable(on=True):
    """Per-launch HIP events around every library launch (otgan_prof_*).  While on, the trainer runs its steps eagerly:
    a replayed hipGraph makes no library calls (nothing to bracket), and an event recorded during a capture belongs to the
    graph, not to the profiler."""
    global _prof_on
    check(lib().otgan_prof_enable(1 if on else 0), "otgan_prof_enable")
    _prof_on = bool(on)


def prof_enabled():
    return _prof_on


def sinkhorn_counters(on=True):
    check(lib().otgan_sinkhorn_counters(1 if on else 0), "otgan_sinkhorn_counters")


def sinkhorn_counters_read(reset=False):
    """-> dict of the Sinkhorn kernels' sweep statistics since they were enabled / last reset (synchronises)."""
    import ctypes
    buf = (ctypes.c_longlong * 8)()
    check(lib().otgan_sinkhorn_counters_read(ctypes.cast(buf, ctypes.c_void_p), 1 if reset else 0), "otgan_sinkhorn_counters_read")
    keys = ("problems", "log_sweeps", "linear_sweeps", "entries", "fold_backs", "first_entry_sweep_sum", "never_entered")
    return {k: int(buf[i]) for i, k in enumerate(keys)}


def prof_reset():
    check(lib().otgan_prof_reset(), "otgan_prof_reset")


def prof_collect():
    out = {}
    buf = (c_double * 4)()
    for i, name in enumerate(PROF_CLASSES):
        check(lib().otgan_prof_collect(i, buf), "otgan_prof_collect")
        out[name] = {"launches": int(buf[0]), "ms": buf[1], "flop": buf[2], "bytes": buf[3]}
    return out
