"""ctypes binding of libneat_hip.so (C ABI declared in include/neat_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

NUM_LAYERS = 19
ABI_VERSION = 13
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16": 3, "fp16x3": 4}
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEAT_HIP_LIB") or os.path.join(_HERE, "csrc", "libneat_hip.so")      # NEAT_HIP_LIB: a probe build (scripts/probes/abl_build.sh)

c_fp = ctypes.c_void_p      # device float* (passed as integer addresses)


class NetParams(ctypes.Structure):
    _fields_ = [("v", c_fp * NUM_LAYERS), ("g", c_fp * NUM_LAYERS), ("b", c_fp * NUM_LAYERS)]


class NetGrads(ctypes.Structure):
    _fields_ = [("dv", c_fp * NUM_LAYERS), ("dg", c_fp * NUM_LAYERS), ("db", c_fp * NUM_LAYERS)]


_SIGNATURES = {
    "neat_abi_version": (ctypes.c_int, []),
    "neat_packed_floats": (ctypes.c_size_t, [ctypes.c_int]),
    "neat_pack_weights": (ctypes.c_int, [ctypes.POINTER(NetParams), c_fp, ctypes.c_int, c_fp]),
    "neat_eik_points": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "neat_camera_rays": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp]),
    "neat_sdf_ws_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "neat_sdf_forward": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "neat_sdf_backward": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp,
                                         ctypes.POINTER(NetGrads), c_fp]),
    "neat_heads_ws_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "neat_heads_forward": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp,
                                          c_fp, c_fp, c_fp]),
    "neat_render_ws_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "neat_render_forward": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           c_fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_fp,
                                           c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "neat_render_eval_ws_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "neat_render_forward_eval": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                c_fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_fp,
                                                c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "neat_render_backward": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, c_fp, ctypes.c_float, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(NetGrads), c_fp, c_fp,
                                            c_fp]),
    "neat_sampler_bound": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp,
                                          ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "neat_sampler_resample": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_float,
                                             c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "neat_sampler_finish": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "neat_sample_pdf": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "neat_uniform_depths": (ctypes.c_int, [c_fp, ctypes.c_float, c_fp, ctypes.c_float, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "neat_sdf_values_gated": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_float, c_fp, c_fp, c_fp, ctypes.c_int, c_fp]),
    "neat_sdf_values_rays": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, ctypes.c_int, c_fp]),
    "neat_sampler_init": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, ctypes.c_int,
                                         c_fp]),
    "neat_sampler_bound_dev": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp,
                                              ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp]),
    "neat_sampler_resample_dev": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_float,
                                                 c_fp, ctypes.c_int, c_fp, c_fp, c_fp,
                                                 c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, ctypes.c_int,
                                                 c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp]),
    "neat_sampler_finish_dev": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_float,
                                               ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "neat_sdf_ldp": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "neat_adam_step_coef": (ctypes.c_int, [c_fp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, c_fp, c_fp,
                                           c_fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_fp]),
    "neat_loss_lines_terms": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_float, c_fp, c_fp, ctypes.c_float,
                                             c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_int,
                                             c_fp, c_fp, c_fp, c_fp, ctypes.c_float, c_fp, c_fp, c_fp, c_fp]),
    "neat_camera_setup": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "neat_project2d_pair": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp]),
    "neat_sdf_values_laid_out": (ctypes.c_int, [c_fp, ctypes.POINTER(NetParams), ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, ctypes.c_int, c_fp]),
    "neat_sampler_init_rays": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "neat_sampler_round": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_fp, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, ctypes.c_int, c_fp]),
    "neat_sampler_finish_picked": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "neat_encode_lines": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "neat_gather_batch": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, ctypes.c_int,
                                         c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "neat_copy_batch": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, c_fp]),
    "neat_ffn_forward": (ctypes.c_int, [c_fp, ctypes.c_int] + [c_fp] * 10),
    "neat_ffn_backward": (ctypes.c_int, [c_fp, ctypes.c_int] + [c_fp] * 15),
    "neat_l3d": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "neat_junction_cost": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "neat_junction_gate": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp,
                                          c_fp, c_fp, c_fp]),
    "neat_loss_terms": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, ctypes.c_int,
                                       c_fp, c_fp, c_fp, c_fp, ctypes.c_float, c_fp]),
    "neat_loss_pairs": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp,
                                       c_fp, c_fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp]),
    "neat_camera_mats": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp]),
    "neat_inv_small": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "neat_project2d": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "neat_project2d_backward": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp, c_fp]),
    "neat_line_loss": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_float, c_fp, c_fp, c_fp, c_fp]),
    "neat_line_losses": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_float, c_fp, c_fp, ctypes.c_float, c_fp]),
    "neat_adam_step": (ctypes.c_int, [c_fp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int),
                                      ctypes.c_int, c_fp, c_fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_fp]),
    "neat_lsap_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "neat_lsap": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "neat_dbscan_ws_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "neat_dbscan_means": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_double, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "neat_volume_weights": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp]),
    "neat_set_tuning": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "neat_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "neat_prof_collect": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def exported_symbols():
    """Names declared in include/neat_hip.h (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGNATURES)


tuning_overrides = []      # the NEAT_TUNING key=value pairs applied when the library was loaded (bench.py records them)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or neat_amd/csrc/build.sh).  neat_amd has no CPU/PyTorch fallback for the hot path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
        # A/B switches from the environment (probe scripts; the defaults are what the tests and the bench run):
        # NEAT_TUNING="16=0,18=4" -> neat_set_tuning(16, 0), neat_set_tuning(18, 4)
        applied = []
        for kv in filter(None, os.environ.get("NEAT_TUNING", "").split(",")):
            k, v = kv.split("=")
            check(l.neat_set_tuning(int(k), int(v)), f"NEAT_TUNING {kv}")
            applied.append(kv)
        if applied:       # never silent: a stray variable changes what the tests and the bench measure
            import sys
            print(f"[neat_amd] NEAT_TUNING overrides applied: {','.join(applied)}", file=sys.stderr, flush=True)
        global tuning_overrides
        tuning_overrides = applied
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError(f"libneat_hip: {what} failed with code {code}")
