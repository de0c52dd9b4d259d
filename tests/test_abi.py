"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/neat_hip.h declares; the Python binding declares the same set.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "neat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(neat_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from neat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = header_functions()
    assert len(names) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/neat_hip.h but not exported"
    assert names == _lib.exported_symbols()
    assert _lib.lib().neat_abi_version() == _lib.ABI_VERSION


def test_workspace_queries_are_consistent():
    from neat_amd import _lib
    lib = _lib.lib()
    for prec, tile in ((0, 64), (1, 128)):
        assert lib.neat_packed_floats(prec) > 19 * 256
        assert lib.neat_sdf_ws_floats(64, 0, prec) < lib.neat_sdf_ws_floats(64, 1, prec)
        assert lib.neat_render_ws_floats(8, 8, 0, prec) > lib.neat_sdf_ws_floats(64, 1, prec)
        assert lib.neat_render_ws_floats(8, 8, 0, prec) == lib.neat_render_ws_floats(8, 7, 8, prec)     # E extra points share the tile grid
        # point stride is padded to the workgroup's point tile (64 fp32 / 128 bf16)
        assert lib.neat_sdf_ws_floats(1, 0, prec) == lib.neat_sdf_ws_floats(tile, 0, prec)
        assert lib.neat_sdf_ws_floats(tile + 1, 0, prec) == lib.neat_sdf_ws_floats(2 * tile, 0, prec)
    assert lib.neat_packed_floats(7) == 0          # unknown precision is rejected
    # NEAT_F16 (3) = the bf16 build's layouts with IEEE half as the 16-bit type; NEAT_BF16X3 (2) = the fp32 build's
    assert lib.neat_packed_floats(3) == lib.neat_packed_floats(1) and lib.neat_packed_floats(2) == lib.neat_packed_floats(0)
    assert lib.neat_render_ws_floats(64, 32, 8, 3) == lib.neat_render_ws_floats(64, 32, 8, 1)
    assert lib.neat_render_eval_ws_floats(64, 32, 3) == lib.neat_render_eval_ws_floats(64, 32, 1)
    assert lib.neat_sdf_ws_floats(100, 1, 3) == lib.neat_sdf_ws_floats(100, 1, 1) and lib.neat_heads_ws_floats(100, 3) == lib.neat_heads_ws_floats(100, 1)


def test_copy_batch_rejects_bad_arguments_before_any_launch():
    """neat_copy_batch validates on the host: more than 16 copies, null tables, misaligned pointers and sizes that are not multiples of 4
    return -1 without touching a device (so this runs without a GPU); zero copies is a no-op."""
    from neat_amd import _lib
    lib = _lib.lib()
    buf = (ctypes.c_float * 16)()
    base = ctypes.addressof(buf)
    def call(srcs, dsts, sizes):
        n = len(srcs)
        return lib.neat_copy_batch((ctypes.c_void_p * n)(*srcs), (ctypes.c_void_p * n)(*dsts), (ctypes.c_longlong * n)(*sizes), n, None)
    assert lib.neat_copy_batch(None, None, None, 0, None) == 0
    assert lib.neat_copy_batch(None, None, None, 3, None) == -1
    assert call([base] * 17, [base + 32] * 17, [4] * 17) == -1            # at most 16 per launch (ops.copy_batch chunks)
    assert call([base + 2], [base + 32], [4]) == -1                        # source not 4-byte aligned
    assert call([base], [base + 33], [4]) == -1                            # destination not 4-byte aligned
    assert call([base], [base + 32], [6]) == -1                            # size not a multiple of 4
    assert call([base], [base + 32], [-4]) == -1
    assert call([base, 0], [base + 32, base + 48], [4, 4]) == -1           # a null source


def test_ops_refuse_cpu_tensors():
    import torch
    from neat_amd import networks, synth
    m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
    with pytest.raises(RuntimeError):
        m.implicit_network.get_sdf_vals(torch.zeros(4, 3))      # no CPU fallback: must fail loudly


def test_torch_ops_are_registered_for_the_gpu_only():
    """torch.ops.neat_hip.* (neat_amd/torch_ops.py) exist with tensor/scalar schemas and have no CPU kernel: the dispatcher refuses."""
    import torch
    from neat_amd import torch_ops
    for name in torch_ops.OPS:
        schema = str(getattr(torch.ops.neat_hip, name).default._schema)
        assert schema.startswith(f"neat_hip::{name}("), schema
    with pytest.raises(NotImplementedError):
        torch.ops.neat_hip.volume_weights(torch.zeros(2, 4), torch.zeros(2, 4), torch.ones(1))
    with pytest.raises(NotImplementedError):
        torch.ops.neat_hip.sdf_values(torch.zeros(4, 3), [torch.zeros(1)] * 27, 3.0, 20.0, 0)
