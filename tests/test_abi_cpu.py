"""CPU (-m "not gpu"): the C-ABI library builds, loads, and exports every symbol include/unflow_hip.h declares;
the host-only entry points (geometry, workspace sizes, status strings) answer without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from unflow_amd import build, _lib
    build.build()
    return _lib.lib()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "unflow_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(unflow_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_status_strings_match_reference_messages(lib):
    lib.unflow_status_string.restype = ctypes.c_char_p
    assert lib.unflow_status_string(0) == b"ok"
    assert lib.unflow_status_string(-2) == b"Invalid correlation settings"          # correlation_op.cc:61
    assert lib.unflow_status_string(-3) == b"kernel_size must be odd"               # correlation_op.h:17
    assert lib.unflow_status_string(-4) == b"Input height and width must be divisible by scale"  # downsample_op.cc:40
    assert lib.unflow_status_string(-5) == b"Input shapes have to be the same"      # correlation_op.cc:48


def test_correlation_geometry_host_side(lib, oracle_lib):
    out = (ctypes.c_int * 3)()
    assert lib.unflow_correlation_out_shape(48, 64, 1, 20, 20, 1, 2, out) == 0
    assert tuple(out) == (441, 48, 64) == oracle_lib.correlation_out_shape(48, 64)
    for args in [(12, 14, 3, 4, 4, 2, 2), (9, 11, 3, 2, 3, 1, 1), (7, 9, 1, 3, 5, 1, 1), (16, 24, 1, 6, 6, 1, 1)]:
        assert lib.unflow_correlation_out_shape(*args, out) == 0
        H, W, k, md, pad, s1, s2 = args
        assert tuple(out) == oracle_lib.correlation_out_shape(H, W, kernel_size=k, max_displacement=md, pad=pad,
                                                              stride_1=s1, stride_2=s2)
    assert lib.unflow_correlation_out_shape(8, 8, 2, 20, 20, 1, 2, out) == -3
    assert lib.unflow_correlation_out_shape(8, 8, 1, 20, 0, 1, 2, out) == -2


def test_null_and_shape_errors_do_not_need_a_gpu(lib):
    n = ctypes.c_void_p(0)
    assert lib.unflow_downsample_fwd(n, n, 1, 8, 8, 3, 2, n) == -1
    assert lib.unflow_backward_warp_fwd(n, n, n, 1, 4, 4, 1, n) == -1
    assert lib.unflow_conv2d_fwd(n, 4, n, n, n, 4, 1, 8, 8, 4, 8, 3, 1, 1, n, ctypes.c_size_t(0), n) == -1
    assert lib.unflow_adam_step(n, n, n, n, ctypes.c_long(4), ctypes.c_long(4), ctypes.c_float(1), ctypes.c_float(0),
                                ctypes.c_float(1e-4), ctypes.c_float(.9), ctypes.c_float(.999), ctypes.c_float(1e-8), n) == -1


def test_workspace_queries(lib):
    lib.unflow_conv_workspace_bytes.restype = ctypes.c_size_t
    lib.unflow_correlation_workspace_bytes.restype = ctypes.c_size_t
    small = lib.unflow_conv_workspace_bytes(8, 192, 256, 64, 128, 5, 2)      # conv2: plenty of tiles, wgrad split only
    deep = lib.unflow_conv_workspace_bytes(8, 6, 8, 1024, 1024, 3, 1)        # conv6_1: split-K partials
    assert 0 < small < 1 << 30 and 0 < deep < 1 << 30
    # NHWC fp32 copies of in0, in1, g0, g1 and dout, + the bf16 x 3 operand planes of the two inputs where the matrix-core
    # kernels take the shape (kernel_size 1, stride_1 1, pad >= max_displacement, C % 16 == 0)
    fp32_part = (4 * 256 * 48 * 64 + 441 * 48 * 64) * 4
    assert lib.unflow_correlation_workspace_bytes(1, 256, 48, 64, 1, 20, 20, 1, 2) == fp32_part + 2 * 3 * 256 * 48 * 64 * 2 + 256
    assert lib.unflow_correlation_workspace_bytes(1, 20, 48, 64, 1, 20, 20, 1, 2) == (4 * 20 * 48 * 64 + 441 * 48 * 64) * 4   # C % 16 != 0


def test_product_path_never_imports_the_oracle():
    """The oracle is a checker: nothing under unflow_amd/ may import, load or call it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "unflow_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle|liboracle|ops_ref|model_ref", s, re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_pyr_level_struct_layout_matches_the_library():
    import ctypes
    from unflow_amd import _lib
    assert ctypes.sizeof(_lib.PyrLevel) == _lib.lib().unflow_sizeof_pyr_level()
    assert _lib.PyrLevel.H.offset == 7 * ctypes.sizeof(ctypes.c_void_p)


def test_work_order_of_the_plane_kernels_is_a_bijection(lib):
    """Every (M tile, N tile, parity class, K split) of a gather / halo launch is decoded by exactly one workgroup, for the
    three work orders, with and without the XCD-contiguous remap, including M-tile counts that are not multiples of 8
    (order 2 pads the grid; the padding workgroups decode M tile -1 and exit); with the remap on, the workgroups of one XCD
    (linear id mod 8) own a contiguous run of the order."""
    import itertools
    shapes = [(768, 1, 1, 1), (192, 2, 1, 2), (48, 4, 1, 8), (3, 8, 1, 16), (30, 3, 4, 2), (17, 1, 4, 5), (7, 5, 1, 3), (1, 1, 1, 1)]
    for (mt, nt, ncls, ns), order, xcd in itertools.product(shapes, (0, 1, 2), (0, 1)):
        grid = lib.unflow_debug_work_order(mt, nt, ncls, ns, order, xcd, None, 0)
        assert grid >= mt * nt * ncls * ns
        buf = (ctypes.c_int * (4 * grid))()
        assert lib.unflow_debug_work_order(mt, nt, ncls, ns, order, xcd, buf, grid) == grid
        seen = set()
        for b in range(grid):
            m, n, c, s = buf[4 * b:4 * b + 4]
            if m < 0:
                assert order == 2
                continue
            assert 0 <= m < mt and 0 <= n < nt and 0 <= c < ncls and 0 <= s < ns
            assert (m, n, c, s) not in seen
            seen.add((m, n, c, s))
        assert len(seen) == mt * nt * ncls * ns, (mt, nt, ncls, ns, order, xcd)
        if xcd and order == 2 and mt >= 16 and mt % 8 == 0:
            # one M group per XCD: the M tiles of XCD x are the x-th run of mt / 8 tiles
            for b in range(grid):
                m = buf[4 * b]
                assert m // (mt // 8) == b % 8
