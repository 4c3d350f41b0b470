"""Host side of pcrl_gemm_group_f32 (csrc/dense.hip): which tile path and shape a launch takes.  No GPU: the planner only reads the
descriptors (pointers are never dereferenced; without a device it assumes the MI355X's 256 CUs)."""
import ctypes

import pytest

from pointcloud_rl_amd import _lib

BASE = 1 << 20            # a 16-byte aligned, never dereferenced address


def desc(M, N, K, a_strides, b_strides, batch=1, ones_col=-1, a_off=0, b_off=0):
    d = _lib.GemmDesc()
    d.A, d.B, d.C = BASE + 4 * a_off, BASE + 4 * b_off, BASE
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.a_stride_m, d.a_stride_k = a_strides
    d.b_stride_k, d.b_stride_n = b_strides
    d.ldc, d.ones_col = N, ones_col
    d.a_batch_stride, d.b_batch_stride, d.c_batch_stride = M * K, N * K, M * N
    return d


def plan(descs):
    arr = (_lib.GemmDesc * len(descs))(*descs)
    out = (ctypes.c_int32 * (3 * len(descs)))()
    _lib.check(_lib.lib().pcrl_gemm_group_plan_f32(arr, len(descs), out))
    return [tuple(out[3 * i:3 * i + 3]) for i in range(len(descs))]


def fwd(M, N, K, heads=1, **kw):
    return desc(M, N, K, (K, 1), (1, K), batch=heads, **kw)


def dgrad(M, N, K, heads=1):
    return desc(M, N, K, (K, 1), (N, 1), batch=heads)


def wgrad(M, N, K, heads=1, ones=True):
    return desc(M, N + (1 if ones else 0), K, (1, M), (N, 1), batch=heads, ones_col=N if ones else -1)


@pytest.fixture(autouse=True)
def default_knob():
    prev = _lib.lib().pcrl_gemm_set_tile64_min(192)
    yield
    _lib.lib().pcrl_gemm_set_tile64_min(prev)


def test_the_finest_shape_with_at_most_one_workgroup_per_cu():
    # the heads' 1 024 x 1 024 layer at a rank's 32 ... 256 rows, 1 / 2 / 4 heads per launch (tools/probes/gemm_staged.hip measured these)
    assert plan([fwd(32, 1024, 1024)]) == [(4, 0, 128)]
    assert plan([fwd(32, 1024, 1024, 2)]) == [(4, 0, 256)]
    assert plan([fwd(32, 1024, 1024, 4)]) == [(4, 1, 256)]
    assert plan([fwd(128, 1024, 1024)]) == [(4, 1, 256)]
    assert plan([fwd(256, 1024, 1024)]) == [(4, 2, 256)]
    assert plan([fwd(256, 1024, 1024, 2)]) == [(4, 3, 256)]
    assert plan([fwd(256, 1024, 1024, 4)]) == [(1, 0, 256)]            # beyond the coarsest shape: the staged 64 x 64 tiles
    assert plan([fwd(1024, 1024, 1024, 4)]) == [(1, 0, 1024)]
    assert plan([dgrad(32, 1024, 1024, 2)]) == [(4, 0, 256)]
    assert plan([dgrad(128, 1024, 1024)]) == [(4, 1, 256)]
    assert plan([dgrad(256, 1024, 1024, 2)]) == [(4, 2, 512)]          # (no 32 x 64 shape with a row-contiguous B)
    assert plan([dgrad(1024, 1024, 1024, 2)]) == [(1, 0, 512)]
    assert plan([dgrad(256, 50, 1024, 2)]) == [(4, 0, 128)]            # the first layer's data gradient: 50 of the 56 input columns


def test_two_forward_problems_of_one_launch_are_planned_together():
    # the target and the online Q heads' second layer: each alone is one round of 32 x 64 wave tiles, together two -- one round of 64 x 64
    # staged tiles instead (round 6); at 128 rows those would leave half the chip idle and the wave tiles stay
    assert plan([fwd(256, 1024, 1024, 2), fwd(256, 1024, 1024, 2)]) == [(1, 0, 128), (1, 0, 128)]
    assert plan([fwd(128, 1024, 1024, 2), fwd(128, 1024, 1024, 2)]) == [(4, 2, 256), (4, 2, 256)]
    assert plan([fwd(256, 1024, 1024, 2)]) == [(4, 3, 256)]                                  # one problem, one round: unchanged
    # a short first layer in the same launch keeps its own path and does not count
    assert plan([fwd(256, 1024, 1024, 1), fwd(256, 1024, 56, 2)])[0] == (4, 2, 256)
    # the weight / data gradient pair is not forward-shaped: unchanged
    assert plan([wgrad(1024, 1024, 256, 2), dgrad(256, 1024, 1024, 2)]) == [(5, 0, 256), (4, 2, 512)]


def test_weight_gradient_panels_and_their_fallbacks():
    assert plan([wgrad(1024, 1024, 256, 2)]) == [(5, 0, 256)]          # 64 x 128 panels: one per CU
    assert plan([wgrad(1024, 1024, 256, 1)]) == [(5, 1, 256)]          # 64 x 64
    assert plan([wgrad(1024, 1024, 32, 2)]) == [(5, 0, 256)]
    assert plan([wgrad(1024, 56, 256, 2)]) == [(0, 0, 128)]            # fewer than 256 real columns: 32 x 32 split-K tiles
    assert plan([wgrad(128, 256, 512, 1)])[0][0] == 0                  # the feature head's weight gradient: too few panels to be worth it
    assert plan([wgrad(1022, 1024, 256, 2)])[0][0] == 3                # M % 4 != 0: a tile per wave (pairs of rows)
    assert plan([wgrad(1024, 1024, 256, 2, ones=False)]) == [(5, 0, 256)]


def test_short_or_odd_contractions_keep_the_split_k_tiles():
    assert plan([fwd(256, 1024, 56)])[0][0] == 0                       # K < 512
    assert plan([fwd(512, 1024, 196)])[0][0] == 0                      # K3 / K2's first layers: the staging prologue does not pay for two chunks
    assert plan([fwd(256, 1024, 1022)])[0][0] == 0                     # K % 4 != 0
    assert plan([fwd(256, 1024, 1024, a_off=1)])[0][0] == 0             # A not 16-byte aligned
    assert plan([desc(0, 5, 8, (8, 1), (1, 8))]) == [(-1, 0, 0)]       # empty problem


def test_the_knob_selects_the_older_paths():
    lib = _lib.lib()
    lib.pcrl_gemm_set_tile64_min(1 << 30)
    assert plan([fwd(256, 1024, 1024, 2)]) == [(0, 0, 512)]
    assert plan([wgrad(1024, 1024, 256, 2)])[0][0] == 3
    lib.pcrl_gemm_set_tile64_min(1)
    assert plan([fwd(256, 1024, 1024)]) == [(1, 0, 64)]
    assert plan([dgrad(256, 1024, 1024)]) == [(1, 0, 64)]


def test_a_group_keeps_each_problems_own_path_and_dispatches_the_long_k_problem_first():
    got = plan([wgrad(1024, 1024, 128, 2), dgrad(128, 1024, 1024, 2), fwd(70, 33, 56), fwd(128, 1024, 1024, 2)])
    assert [g[0] for g in got] == [5, 4, 0, 4]
    with pytest.raises(_lib.PcrlError):
        plan([fwd(8, 8, 8)] * 5)


def test_a_mixed_launch_keeps_each_problems_path():
    # the four-head forward takes the 64 x 64 staged tiles, its neighbour its wave tiles
    got = plan([fwd(256, 1024, 1024, 4), fwd(128, 1024, 1024)])
    assert got[0][0] == 1 and got[1][0] == 4, got                # each problem keeps its own path: the launch takes the kernel that has both
    assert plan([fwd(128, 1024, 1024)])[0][0] == 4


def test_encoder_backward_schedule_knob_validates_its_mode():
    """pcrl_encoder_bwd_set_fused (include/pcrl.h): 0 / 1 / 2 are accepted, anything else is PCRL_E_ARG and changes nothing (no GPU needed:
    the knob is host state read by the next backward)."""
    import ctypes as ct
    lib = _lib.lib()
    workspace = {}
    for B in (32, 1024):        # the workspace holds both schedules' regions whatever the mode: its size does not depend on the knob
        sizes = []
        for mode in (0, 2, 1):
            assert lib.pcrl_encoder_bwd_set_fused(mode) == 0
            n = ct.c_size_t()
            _lib.check(lib.pcrl_encoder_bwd_workspace_bytes(B, 6, 64, 128, 256, ct.byref(n)))
            sizes.append(n.value)
        assert len(set(sizes)) == 1 and sizes[0] > 0
        workspace[B] = sizes[0]
    assert workspace[1024] > workspace[32]
    for bad in (-1, 3, 17):
        assert lib.pcrl_encoder_bwd_set_fused(bad) < 0
        assert b"mode" in lib.pcrl_last_error()
    assert lib.pcrl_encoder_bwd_set_fused(1) == 0
