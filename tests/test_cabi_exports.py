"""The C-ABI shared object loads on a machine without a GPU and exports every symbol include/pcrl.h
declares; argument validation that happens before any HIP call reports errors through the ABI."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcrl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcrl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for required in ("pcrl_encoder_fwd_f32", "pcrl_encoder_bwd_f32", "pcrl_encoder_pack_weights_f32", "pcrl_gemm_f32", "pcrl_gemm_group_f32",
                     "pcrl_tanh_gaussian_fwd_f32", "pcrl_tanh_gaussian_bwd_f32", "pcrl_sac_critic_loss_f32",
                     "pcrl_sac_actor_loss_f32", "pcrl_adam_step_f32", "pcrl_polyak_f32", "pcrl_last_error", "pcrl_version"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from pointcloud_rl_amd import _lib
    lib = _lib.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/pcrl.h but not exported by libpcrl_hip.so"
    assert lib.pcrl_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    from pointcloud_rl_amd import _lib
    lib = _lib.lib()
    n = ctypes.c_size_t()
    assert lib.pcrl_encoder_packed_bytes(6, 64, 128, 256, ctypes.byref(n)) == 0 and n.value > 4 * (64 * 6 + 128 * 64 + 256 * 128)
    rc = lib.pcrl_encoder_packed_bytes(6, 64, 128, 1024, ctypes.byref(n))      # c3 = 1024: not supported by the fused kernel
    assert rc == -1 and b"unsupported encoder dims" in lib.pcrl_last_error()
    assert lib.pcrl_encoder_packed_bytes(17, 64, 128, 256, ctypes.byref(n)) == -1            # more than 16 channels
    assert lib.pcrl_encoder_fwd_f32(None, None, None, None, None, None, None, ctypes.c_size_t(0), None) == -1
    assert lib.pcrl_gemm_f32(None, None) == -1
    with pytest.raises(_lib.PcrlError):
        _lib.check(-1)


def test_struct_layouts_match_the_header():
    from pointcloud_rl_amd import _lib
    # sizes the C compiler gives the structs of include/pcrl.h (LP64, natural alignment)
    assert ctypes.sizeof(_lib.FeatSeg) == 48
    assert ctypes.sizeof(_lib.CloudDesc) == 16 + 4 * 48
    assert ctypes.sizeof(_lib.AugDesc) == 64
    assert ctypes.sizeof(_lib.EncoderWeights) == 16 + 8 * 8 + 8
    assert ctypes.sizeof(_lib.GemmDesc) == 5 * 8 + 4 * 4 + 11 * 8 + 4 * 4 + 8 + 8
