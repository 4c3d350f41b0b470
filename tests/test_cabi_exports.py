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
    rc = lib.pcrl_encoder_packed_bytes(6, 64, 128, 512, ctypes.byref(n))       # not one of the built (c1, c2, c3) shapes
    assert rc == -1 and b"unsupported encoder dims" in lib.pcrl_last_error()
    assert lib.pcrl_encoder_packed_bytes(17, 64, 128, 256, ctypes.byref(n)) == -1            # more than 16 channels
    assert lib.pcrl_encoder_fwd_f32(None, None, None, None, None, None, None, ctypes.c_size_t(0), None) == -1
    assert lib.pcrl_gemm_f32(None, None) == -1
    with pytest.raises(_lib.PcrlError):
        _lib.check(-1)


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct of include/pcrl.h as gcc lays it out == the ctypes mirror in _lib.py."""
    import subprocess
    from pointcloud_rl_amd import _lib
    pairs = [("pcrl_feat_seg", _lib.FeatSeg), ("pcrl_cloud_desc", _lib.CloudDesc), ("pcrl_aug_desc", _lib.AugDesc),
             ("pcrl_encoder_weights", _lib.EncoderWeights), ("pcrl_gemm_desc", _lib.GemmDesc), ("pcrl_ln_job", _lib.LnJob),
             ("pcrl_gather_seg", _lib.GatherSeg), ("pcrl_adam_pending", _lib.AdamPending), ("pcrl_colsum_job", _lib.ColsumJob),
             ("pcrl_adam_rider", _lib.AdamRider), ("pcrl_col_gather", _lib.ColGather)]
    lines = []
    for cname, cls in pairs:
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pcrl.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_binding_loads_torch_before_the_library():
    """One HIP runtime per process: the binding imports torch (whose wheel carries its own libamdhip64) before it maps
    libpcrl_hip.so, whatever the caller imported first -- build() followed by smoke() in one process failed on the GPU box
    ("no ROCm-capable device is detected" from the library's launches) when the library came first."""
    import subprocess
    import sys
    code = ("import sys; from pointcloud_rl_amd import _lib; assert 'torch' not in sys.modules; "
            "_lib.lib(); assert 'torch' in sys.modules; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_launch_tracer_marks_every_entry_point(tmp_path):
    """PCRL_TRACE_LAUNCHES=<dir> (pointcloud_rl_amd/_lib.py::_TracedLib): `> name` before and `< name rc` after every pcrl_* call, one file per
    process, written unbuffered -- what names the launch in flight of a rank that dies of an asynchronous GPU fault (round 6: the 8-rank
    rehearsal).  Host-only entry points here: no GPU."""
    import subprocess
    import sys
    code = ("import ctypes; from pointcloud_rl_amd import _lib; l = _lib.lib(); assert l.pcrl_version() == 100; n = ctypes.c_size_t(); "
            "assert l.pcrl_adam_workspace_bytes(ctypes.c_size_t(5000), ctypes.byref(n)) == 0 and n.value > 0; "
            "assert l.pcrl_encoder_bwd_set_fused(7) != 0; assert b'mode' in l.pcrl_last_error(); print('ok')")
    env = dict(os.environ, PCRL_TRACE_LAUNCHES=str(tmp_path), PCRL_TRACE_SYNC="0", RANK="3")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    files = list(tmp_path.glob("rank3_pid*.trace"))
    assert len(files) == 1, list(tmp_path.iterdir())
    lines = files[0].read_text().splitlines()
    assert lines[:4] == ["> pcrl_version", "< pcrl_version 100", "> pcrl_adam_workspace_bytes", "< pcrl_adam_workspace_bytes 0"], lines
    assert lines[4] == "> pcrl_encoder_bwd_set_fused" and lines[5].startswith("< pcrl_encoder_bwd_set_fused -"), lines      # pcrl_last_error is not traced
    assert len(lines) == 6


_ORDER_PROBE = """
import ctypes, json, os, sys
sys.path.insert(0, {root!r})
from pointcloud_rl_amd import _lib
links = _lib.ensure_runtime_links()                  # what `make` / the first _lib.lib() of a machine leaves behind
assert 'torch' not in sys.modules
def runtimes():
    return sorted({{l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}})
lib = ctypes.CDLL(_lib.LIB_PATH)                     # an embedder maps the library itself, BEFORE torch
first = runtimes()
import torch
both = runtimes()
out = dict(links=links, first=first, both=both, torch_lib=os.path.join(os.path.dirname(torch.__file__), 'lib'), version=lib.pcrl_version())
if {gpu}:
    lib.pcrl_last_error.restype = ctypes.c_char_p
    t, s_ = torch.zeros(1000, device='cuda'), torch.arange(1000, device='cuda', dtype=torch.float32)
    rc = lib.pcrl_polyak_f32(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(s_.data_ptr()), ctypes.c_size_t(1000), ctypes.c_float(0.25),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    out.update(rc=rc, err=(lib.pcrl_last_error() or b'').decode(), ok=bool(torch.equal(t, 0.25 * s_)), torch_ok=float((s_ * 2).sum()) == 999000.0)
print('ORDER_JSON ' + json.dumps(out))
"""


def _order_probe(gpu):
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _ORDER_PROBE.format(root=ROOT, gpu=gpu)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("ORDER_JSON ")][0].split(" ", 1)[1])


def test_mapping_the_library_before_torch_leaves_one_hip_runtime():
    """libpcrl_hip.so searches pointcloud_rl_amd/_hiprt first (RPATH; symlinks to the runtime torch's wheel bundles, made by `make` /
    _lib.ensure_runtime_links): mapped BEFORE `import torch` it brings in torch's own libamdhip64, and torch then re-uses that
    mapping -- one HIP runtime in the process whatever the load order (round 3: /opt/rocm's came first and torch added a second)."""
    d = _order_probe(gpu=False)
    assert d["links"] and d["version"] >= 100
    assert len(d["first"]) == 1 and d["first"][0].startswith(d["torch_lib"]), d
    assert d["both"] == d["first"], d


@pytest.mark.gpu
def test_library_mapped_before_torch_still_launches(cuda):
    d = _order_probe(gpu=True)
    assert d["both"] == d["first"] and len(d["first"]) == 1, d
    assert d["rc"] == 0 and d["ok"] and d["torch_ok"], d
