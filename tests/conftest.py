import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# Order of the `-m gpu` run (the driver runs it with -x): kernel parity against the oracle FIRST -- forward, backward, dense heads, head
# tails, the whole update step, the full-size steps, bf16 -- then augmentations / colour / integration, and every test that starts other
# processes (bench.py launches, mp.spawn ranks, the eight-rank rehearsal) LAST, the rehearsal at the very end.  A flaky launcher test must
# never stand between the driver and the parity evidence (round 5: alphabetical order put test_data_parallel_gpu.py ahead of all of it).
_GPU_FILE_ORDER = [
    "test_encoder_fwd_gpu.py", "test_encoder_bwd_gpu.py", "test_dense_tail_gpu.py", "test_gemm_plan.py", "test_headtail_gpu.py",
    "test_update_step_gpu.py", "test_fullsize_parity_gpu.py", "test_k2_fullsize_bf16_gpu.py",
    "test_aux_aug_acting_gpu.py", "test_color_jitter_gpu.py", "test_cabi_exports.py", "test_integration_stub_gpu.py",
    "test_reference_integration.py", "test_data_parallel_gpu.py",
]
_LAST_TESTS = ("test_bench_dry_run_of_the_eight_rank_launch",)


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_GPU_FILE_ORDER)}

    def key(pair):
        pos, item = pair
        if item.get_closest_marker("gpu") is None:
            return (0, 0, pos)                                  # CPU tests keep their collection order, ahead of everything
        fname = os.path.basename(str(item.fspath))
        last = any(item.name.startswith(n) for n in _LAST_TESTS)
        return (2 if last else 1, rank.get(fname, len(rank) - 1), pos)

    items[:] = [it for _, it in sorted(enumerate(items), key=key)]
