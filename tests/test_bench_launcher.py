"""bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run), as far as a box without a GPU can show:
the parent starts N children with the rendezvous variables set, touches no GPU itself, and reports their failure as its own."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_gpus_n_invocation_starts_n_ranks_and_propagates_their_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the two-rank run itself is the test (tests/test_data_parallel_gpu.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 1
    assert 1 <= out.stderr.count("bench.py needs an MI355X") <= 3      # children ran main() as ranks (the first to fail ends the others)
    assert "bench.py launcher: 3 ranks failed" in out.stderr and "rank" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]            # no result line is invented


def test_a_rank_sees_the_launchers_rendezvous_variables(tmp_path):
    """The children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT: checked with a stand-in interpreter
    hook (PYTHONSTARTUP is not read by scripts, so the check goes through sitecustomize on PYTHONPATH)."""
    hook = tmp_path / "sitecustomize.py"
    hook.write_text(
        "import os, sys\n"
        "if os.environ.get('WORLD_SIZE') and sys.argv and sys.argv[0].endswith('bench.py'):\n"
        "    open(os.path.join(os.environ['PCRL_TEST_DIR'], 'rank' + os.environ['RANK']), 'w').write(\n"
        "        ' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the two-rank run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=str(tmp_path) + os.pathsep + env.get("PYTHONPATH", ""), PCRL_TEST_DIR=str(tmp_path), PCRL_CAPTURE_EXCHANGE="0")
    subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                   capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    seen = sorted((tmp_path / f"rank{r}").read_text().split() for r in range(2))
    assert [s[:4] for s in seen] == [["0", "0", "2", "127.0.0.1"], ["1", "1", "2", "127.0.0.1"]]
    assert seen[0][4] == seen[1][4] and int(seen[0][4]) > 1024
