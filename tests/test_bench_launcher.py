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


def test_ranks_that_share_a_device_run_without_the_sdma_engines(tmp_path):
    """Round 6 (profiles/r06_dry_run_loop.md): eight processes on ONE device trip over the platform's SDMA copy-engine path (a rank lost to
    HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in 18 of 70 launches, 0 of 30 with the engines off) -- the rehearsal's ranks are started with
    HSA_ENABLE_SDMA=0, the ranks of a real multi-GPU launch (a device and its engines each) are not, and a caller's own setting is kept."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the rehearsal itself is the test (tests/test_data_parallel_gpu.py)")
    hook = tmp_path / "sitecustomize.py"
    hook.write_text(
        "import os, sys\n"
        "if os.environ.get('WORLD_SIZE') and sys.argv and sys.argv[0].endswith('bench.py'):\n"
        "    open(os.path.join(os.environ['PCRL_TEST_DIR'], 'rank' + os.environ['RANK']), 'w').write(os.environ.get('HSA_ENABLE_SDMA', 'unset'))\n")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_SDMA")}
    for name, argv, extra, want in (("rehearsal", ["--dry-run-ranks", "2"], {}, "0"), ("real", ["--gpus", "2"], {}, "unset"),
                                    ("share", ["--gpus", "2", "--share-gpu", "--backend", "gloo"], {}, "0"),
                                    ("kept", ["--dry-run-ranks", "2"], {"HSA_ENABLE_SDMA": "1"}, "1")):
        d = tmp_path / name
        d.mkdir()
        env = dict(base, PYTHONPATH=str(tmp_path) + os.pathsep + base.get("PYTHONPATH", ""), PCRL_TEST_DIR=str(d), **extra)
        subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv + ["--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert sorted((d / f"rank{r}").read_text() for r in range(2)) == [want, want], name


def test_the_launcher_retries_only_on_the_dedicated_signal(tmp_path):
    """A failed first attempt is repeated with PCRL_CAPTURE_EXCHANGE=0 ONLY when a rank left with RC_EXCHANGE_IN_GRAPH (the step's
    captured all-reduces failed while replaying); an ordinary failure (here: no GPU -> SystemExit, rc 1) is reported once, with no
    second attempt and no note blaming the captured exchange."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the two-rank run")
    hook = tmp_path / "sitecustomize.py"
    hook.write_text(
        "import os, sys\n"
        "if os.environ.get('WORLD_SIZE') and sys.argv and sys.argv[0].endswith('bench.py'):\n"
        "    d = os.environ['PCRL_TEST_DIR']\n"
        "    open(os.path.join(d, 'attempt_%s_cap%s_rank%s' % (len(os.listdir(d)), os.environ.get('PCRL_CAPTURE_EXCHANGE', '1'), os.environ['RANK'])), 'w').close()\n"
        "    if os.environ.get('PCRL_TEST_RC') and os.environ.get('PCRL_CAPTURE_EXCHANGE', '1') != '0':\n"
        "        os._exit(int(os.environ['PCRL_TEST_RC']))\n")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PCRL_CAPTURE_EXCHANGE")}
    for rc, attempts in ((None, 1), ("75", 2), ("1", 1)):
        d = tmp_path / f"run_{rc}"
        d.mkdir()
        env = dict(base, PYTHONPATH=str(tmp_path) + os.pathsep + base.get("PYTHONPATH", ""), PCRL_TEST_DIR=str(d))
        if rc:
            env["PCRL_TEST_RC"] = rc
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                             capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert out.returncode == 1                                   # no GPU here: every attempt ends in a failure
        caps = sorted({f.split("_")[2] for f in os.listdir(d)})
        assert caps == (["cap0", "cap1"] if attempts == 2 else ["cap1"]), (rc, sorted(os.listdir(d)))
        assert ("once more with PCRL_CAPTURE_EXCHANGE=0" in out.stderr) == (attempts == 2), out.stderr[-800:]


def test_which_errors_count_as_a_failure_of_the_captured_exchange(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("PCRL_CAPTURE_EXCHANGE", raising=False)
    yes = [RuntimeError("update step did not publish its metrics within PCRL_STEP_TIMEOUT_S (90 s): a kernel hangs or a peer of a collective is gone"),
           RuntimeError("NCCL error in: ProcessGroupNCCL.cpp, unhandled cuda error"), RuntimeError("HIP error: operation not permitted when stream is capturing")]
    no = [RuntimeError("HIP out of memory. Tried to allocate 2.00 GiB (captured graph pool)"), AssertionError("batch must divide over the ranks"),
          ValueError("graph"), RuntimeError("pcrl error -2: bad argument")]
    assert all(bench.exchange_in_graph_failure(e) for e in yes)
    assert not any(bench.exchange_in_graph_failure(e) for e in no)
    monkeypatch.setenv("PCRL_CAPTURE_EXCHANGE", "0")
    assert not any(bench.exchange_in_graph_failure(e) for e in yes)      # the segmented schedule has nothing left to fall back to
    monkeypatch.delenv("PCRL_CAPTURE_EXCHANGE")
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert not any(bench.exchange_in_graph_failure(e) for e in yes)


def test_only_capture_and_collective_failures_select_the_segmented_schedule():
    """methods/sac.py::_is_capture_error decides whether a failed capture of the EXCHANGING step falls back to per-segment graphs: stream-
    capture refusals and collective failures inside the capture do; out of memory, assertions, this library's argument errors and
    unrelated messages that merely mention a capture do not."""
    from pointcloud_rl_amd.methods.sac import _is_capture_error
    yes = ["HIP error: operation not permitted when stream is capturing", "hipErrorStreamCaptureUnsupported: operation not permitted",
           "CUDA error: operation failed due to a previous error during capture", "NCCL error in: ProcessGroupNCCL.cpp:1970, unhandled cuda error",
           "ncclSystemError: System call (e.g. socket, malloc) or external library call failed", "RCCL error: invalid usage",
           "hipErrorStreamCaptureInvalidated"]
    no = ["HIP out of memory while capturing stream", "pcrl_gemm_group_f32: PCRL_E_ARG 1 <= n <= 4 problems (capture)", "invalid argument",
          "the screen capture tool failed", "shape mismatch"]
    for text in yes:
        assert _is_capture_error(RuntimeError(text)), text
    for text in no:
        assert not _is_capture_error(RuntimeError(text)), text
    assert not _is_capture_error(AssertionError("stream is capturing"))


def test_the_fused_optimizer_survives_deepcopy_and_pickle(monkeypatch):
    """torch's Optimizer.__getstate__ keeps defaults / state / param_groups only; HipAdam's flat buffers, moments and device step count
    travel with it.  Plain tensor bookkeeping: runs without a GPU."""
    import copy
    import pickle
    import torch
    import pointcloud_rl_amd.hip as hip
    from pointcloud_rl_amd.methods.sac import FlatBuffer, HipAdam
    monkeypatch.setattr(hip, "adam_workspace_bytes", lambda n: 64)
    p1, p2 = torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(3, 2))
    opt = HipAdam(FlatBuffer([("a", p1), ("b", p2)]), lr=3e-4)
    opt.exp_avg.fill_(2.0)
    opt.exp_avg_sq.fill_(0.5)
    opt.step_counter.fill_(7)
    for clone in (copy.deepcopy(opt), pickle.loads(pickle.dumps(opt))):
        assert isinstance(clone, HipAdam) and len(clone.param_groups) == 2 and clone.param_groups[0]["lr"] == 3e-4
        assert torch.equal(clone.exp_avg, opt.exp_avg) and torch.equal(clone.exp_avg_sq, opt.exp_avg_sq) and int(clone.step_counter) == 7
        assert clone.flat is not opt.flat and torch.equal(clone.flat.data, opt.flat.data)
        sd = clone.state_dict()
        assert float(sd["state"][0]["step"]) == 7.0 and torch.equal(sd["state"][1]["exp_avg"], torch.full((3, 2), 2.0))
