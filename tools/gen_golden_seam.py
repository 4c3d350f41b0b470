"""Generate tests/golden/ref_seam_batch_drq_maniskill.npz (build container only): the batch the REFERENCE's own replay container hands to
the bound agent's step -- pushes into pyrl's ReplayMemory, `sample()` -> `GDict.to_torch()` -> `agent._fetcher(memory)()` -- for
configs/mfrl/drq/maniskill/pn_jitter.py (xyz f32, rgb u8, seg bool, agent f32; 8 transitions of 96 points).  Written by
tests/_integration_probe.py::replay_seam; tests/test_update_step_gpu.py feeds exactly this structure to update_parameters on the GPU.

    python tools/gen_golden_seam.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_seam_batch_drq_maniskill.npz")

if __name__ == "__main__":
    env = dict(os.environ, PCRL_SEAM_FIXTURE=OUT, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_integration_probe.py")], check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL)
    print(OUT, f"{os.path.getsize(OUT) / 1e3:.1f} KB")
