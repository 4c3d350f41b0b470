"""CPU oracle for the point-cloud actor-critic hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``pointcloud_rl_amd/`` does.  See ``pcrl_oracle.c`` (scalar C
restatement of the kernels' arithmetic) and ``torch_ref.py`` (op-for-op PyTorch-CPU restatement
of the reference modules and update step, also the timed CPU baseline).
"""
