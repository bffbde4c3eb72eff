"""CPU oracle for the PPOV2.0/2.1 hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (numpy / torch-CPU) of the reference
algorithm (su1phurd/UAV-WRF-LES-PPO-LSTM, PPOV1.1 / PPOV2.0 / PPOV2.1 `environment.py`,
`model.py`, `train_ppo2.0.py`).  It exists to CHECK the HIP product path:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
    may import it;
  * the product package (`uav-wrf-les-ppo-lstm_amd/`) never imports it and has no CPU
    fallback -- it raises if `libuavppo.so` is missing.

Parity pin: every function here is checked against golden vectors produced by the
reference itself (imported in the build container by `oracle/gen_golden.py`; fixtures
committed under `tests/golden/`).  See DESIGN.md "Oracle".
"""
