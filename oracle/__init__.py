"""CPU oracle for the speaker/follower hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's hot path
(tasks/R2R/model.py modules plus the per-step agent glue in follower.py /
speaker.py and the feature assembly in env.py).  It exists to *check* the HIP
implementation and to provide the `cpu_baseline` leg of bench.py.

Rules (enforced by tests/test_layout.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here;
  * nothing under speaker_follower_amd/ imports it -- the product path fails
    loudly when the HIP library is missing, it never falls back to this code.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
modules themselves, generated in the build container by
tests/golden/make_golden.py (which imports /root/reference/tasks/R2R/model.py
on torch-CPU fp32) and committed as tests/golden/*.npz.

  np_env.py     env.py / follower.py host helpers (loc embeddings, action
                embeddings, instruction batching)            -- numpy
  np_model.py   literal forward restatement of model.py + agent glue -- numpy fp32
  torch_ref.py  the same algorithm as differentiable torch-CPU functions
                (forward + autograd backward), for gradient parity
  rng.py        the counter-based dropout mask generator the HIP kernels use
"""
