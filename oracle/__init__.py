"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A dtype-explicit CPU restatement of the reference's self-play hot path (SURVEY.md
section 8a).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import anything from this package, and only as the checker / reported baseline.
The product (`alphapig_amd/`) never imports it and fails loudly when its native
libraries are missing.

Pinning status
  * board / planes / winner / augmentation / PUCT search / self-play / pure MCTS
    (rows a1-a5, a10-a13): PINNED -- checked bit-for-bit against golden vectors
    produced by running the reference's own code (tools/capture_golden.py ->
    tests/golden/*.npz; tests/test_oracle_golden.py).
  * network arithmetic (rows a6-a9): PARITY UNPINNED -- the arithmetic lives in
    MXNet 1.6.0 (requirements.txt:8), which is not in /root/reference and not
    installable here; the reference holds no weights and no recorded outputs.  The
    restatement follows policy_value_net_mxnet.py:41-102, policy_value_loss.json and
    the published MXNet-1.x operator definitions.
"""
