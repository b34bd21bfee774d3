"""CPU oracle for the CalciumGAN WGAN-GP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``calciumgan_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / timed CPU baseline.

PARITY UNPINNED: the reference (bryanlimy/CalciumGAN) ships no tests, golden
vectors or fixtures for this path, and its arithmetic lives in TensorFlow
2.3.1 / Keras (setup.sh:26-29), which is neither vendored in /root/reference
nor installed here.  The restatement below follows the reference call sites
(cited per function) and the published TF/Keras op semantics (SURVEY.md
Appendix A); it is pinned only by the index-level known-answer tests in
``tests/test_oracle_kat.py``.
"""
from .calciumgan_oracle import *  # noqa: F401,F403
