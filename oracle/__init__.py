"""CPU oracle for the DoubleMHA speaker-embedding hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The shipped path
(``doubleattentionspeakerverification_amd``) never imports this package and
raises when its HIP library is missing.

Parity status: PINNED.  Every function here is checked in
``tests/test_oracle_golden.py`` against golden vectors captured from the real
reference (imported read-only from /root/reference in the build container by
``tests/golden/make_golden.py``; the vectors are committed under
``tests/golden/``).
"""
from .dmha_oracle import *  # noqa: F401,F403
from .train_oracle import *  # noqa: F401,F403
