"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  ``openintel_amd`` (the product)
never does; it fails loudly when ``libopenintel_hip.so`` is missing.

``oracle.lib`` wraps ``liboi_oracle.so`` (plain C, built by ``oracle/Makefile``)
with numpy in/out.  ``oracle.pyref`` is an independent pure-Python restatement
of the lexicon scorer used to cross-check the C one on small cases.

Pinning status:
  * lexicon score / social summary / fusion scalars: PINNED by the reference's
    own fixtures and test assertions (tests/golden/reference_fixture.json).
  * BM25 / cosine / top-k / RRF: PARITY UNPINNED -- the reference has no such
    code; these restate textbook definitions with builder-chosen parameters.
"""
from . import lib, pyref  # noqa: F401
