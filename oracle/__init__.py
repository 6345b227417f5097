"""CPU oracle for the LaM-SLidE second-stage sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lam_slide_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / timed CPU baseline.

Pinning status: the reference repository ships no tests, golden vectors or
fixtures for this path, so the oracle is pinned against
outputs of the reference's own Python modules imported in the build container
(``tools/make_fixtures.py`` -> ``tests/golden/*.npz``).  Two boundaries stay
"parity unpinned" by the reference itself: the ``torchdiffeq`` fixed-grid Euler
solver (third-party, un-vendored, version unpinned; restated from its published
algorithm) and the Lightning harness ``lightning_base.py`` (cannot be imported
here; restated from source lines cited in ``oracle/harness.py``).
"""
