"""CPU oracle for the xeofs EOF / randomized-SVD hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``xeofs_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.
"""
