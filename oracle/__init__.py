"""CPU oracle for the NSF / NPE hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``sbi_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker / reported CPU baseline, never as the product path.

PARITY UNPINNED at the nflows boundary: the arithmetic of this path lives in
the third-party package ``nflows==0.14`` (pinned in /root/reference/uv.lock:2781,
declared pyproject.toml:36) which is neither vendored in /root/reference nor
installable here (no network).  ``nsf_oracle.py`` restates nflows' published
algorithm; the pieces of the path that DO live in the reference tree
(z-scoring, masks, searchsorted, shape handling, simulators) are pinned against
the real reference code through ``tests/golden/`` (see tools/make_golden.py).
"""
