"""CPU oracle for the GOPS ADP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `gops_amd/` imports this package; it exists so that
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` can check (never
replace) the HIP path.  Pinned against the unmodified reference through `tests/golden/`.
"""
