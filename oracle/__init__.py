"""TEST INFRASTRUCTURE ONLY — CPU oracle (restatement of the reference's algorithms).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker — never as the thing measured or shipped.
The product package ``openvis_amd`` must not import it (tests/test_no_oracle_in_product.py).
"""
