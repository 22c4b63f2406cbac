"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's CUR retrieval path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported baseline.  The product (``anncur_amd``) never
imports this package and fails loudly when its HIP library is missing.
"""
