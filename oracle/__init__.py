"""CPU oracle for the CifCaf decode path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``openpifpaf_amd``) never does.

* :mod:`oracle.port`       -- ctypes binding of ``libcifcaf_oracle.so`` (the plain
  C++ restatement in ``cifcaf_oracle.cpp``).
* :mod:`oracle.reference`  -- loader for ``_ref/openpifpaf_ref.so`` (the REAL
  reference sources compiled by ``build_ref.py``), used to pin the restatement
  and as the ``"reference"`` CPU baseline.
"""
