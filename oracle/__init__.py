"""CPU oracle for the PAULE gradient-planning inner loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``paule_amd/`` (the product) may import
this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it -- as the checker / CPU baseline,
never as the thing shipped or measured as the product.

Two independent restatements of the reference algorithm live here:

* ``oracle.planner``  -- torch (``nn.LSTM`` + autograd + ``optim.Adam``), i.e. the
  same torch operators the reference executes (paule/models.py:326-448,
  paule/util.py:564-637, paule/paule.py:75-88, :592-776, :797, :910-1211),
  with the per-utterance batch rule of SURVEY.md section 8 (a-0).
* ``oracle.manual``   -- numpy float64, explicit BPTT (no autograd).  It is the
  math the HIP kernels implement (backward-data only, no weight gradients) and
  is checked against ``oracle.planner``.

Pinning: the reference's own tests hold no golden vectors for this path
(tests/test_paule.py:65-70 asserts nothing), so the oracle is pinned by
fixtures generated from the reference's own code objects in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
"""
