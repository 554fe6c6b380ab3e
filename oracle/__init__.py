"""CPU oracle for the SVBRDF rendering-loss hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``svbrdf_estimation_amd``) never does.

``oracle.c_oracle``     ctypes binding of ``svbrdf_oracle.c`` (plain-C restatement,
                        fp32 in the reference's op order + an fp64 evaluation)
``oracle.eager_torch``  eager-PyTorch restatement of the reference's execution
                        model (one tensor op per arithmetic step), used as the
                        timed CPU baseline in bench.py

Parity pin: fixtures in ``tests/golden/`` generated from the reference itself by
``tests/golden/make_golden.py`` (see that file's header).
"""
