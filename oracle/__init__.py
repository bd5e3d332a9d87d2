"""TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is the checker for the HIP path: a CPU restatement
of the reference algorithm plus (in the build container only) a loader for the
reference's own Python classes.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
``pointtinybenchmark_amd`` never does.
"""
