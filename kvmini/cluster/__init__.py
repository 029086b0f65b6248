"""Pluggable cluster backends.

The reference shells out to ``kubectl`` from every stage (``analyze.py:362-376``,
``cost_estimator.py:159-198``, ``bench.sh:186-195``) and is tested by putting a fake
``kubectl`` script first on ``PATH`` (``.github/workflows/repro-smoke.yml:93-101``).
Here every stage talks to a :class:`ClusterBackend`; three implementations:

* :class:`KubectlBackend` — a real cluster through ``kubectl``.
* :class:`LocalBackend`   — servers started on this host by ``kvmini serve`` /
  ``kvmini deploy --backend local``; "pods" are process records with start times,
  "nodes" carry ``amd.com/gpu.*`` labels read from amd-smi.
* :class:`FakeBackend`    — in-memory / JSON-file cluster for tests and CPU CI.
"""

from .backends import (  # noqa: F401
    ClusterBackend,
    FakeBackend,
    KubectlBackend,
    LocalBackend,
    get_backend,
    iso_utc,
    make_pod,
    parse_k8s_time,
)
