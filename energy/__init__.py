"""Energy accounting entry points (``energy/collector.py`` is the CLI shim; the implementation
lives in :mod:`kvmini.telemetry.energy`).  Importable as ``energy`` for scripts written against
the reference layout (``energy/__init__.py:1-17``)."""

from kvmini.telemetry.energy import PowerSample, integrate_energy, load_power_samples, trapezoidal_wh  # noqa: F401

__all__ = ["PowerSample", "trapezoidal_wh", "load_power_samples", "integrate_energy"]
