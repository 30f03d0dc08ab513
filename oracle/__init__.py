"""TEST INFRASTRUCTURE ONLY: parity checker for the rt355 render path.

Nothing in ``raytracer_amd/`` imports this package.  ``cpu_oracle`` is the FP64 C++ restatement
of the reference path (pinned by tests/golden/, which ``ref_harness`` + ``gen_golden`` produce
from the reference itself); ``ref_harness`` needs ``/root/reference`` and so only runs in the
build container.
"""
