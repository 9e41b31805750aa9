"""Minimal `forge` stand-in (see flags.py / experiment_tools.py for provenance)."""
from . import flags
from . import experiment_tools


def config(argv=None):
    return flags.parse(argv)
