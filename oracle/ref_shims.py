"""TEST INFRASTRUCTURE (this container only): import shims that let the compiled,
unmodified reference cparsers (built by oracle/build_reference.sh into /tmp) and the
reference's own parsers.py run under Python 3.  Used only by tests/golden/make_golden.py
and by the optional cross-check tests that skip when /root/reference is absent.

Shims: itertools.izip, a stand-in `core` module exposing
Segment with the attribute surface of core.py:115-223 (the real one calls
kwargs.iteritems()).  Nothing here is on the product path.
"""
import importlib.util
import itertools
import os
import sys
import types

import numpy as np

ORACLE_DIR = os.environ.get("PYPORE_ORACLE_DIR", "/tmp/pypore_oracle")
REFERENCE = "/root/reference"


def available():
    return os.path.isdir(REFERENCE) and any(
        f.startswith("cparsers.") and f.endswith(".so") for f in os.listdir(ORACLE_DIR)
    ) if os.path.isdir(ORACLE_DIR) else False


def load_cparsers():
    itertools.izip = zip
    if "core" not in sys.modules:
        core = types.ModuleType("core")

        class Segment(object):
            def __init__(self, current, **kw):
                self.current = current
                for k, v in kw.items():
                    if hasattr(self, k):
                        continue
                    try:
                        setattr(self, k, v)
                    except AttributeError:
                        pass
            mean = property(lambda s: np.mean(s.current))
            std = property(lambda s: np.std(s.current))
            min = property(lambda s: np.min(s.current))
            max = property(lambda s: np.max(s.current))
            n = property(lambda s: len(s.current))

        core.Segment = Segment
        core.__all__ = ["Segment"]
        sys.modules["core"] = core
    if ORACLE_DIR not in sys.path:
        sys.path.insert(0, ORACLE_DIR)
    import cparsers
    return cparsers


def load_calignment():
    """The compiled, unmodified reference calignment (cSegmentAligner)."""
    if ORACLE_DIR not in sys.path:
        sys.path.insert(0, ORACLE_DIR)
    import calignment
    return calignment


def load_reference_parsers():
    """The reference's own parsers.py (lambda_event_parser, SpeedyStatSplit wrapper) under Py3."""
    cparsers = load_cparsers()
    sys.dont_write_bytecode = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import PyPore
    sys.modules["PyPore.cparsers"] = cparsers
    PyPore.cparsers = cparsers
    spec = importlib.util.spec_from_file_location("PyPore.parsers", REFERENCE + "/PyPore/parsers.py")
    parsers = importlib.util.module_from_spec(spec)
    sys.modules["PyPore.parsers"] = parsers
    spec.loader.exec_module(parsers)
    return parsers
