"""CPU: the C-ABI library loads and exports every symbol the headers declare, and the
ctypes binding covers all of them (no compute calls here -- no GPU in this tier)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    names = set()
    for h in ("otgan.h", "otgan_layers.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(otgan_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    from otgan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 14
    for n in sorted(decl):
        assert hasattr(L, n), f"{n} declared in include/ but not exported"


def test_binding_covers_header():
    from otgan_amd import _lib
    assert set(_lib.SIGNATURES) == _declared()


def test_version_and_workspace_queries():
    from otgan_amd import _lib
    L = _lib.lib()
    assert L.otgan_version() == 1
    assert L.otgan_matching_workspace_bytes(0, 128, 32768) > 6 * 128 * 128 * 4 * 3
    assert L.otgan_matching_workspace_bytes(1, 256, 7296) > 0
    assert L.otgan_matching_workspace_bytes(0, 0, 16) == 0


def test_no_cpu_fallback():
    import pytest
    import torch
    from otgan_amd import _lib
    from otgan_amd.utils import matching
    x = torch.zeros(4, 8)
    with pytest.raises(_lib.OtganError):
        matching.get_matched_features([x, x], [x, x], 1.0, 1)


def test_operator_argument_errors():
    """Contract violations of the operator API raise Python exceptions like the reference's
    (an odd shard count is the reference's `assert args.nr_gpu % 2 == 0`, train.py:34)."""
    import pytest
    import torch
    from otgan_amd.utils import matching
    with pytest.raises(ValueError):
        matching.get_matched_features([], [], 1.0, 1)                       # empty lists
    x = torch.zeros(4, 8)
    with pytest.raises(ValueError):
        matching.get_matched_features([x, x], [x], 1.0, 1)                  # list length mismatch


def test_every_switch_the_sources_read_is_known_and_dead_ones_are_reported():
    """A/B tools kept setting switches that the round-5 prune had removed and compared identical configurations.
    `_lib.KNOWN_SWITCHES` lists every OTGAN_* variable the build reads -- checked here against the sources (getenv in csrc/,
    os.environ in the package and bench.py) -- and loading the library warns once about any other OTGAN_* name."""
    import glob
    import re
    from otgan_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for path in glob.glob(os.path.join(root, "ot-gan_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "ot-gan_amd", "csrc", "*.inc")):
        read |= set(re.findall(r'getenv\("(OTGAN_[A-Z0-9_]+)"\)', open(path).read()))
    for path in (glob.glob(os.path.join(root, "ot-gan_amd", "*.py")) + glob.glob(os.path.join(root, "ot-gan_amd", "*", "*.py")) +
                 [os.path.join(root, "bench.py")]):
        src = open(path).read()
        read |= set(re.findall(r'environ(?:\.get|\.setdefault|\.pop)?[\(\[]\s*"(OTGAN_[A-Z0-9_]+)"', src))
    assert read, "no switch found: the scan is broken"
    assert read <= _lib.KNOWN_SWITCHES, sorted(read - _lib.KNOWN_SWITCHES)
    host_side = {s for s in read if s in ("OTGAN_LIB_PATH", "OTGAN_DIST_BACKEND", "OTGAN_FORCE_COLLECTIVES", "OTGAN_SINGLE_DEVICE",
                                          "OTGAN_COLLECTIVES", "OTGAN_SIDE_STREAM", "OTGAN_STEP_GRAPH")}
    assert len(host_side) <= 10
    assert _lib.unknown_switches({"OTGAN_DENSE_SPLIT": "0", "OTGAN_SIDE_STREAM": "0", "PATH": "x"}) == ["OTGAN_DENSE_SPLIT"]
