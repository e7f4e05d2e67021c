"""The command-line surface: `otgan_amd.train.build_parser()` against the flag table
extracted from the reference's train.py:14-33 (tests/golden/cli_flags.json, produced by
oracle/make_golden_cli.py which parses -- never executes -- the reference file)."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN


def _golden():
    with open(os.path.join(GOLDEN, "cli_flags.json")) as f:
        return json.load(f)["flags"]


def test_reference_flag_table_is_complete():
    flags = _golden()
    assert len(flags) == 18                                   # 18 flags
    assert [f["name"] for f in flags][:3] == ["--seed", "--batch_size", "--learning_rate_disc"]


def test_parser_has_every_reference_flag_with_same_type_default_action():
    from otgan_amd.train import build_parser
    p = build_parser()
    actions = {a.option_strings[0]: a for a in p._actions if a.option_strings}
    for f in _golden():
        a = actions.get(f["name"])
        assert a is not None, f"missing flag {f['name']} (reference train.py:{f['line']})"
        if f["action"] == "store_true":
            assert type(a).__name__ == "_StoreTrueAction", f["name"]
            assert a.default is False
            assert a.dest == (f["dest"] or f["name"].lstrip("-"))
        else:
            assert a.type is not None and a.type.__name__ == f["type"], f["name"]
            assert a.default == f["default"] and type(a.default) is type(f["default"]), (f["name"], a.default)


def test_defaults_namespace_matches_reference():
    from otgan_amd.train import build_parser
    from otgan_amd.trainer import default_args
    ns = build_parser().parse_args([])
    d = default_args()
    for f in _golden():
        key = f["dest"] or f["name"].lstrip("-")
        want = False if f["action"] == "store_true" else f["default"]
        assert getattr(ns, key) == want, key
        assert getattr(d, key) == want, key                  # trainer.default_args mirrors the same table
    # every added flag also has a trainer default
    for k in vars(ns):
        assert hasattr(d, k), k


def test_odd_shard_count_is_rejected_like_the_reference():
    # train.py:34 `assert args.nr_gpu % 2 == 0` fires before anything touches a device
    from otgan_amd import train
    with pytest.raises(AssertionError):
        train.main(["--nr_gpu", "3", "--synthetic"])


def test_load_cifar_layout(tmp_path):
    """The pickled-batch layout data/cifar10_data.py:29-53 reads: [N, 3072] uint8 rows, channel-major."""
    from otgan_amd import train
    d = tmp_path / "cifar-10-python" / "cifar-10-batches-py"
    os.makedirs(d)
    rng = np.random.RandomState(0)
    raw = []
    for i in range(1, 6):
        a = rng.randint(0, 256, size=(4, 3072)).astype(np.uint8)
        raw.append(a)
        with open(d / f"data_batch_{i}", "wb") as f:
            pickle.dump({"data": a, "labels": [0] * 4}, f)
    x = train.load_cifar(str(tmp_path))
    assert x.shape == (20, 32, 32, 3) and x.dtype == np.float32
    allraw = np.concatenate(raw).reshape(-1, 3, 32, 32)
    np.testing.assert_allclose(x[7, 5, 9], allraw[7, :, 5, 9] / 127.5 - 1.0, rtol=0, atol=1e-6)   # train.py:158
    assert x.min() >= -1.0 and x.max() <= 1.0
