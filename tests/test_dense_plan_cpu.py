"""Host logic of the dense-block split (ops._split_block_plan, _input_row_order): which wide convolutions a block is cut
into, which earlier outputs stay on each layer's own chain, and the row permutation between the reference's
per-list-element channel order and the single-tensor order of the kernels.  No GPU: the library only answers
otgan_conv2d_filter_bytes (host code)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from otgan_amd import ops  # noqa: E402

CPU = torch.device("cpu")


def test_row_order_is_the_reference_interleave():
    # reference nn.py:198-200: crelu of a list = [x0, -x0, x1, -x1, ...]; the kernels read [x0 x1 ..., -x0 -x1 ...]
    order = ops._input_row_order((3, 2), ops.ACT["crelu"], CPU).tolist()
    assert order == [0, 1, 2, 6, 7, 3, 4, 5, 8, 9]
    assert ops._input_row_order((8,), ops.ACT["crelu"], CPU) is None          # single tensor: already [x, -x]
    assert ops._input_row_order((3, 2), ops.ACT["elu"], CPU) is None           # no doubling: identity
    back = ops._row_order_back(ops._input_row_order((3, 2), ops.ACT["crelu"], CPU)).tolist()
    assert [back[o] for o in order] == list(range(10))          # back[order[i]] == i


@pytest.mark.parametrize("C0,segs0,H", [(224, (208, 16), 32), (160, (144, 16), 16), (144, (144,), 16), (200, (200,), 8), (32, (32,), 32)])
def test_densenet_blocks_are_cut_into_input_and_halves(C0, segs0, H, monkeypatch):
    monkeypatch.delenv("OTGAN_DISABLE_WINOGRAD", raising=False)
    L, F = 16, 16
    plan = ops._split_block_plan(256, H, H, C0, L, F, segs0, ops.ACT["crelu"], CPU)
    assert plan is not None and len(plan["wide"]) == 2
    w_in, w_half = plan["wide"]
    assert (w_in["C"], w_in["d0"], w_in["nrows"], w_in["after"], w_in["accumulate"]) == (C0, 0, 2 * C0, -1, 0)
    assert (w_half["C"], w_half["d0"], w_half["nrows"], w_half["after"], w_half["accumulate"]) == (8 * F, 8, 2 * 8 * F, 7, 1)
    assert w_half["x_off"] == C0 and w_half["row0"] == 2 * C0
    assert w_half["desc"].y_coff == C0 + 8 * F and w_half["desc"].Cout == 8 * F and w_in["desc"].Cout == L * F
    # every (earlier output j, layer k) pair is covered exactly once: by a wide convolution or by the layer's own chain
    for k in range(L):
        via_wide = set()
        for wd in plan["wide"][1:]:
            if wd["d0"] <= k:
                via_wide |= set(range((wd["x_off"] - C0) // F, (wd["x_off"] - C0) // F + wd["C"] // F))
        own = set(range(plan["g0"][k], k))
        assert via_wide.isdisjoint(own) and via_wide | own == set(range(k))
        assert plan["own_len"][k] == len(own) and plan["own_row0"][k] == 2 * (C0 + plan["g0"][k] * F)


def test_blocks_that_are_not_cut(monkeypatch):
    act = ops.ACT["crelu"]
    assert ops._split_block_plan(8, 8, 8, 32, 2, 16, (32,), act, CPU) is None            # 2 layers: 32 columns, too narrow
    assert ops._split_block_plan(8, 8, 8, 32, 16, 12, (32,), act, CPU) is None           # growth rate other than 16
    assert ops._split_block_plan(8, 8, 8, 30, 16, 16, (22, 8), act, CPU) is None         # list widths not multiples of 4
    assert ops._split_block_plan(8, 6, 6, 32, 16, 16, (32,), act, CPU) is None           # 6x6 is not tiled by 4x4
    monkeypatch.setattr(ops, "DENSE_SPLIT", False)
    assert ops._split_block_plan(8, 8, 8, 32, 16, 16, (32,), act, CPU) is None
    monkeypatch.setattr(ops, "DENSE_SPLIT", True)
    monkeypatch.setattr(ops, "DENSE_GROUP", 0)                                           # block input only
    plan = ops._split_block_plan(8, 8, 8, 32, 16, 16, (32,), act, CPU)
    assert len(plan["wide"]) == 1 and plan["g0"] == [0] * 16


def test_long_blocks_keep_the_fp32_growth_kernels(monkeypatch):
    """the two-scaled-fp16-piece chain path prepares at most 16 chain layers per library call and a chain call
    takes at most 17 slices; a block with more (layers_per_block = 32 at the default grouping: 30 own-chain layers) must not
    select it -- it used to raise OtganError in the forward pass."""
    act = ops.ACT["crelu"]
    p16 = ops._split_block_plan(8, 16, 16, 32, 16, 16, (32,), act, CPU)
    p32 = ops._split_block_plan(8, 16, 16, 32, 32, 16, (32,), act, CPU)
    assert p16 is not None and p32 is not None
    assert sum(1 for n in p16["own_len"] if n) == 14 and sum(1 for n in p32["own_len"] if n) == 30
    assert p32["h2"] is False
    # (whether the 16-layer block takes the fp16 chain kernels is the library's answer: otgan_dense16_h2_ok)
    monkeypatch.setattr(ops, "DENSE_GROUP", 0)                # block input only: one group of 20 slices > 17 per chain call
    p20 = ops._split_block_plan(8, 16, 16, 32, 20, 16, (32,), act, CPU)
    assert p20 is not None and len(p20["wide"]) == 1 and p20["h2"] is False
