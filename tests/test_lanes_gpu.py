"""Lane groups (include/cinema_hip.h cinema_lanes_*): the launches of independent, identically shaped sequences recorded and issued zipped as wide
launches.  A merged launch must compute exactly what the single launches compute; mismatching positions fall back to single launches in order."""

from __future__ import annotations

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import cinema_oracle as O  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from test_model_gpu import mini_kwargs, model_sizes  # noqa: E402

DEV = "cuda"


def rnd(*shape, dtype=torch.bfloat16, seed=0):  # noqa: ANN001, ANN002, ANN201
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(dtype).to(DEV)


def test_lane_group_merges_identical_sequences_and_falls_back_otherwise() -> None:
    a = [rnd(300, 256, seed=i) for i in range(3)]
    w = [rnd(128, 256, seed=10 + i) for i in range(3)]
    bias = [rnd(128, dtype=torch.float32, seed=20 + i) for i in range(3)]
    want = [K.gemm(a[i], w[i], bias=bias[i], act=1) for i in range(3)]
    want_ln = []
    for i in range(3):
        y16, _, _, _ = K.layernorm_fwd(want[i].float(), torch.ones(128, device=DEV), torch.zeros(128, device=DEV), 1e-6)
        want_ln.append(y16)
    stats0 = list(K.LANE_STATS)
    got, got_ln = [None] * 3, [None] * 3
    with K.lanes(3) as g:
        for i in range(3):
            g.select(i)
            got[i] = K.gemm(a[i], w[i], bias=bias[i], act=1)
            x32 = K.cast(got[i], torch.float32)
            got_ln[i] = K.layernorm_fwd(x32, torch.ones(128, device=DEV), torch.zeros(128, device=DEV), 1e-6)[0]
    assert K.LANE_STATS[0] - stats0[0] == 3 and K.LANE_STATS[1] - stats0[1] == 0  # gemm, cast, layernorm: one wide launch each
    for i in range(3):
        assert torch.equal(got[i], want[i]) and torch.equal(got_ln[i], want_ln[i]), i
    # split-K weight gradients with per-lane slab workspaces + reduce, through the side... main stream only here
    dy = [rnd(5000, 64, seed=30 + i) for i in range(3)]
    x = [rnd(5000, 128, seed=40 + i) for i in range(3)]
    ref = [torch.zeros(64, 128, device=DEV) for _ in range(3)]
    for i in range(3):
        K.gemm(dy[i], x[i], a_kmajor=False, b_kmajor=False, out=ref[i], accumulate=True, split_k=8)
    out = [torch.zeros(64, 128, device=DEV) for _ in range(3)]
    with K.lanes(3) as g:
        for i in range(3):
            g.select(i)
            K.gemm(dy[i], x[i], a_kmajor=False, b_kmajor=False, out=out[i], accumulate=True, split_k=8)
    for i in range(3):
        assert torch.equal(out[i], ref[i]), i
    # different shapes per lane: no merging, still correct and in order
    stats1 = list(K.LANE_STATS)
    res = [None] * 2
    with K.lanes(2) as g:
        g.select(0)
        res[0] = K.gemm(a[0], w[0])
        g.select(1)
        res[1] = K.gemm(a[1][:200], w[1])
        tail = K.cast(res[1], torch.float32)  # lane 1 is longer than lane 0: sequential fallback for the whole group
    assert K.LANE_STATS[0] == stats1[0] and K.LANE_STATS[1] - stats1[1] == 3
    assert torch.equal(res[0], K.gemm(a[0], w[0])) and torch.equal(tail, K.gemm(a[1][:200], w[1]).float())


@pytest.mark.parametrize("replay", [False, True])
def test_model_step_with_lane_groups_equals_plain_launches(replay: bool) -> None:
    """The 4-view MAE step with the three long-axis stems as a lane group (forward and backward) against the same step with plain launches:
    same loss, same gradient norm, same parameters after 3 steps (a few fp32 atomics - bias-gradient row sums - reorder between runs)."""
    from cinema_amd.optim import TrainStep

    out = {}
    for lanes_on in (False, True):
        K.LANES_ENABLED = lanes_on
        try:
            torch.manual_seed(3)
            model = CineMA(**mini_kwargs()).to(DEV)
            step = TrainStep(model, lr=1e-3, replay=replay)
            torch.manual_seed(5)
            batch = {v: torch.rand(2, 1, *s, device=DEV) for v, s in model_sizes(model).items()}
            before = list(K.LANE_STATS)
            traj = []
            for _ in range(3):
                loss, gn, _ = step(batch, 0.75)
                traj.append((float(loss), float(gn)))
            out[lanes_on] = (traj, step.flat.flat_param.clone(), K.LANE_STATS[0] - before[0])
        finally:
            K.LANES_ENABLED = True
    assert out[False][2] == 0 and out[True][2] > 50  # merged launches were really issued (forward and backward groups)
    for (la, ga), (lb, gb) in zip(out[False][0], out[True][0]):
        assert abs(la - lb) <= 1e-5 * abs(la) and abs(ga - gb) <= 1e-4 * abs(ga), (out[False][0], out[True][0])
        assert math.isfinite(la)
    assert (out[False][1] - out[True][1]).abs().max() <= 2e-3  # Adam's first steps are +-lr: identical up to sign flips of ~0 gradients


@pytest.mark.parametrize("replay", [False, True])
def test_long_axis_stream_changes_nothing_but_the_schedule(replay: bool) -> None:
    """``tape.LAX_STREAM``: the long-axis lane groups (stems, fusion, prediction heads; forward and backward) on a stream of their own beside the short-axis chain.
    Same launches, same operands, another queue: loss and gradient norm of every step and the parameters after 4 steps are BIT-identical to the two-stream
    schedule when the only reorderable sums (fp32 atomics of bias-gradient row sums) are absent, and within their reordering noise otherwise; the stream is
    really used (the forks into it are counted), and 12 replays of the recorded step keep matching the eager trajectory (a missing join shows up as a race)."""
    from cinema_amd import tape as T
    from cinema_amd.optim import TrainStep

    out = {}
    forks = {}
    real_fork = K.stream_fork
    for lax_on in (False, True):
        T.LAX_STREAM = lax_on
        n_forks = [0]

        def counting_fork(a: int, b: int) -> None:
            if T._LAX_STREAMS and b == T.lax_stream().cuda_stream:  # noqa: SLF001
                n_forks[0] += 1
            real_fork(a, b)

        K.stream_fork = counting_fork
        try:
            torch.manual_seed(3)
            model = CineMA(**mini_kwargs()).to(DEV)
            step = TrainStep(model, lr=1e-3, replay=replay)
            torch.manual_seed(5)
            batch = {v: torch.rand(2, 1, *s, device=DEV) for v, s in model_sizes(model).items()}
            traj = []
            for _ in range(12 if replay else 4):
                loss, gn, _ = step(batch, 0.75)
                traj.append((float(loss), float(gn)))
            torch.cuda.synchronize()
            out[lax_on] = (traj, step.flat.flat_param.clone())
            forks[lax_on] = n_forks[0]
        finally:
            K.stream_fork = real_fork
            T.LAX_STREAM = True
    assert forks[False] == 0 and forks[True] >= 6, forks  # eager / recording pass: forward + backward forks of the three groups (replays re-issue them from the list)
    for (la, ga), (lb, gb) in zip(out[False][0], out[True][0]):
        assert math.isfinite(la) and abs(la - lb) <= 1e-5 * abs(la) and abs(ga - gb) <= 1e-4 * abs(ga), (out[False][0], out[True][0])
    assert (out[False][1] - out[True][1]).abs().max() <= 2e-3
