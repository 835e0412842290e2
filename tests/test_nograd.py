"""The `nograd` collision metric: oracle/nograd_oracle.py restates the reference's host loop (simulator.py:1111-1149,
infractions.py:352-375, 429-500) with shapely's `intersection(...).area != 0` answered twice -- float64 convex clipping and an exact
rational predicate -- and the HIP kernel (tds_overlap_count_f32) must give the same counts.  PARITY UNPINNED against shapely itself
(absent here; no reference test exercises the metric)."""
import numpy as np
import pytest
import torch

from oracle import nograd_oracle as ng

gpu = pytest.mark.gpu


def random_scene(seed, B, A, spread=25.0, p_present=0.85):
    g = np.random.default_rng(seed)
    state = np.concatenate([g.uniform(-spread, spread, (B, A, 2)), g.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    size = (np.array([4.5, 2.0]) * g.uniform(0.9, 1.1, (B, A, 2))).astype(np.float32)
    present = g.uniform(size=(B, A)) < p_present
    return state, size, present


def special_scene():
    """axis-aligned, exactly representable: touching along an edge, touching in a corner, contained, identical, crossing without a corner
    inside (a plus sign), a degenerate (zero-width) box through another one, an absent box on top of everything"""
    boxes = [                    # x, y, length, width, psi
        (0, 0, 4, 2, 0), (4, 0, 4, 2, 0),            # 0-1 share the edge x = 2: no area
        (4, 2, 4, 2, 0),                             # 2 touches 1 along y = 1 and 0 in the corner (2, 1): no area
        (20, 0, 4, 2, 0), (20, 0, 1, 0.5, 0),        # 4 inside 3
        (40, 0, 4, 2, 0), (40, 0, 4, 2, 0),          # identical
        (60, 0, 6, 1, 0), (60, 0, 1, 6, 0),          # a plus sign: no corner of either inside the other
        (80, 0, 4, 0, 0), (80, 0, 2, 2, 0),          # a segment (width 0) through a box: no area
        (100, 0, 4, 2, 0), (101.5, 0.75, 4, 2, 0.5),  # an ordinary overlap
        (0, 0, 500, 500, 0),                         # absent: covers everything, counts nowhere
    ]
    b = np.array(boxes, np.float32)[None]
    state = np.concatenate([b[..., :2], b[..., 4:5], np.zeros_like(b[..., :1])], -1)
    present = np.ones((1, len(boxes)), bool)
    present[0, -1] = False
    return state, b[..., 2:4].copy(), present, [0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0]


def test_oracle_known_answers():
    state, size, present, expect = special_scene()
    for predicate in ('clip', 'exact'):
        np.testing.assert_array_equal(ng.nograd_collision(state, size, present, predicate)[0], expect)
    # the reference's docstring example shape: upper triangle, symmetric after mirroring, padded back to A
    st = np.array([[[0, 0, 0, 0], [2, 0, 0, 0], [30, 30, 0, 0], [1, 0.5, 0.5, 0], [4, 0, 0, 0]]], np.float32)
    sz = np.tile(np.array([4.0, 2.0], np.float32), (1, 5, 1))
    pr = np.array([[True, True, True, False, True]])
    assert ng.nograd_collision(st, sz, pr).tolist() == [[1.0, 2.0, 0.0, 0.0, 1.0]]
    r = ng.rectangle_vertices(*np.split(np.array([[1, 2, 4, 2, 0]], np.float32), 5, axis=-1))
    np.testing.assert_array_equal(r[0], [[-1, 1], [3, 1], [3, 3], [-1, 3]])        # length along the heading, counter-clockwise
    assert r.dtype == np.float32


def test_the_two_restatements_of_shapely_agree_on_random_scenes():
    state, size, present = random_scene(3, 3, 48)
    a, b = ng.nograd_collision(state, size, present, 'clip'), ng.nograd_collision(state, size, present, 'exact')
    np.testing.assert_array_equal(a, b)
    assert a.sum() > 20 and not a[~present].any()


def _sim(state, size, present):
    from test_gpu_simulator import make_sim, DEV
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.simulator import CollisionMetric
    sim = make_sim(state, size, present, BirdviewMesh.empty(batch_size=state.shape[0]).to(DEV))
    sim.cfg.collision_metric = CollisionMetric.nograd
    return sim


@gpu
@pytest.mark.parametrize('A,B,seed', [(64, 6, 21), (96, 3, 22), (200, 1, 23), (7, 9, 24)])
def test_nograd_equals_the_oracle_on_random_scenes(A, B, seed):
    state, size, present = random_scene(seed, B, A, spread={7: 5.0, 64: 25.0, 96: 25.0, 200: 45.0}[A])
    out = _sim(state, size, present).compute_collision()
    assert out.dtype == torch.float64 and out.shape == (B, A)
    clip, exact = ng.nograd_collision(state, size, present, 'clip'), ng.nograd_collision(state, size, present, 'exact')
    np.testing.assert_array_equal(clip, exact)                      # no razor-edge pair in these seeds: the comparison below is unambiguous
    np.testing.assert_array_equal(out.cpu().numpy(), exact)
    assert exact.sum() > 10 and not out.cpu().numpy()[~present].any()


@gpu
def test_nograd_touching_contained_identical_and_degenerate():
    state, size, present, expect = special_scene()
    np.testing.assert_array_equal(_sim(state, size, present).compute_collision().cpu().numpy()[0], expect)
    # NaN poses are scrubbed to the origin (as compute_collision does for the other metrics) and nothing is differentiable
    state[0, 3, 0] = np.nan
    out = _sim(state, size, present).compute_collision()
    assert torch.isfinite(out).all() and not out.requires_grad
    sim = _sim(*random_scene(5, 2, 16)[:3])
    with pytest.raises(AssertionError):
        sim.compute_collision(agent_types=['vehicle'])


@gpu
def test_nograd_scenes_agree_with_the_iou_metric_where_overlaps_are_deep():
    """consistency with K2a: an IoU above 1e-4 (an overlap far above the fp32 noise of the IoU pipeline) is always counted, and a pair
    the exact predicate rejects never has a positive IoU beyond that noise"""
    from torchdrivesim_amd.simulator import CollisionMetric
    state, size, present = random_scene(31, 4, 64)
    present[:] = True
    sim = _sim(state, size, present)
    cnt = sim.compute_collision().cpu().numpy()
    sim.cfg.collision_metric = CollisionMetric.iou
    iou = sim.compute_collision().cpu().numpy()
    assert ((iou > 1e-4) <= (cnt > 0)).all() and ((cnt == 0) <= (iou < 1e-4)).all()
