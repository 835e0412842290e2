"""
Parity of the HIP kernels (through the C ABI, torchdrivesim_amd._ops) against the CPU oracle on identical inputs and
against the committed golden vectors.  Needs a real MI355X: run with `-m gpu`.
Bars (BASELINE.json north_star): bit-exact masks / pixels / integer outputs; <= 1e-5 rel on fp32 kinematic state.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    from torchdrivesim_amd import _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def sc_np(t):
    return t.detach().cpu().numpy()


def boxes_of(state, size):
    return np.concatenate([state[..., :2], size, state[..., 2:3]], -1)


LEVELS = dict(direction=2, vehicle=4, left_lane=12, joint_lane=13, right_lane=14, road=15)
COLORS = dict(road=(155, 155, 155), vehicle=(32, 74, 135), left_lane=(80, 127, 86), right_lane=(128, 0, 128),
              joint_lane=(255, 255, 255), direction=(100, 255, 255))
LEVEL_TABLE = sorted(set(float(v) for v in LEVELS.values()), reverse=True)


def pack(rgb):
    return (rgb[0] << 16) | (rgb[1] << 8) | rgb[2]


def make_map(ops, verts, faces, vert_category, categories, render=True):
    verts, faces = np.asarray(verts, np.float32), np.asarray(faces, np.int32)
    if not render or len(faces) == 0:
        return ops.StaticMap(verts, faces, None if not render else np.zeros(0, np.float32), None if not render else np.zeros(0, np.uint32),
                             LEVEL_TABLE if render else None, device=DEV)
    cat = np.asarray(vert_category)[faces[:, 0]]
    fz = np.array([LEVELS[categories[c]] for c in cat], np.float32)
    frgb = np.array([pack(COLORS[categories[c]]) for c in cat], np.uint32)
    return ops.StaticMap(verts, faces, fz, frgb, LEVEL_TABLE, device=DEV)


def actor_keys(smap, B, N):
    kb = (smap.rank_of(LEVELS['vehicle']) << 24) | pack(COLORS['vehicle'])
    kd = (smap.rank_of(LEVELS['direction']) << 24) | pack(COLORS['direction'])
    return torch.tensor([kb, kd], dtype=torch.int64).to(torch.int32).expand(B, N, 2).contiguous().to(DEV)


@pytest.fixture(scope='module')
def town():
    t = load_golden('town01_mesh.npz')
    return dict(verts=t['verts'], faces=t['faces'], vert_category=t['vert_category'], categories=[str(c) for c in t['categories']])


# ---------------------------------------------------------------------------------------------------- K1
def test_k1_bicycle_matches_oracle_and_golden(ops, oracle):
    g = load_golden('g1_kinematic.npz')
    s, a, lr = dev(g['state']), dev(g['action']), dev(g['lr'])
    rel = lambda x, y: np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-3))
    for kw, key in ((dict(), 'out_bicycle'), (dict(left_handed=True), 'out_bicycle_lh'), (dict(dt=0.25), 'out_bicycle_dt'),
                    (dict(no_reversing=True), 'out_norev')):
        out = ops.bicycle_step(s, a, lr, **kw).cpu().numpy()
        assert rel(out, g[key]) <= 1e-5                                       # vs the reference itself
        assert rel(out, oracle.bicycle_step(g['state'], g['action'], g['lr'], **kw)) <= 1e-5
    assert rel(ops.simple_step(s, dev(g['action4'])).cpu().numpy(), g['out_simple']) <= 1e-5
    assert rel(ops.simple_step(s, dev(g['action4']), oriented=True).cpu().numpy(), g['out_oriented']) <= 1e-5


@pytest.mark.parametrize('name,cls,kw', [('disp', 'BicycleByDisplacement', {}), ('disp_max5', 'BicycleByDisplacement', dict(max_dx=5, dt=0.2)),
                                         ('oriented', 'BicycleByOrientedDisplacement', {})])
def test_k1_bicycle_by_displacement_matches_golden(oracle, name, cls, kw):
    """BicycleByDisplacement / BicycleByOrientedDisplacement over K1 (kinematic.py:526-587) against the reference's outputs (G1b)"""
    from torchdrivesim_amd import kinematic
    g = load_golden('g1b_bicycle_displacement.npz')
    rel = lambda x, y: np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-3))
    m = getattr(kinematic, cls)(**kw)
    m.set_params(lr=dev(g['lr']))
    m.set_state(dev(g['state']))
    assert np.abs(m.fit_action(dev(g['future'])).cpu().numpy() - g[f'fit_{name}']).max() <= 1e-5
    m.step(dev(g['action']))
    assert rel(m.get_state().cpu().numpy(), g[f'out_{name}']) <= 1e-5
    okw = dict(max_dx=float(kw.get('max_dx', 20)), model_dt=kw.get('dt', 0.1), oriented=cls.endswith('OrientedDisplacement'))
    assert rel(m.get_state().cpu().numpy(), oracle.bicycle_by_displacement_step(g['state'], g['action'], g['lr'], **okw)) <= 1e-5
    # second step (dt differs from the model's own) from the REFERENCE's first-step state: the fitted velocity (x + dx dt - x) / dt
    # cancels catastrophically at |x| ~ 300 m, so a 1-ulp difference of the first step would show as 1e-4 in the second (the reference's
    # own conditioning, not a kernel property)
    # With dt = 0.05 one ulp of x (3e-5 at 300 m) in the target is 6e-4 m/s in the fitted speed and, through the fitted steering, 3e-5 rad
    # in the heading: that bounds what ANY two float32 evaluations of these models can agree to (the oriented variant's rotation already
    # differs by an ulp of sin / cos between devices).  Positions keep the 1e-5 bar.
    m.set_state(dev(g[f'out_{name}']))
    m.step(dev(np.ascontiguousarray(g['action'][:, ::-1])), dt=0.05)
    two, ref2 = m.get_state().cpu().numpy(), g[f'out_{name}_2steps']
    assert rel(two[..., :2], ref2[..., :2]) <= 1e-5
    ulp_x = float(np.spacing(np.float32(np.abs(g['state'][..., :2]).max())))
    assert np.abs(two[..., 3] - ref2[..., 3]).max() <= 4 * ulp_x / 0.05 and np.abs(two[..., 2] - ref2[..., 2]).max() <= 4 * ulp_x
    other = m.copy()
    assert type(other) is type(m) and other.max_dx == m.max_dx and torch.equal(other.get_state(), m.get_state())
    # gradients flow through the fitted action and the K1 step
    st = dev(g['state']).requires_grad_(True)
    act = dev(g['action']).requires_grad_(True)
    m.set_state(st)
    m.step(act)
    m.get_state()[..., :2].sum().backward()
    # (entry [0, 0] has zero displacement: d sqrt(vx^2 + vy^2) is undefined there, NaN in the reference's autograd as well)
    grad = act.grad.flatten(0, 1)[1:]
    assert torch.isfinite(grad).all() and grad.abs().sum() > 0


def test_k1_large_random_and_state_not_mutated(ops, oracle):
    gen = torch.Generator().manual_seed(5)
    n = (257, 64)
    state = torch.cat([(torch.rand(*n, 2, generator=gen) - 0.5) * 800, (torch.rand(*n, 1, generator=gen) - 0.5) * 12,
                       (torch.rand(*n, 1, generator=gen) - 0.2) * 20], -1)
    action = (torch.rand(*n, 2, generator=gen) - 0.5) * 2
    lr = 1 + torch.rand(*n, generator=gen)
    sd = state.to(DEV)
    before = sd.clone()
    out = ops.bicycle_step(sd, action.to(DEV), lr.to(DEV))
    assert torch.equal(sd, before) and out.data_ptr() != sd.data_ptr()
    ref = oracle.bicycle_step(state.numpy(), action.numpy(), lr.numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def torch_bicycle(state, action, lr, dt=0.1, lh=False, norev=False):
    nf = torch.tensor([5.0, np.pi / 2], dtype=torch.float32)
    act = action * nf
    a, beta = act[..., 0], act[..., 1]
    x, y, psi, v = state.unbind(-1)
    if norev:
        a = torch.where(v + a * dt < 0, -v / dt, a)
    if lh:
        beta = -beta
    v = v + a * dt
    x = x + v * torch.cos(psi + beta) * dt
    y = y + v * torch.sin(psi + beta) * dt
    psi = psi + (v / lr) * torch.sin(beta) * dt
    return torch.stack([x, y, psi, v], -1)


@pytest.mark.parametrize('lh,norev', [(False, False), (True, False), (False, True)])
def test_k1_backward_matches_torch_autograd(ops, lh, norev):
    gen = torch.Generator().manual_seed(9)
    state = torch.cat([(torch.rand(6, 9, 3, generator=gen) - 0.5) * 6, (torch.rand(6, 9, 1, generator=gen) - 0.3) * 4], -1)
    action = (torch.rand(6, 9, 2, generator=gen) - 0.5) * 2
    lr = 1 + torch.rand(6, 9, generator=gen)
    wgt = torch.rand(6, 9, 4, generator=gen)
    ref_in = [t.clone().requires_grad_(True) for t in (state, action, lr)]
    (torch_bicycle(*ref_in, lh=lh, norev=norev) * wgt).sum().backward()
    gpu_in = [t.to(DEV).requires_grad_(True) for t in (state, action, lr)]
    (ops.bicycle_step(*gpu_in, left_handed=lh, no_reversing=norev) * wgt.to(DEV)).sum().backward()
    for r, gq in zip(ref_in, gpu_in):
        np.testing.assert_allclose(gq.grad.cpu().numpy(), r.grad.numpy(), rtol=2e-4, atol=2e-5)


def test_k1_unicycle_closed_form(ops):
    s = torch.tensor([[1.0, 2.0, 0.5, 3.0]], device=DEV)
    out = ops.unicycle_step(s, torch.tensor([[0.4, -0.2]], device=DEV), dt=0.1, max_acc=5.0, max_yaw_rate=1.0).cpu().numpy()[0]
    v = 3.0 + 0.4 * 5 * 0.1
    np.testing.assert_allclose(out, [1 + v * np.cos(0.5) * 0.1, 2 + v * np.sin(0.5) * 0.1, 0.5 - 0.2 * 0.1, v], rtol=1e-6)


# ---------------------------------------------------------------------------------------------------- K2a
@pytest.mark.parametrize('metric', ['iou', 'discs'])
@pytest.mark.parametrize('tag', ['cur', 'rnd0', 'rnd400'])
def test_k2_pairwise_bit_exact(ops, oracle, metric, tag):
    g = load_golden('g2_boxes.npz')
    b1, b2 = dev(g[tag + '_box1']), dev(g[tag + '_box2'])
    s1, s2 = ops.metric_sc(b1, metric), ops.metric_sc(b2, metric)
    out = ops.pairwise_overlap(b1, b2, metric, s1, s2).cpu().numpy()
    fn = oracle.iou_pairs if metric == 'iou' else oracle.discs_pairs
    ref = fn(g[tag + '_box1'], g[tag + '_box2'], sc_np(s1), sc_np(s2))
    same = (out == ref) | (np.isnan(out) & np.isnan(ref))
    assert same.all(), f'{(~same).sum()} of {same.size} pairs differ, max {np.nanmax(np.abs(out - ref))}'
    # and against the reference's own numbers (its sin/cos come from torch-CPU): values to 2e-3 abs far from the origin
    # (SURVEY Q3), overlap flags exact away from the touching configurations of the curated set
    gold = g[tag + ('_iou' if metric == 'iou' else '_discs')]
    fin = np.isfinite(gold)
    np.testing.assert_allclose(out[fin], gold[fin], atol=2e-3 if tag == 'rnd400' else 2e-6, rtol=0)
    if tag != 'cur':
        assert ((out > 0) == (gold > 0)).mean() > 0.999


@pytest.mark.parametrize('num_discs', [3, 5, 7, 9, 25])
def test_k2_discs_with_other_disc_counts(ops, oracle, num_discs):
    from torchdrivesim_amd.infractions import collision_detection_with_discs
    g = load_golden('g13_discs_n.npz')
    b1, b2 = dev(g['box1']), dev(g['box2'])
    out = collision_detection_with_discs(b1, b2, num_discs=num_discs).cpu().numpy()
    s1, s2 = ops.metric_sc(b1, 'discs'), ops.metric_sc(b2, 'discs')
    ref = oracle.discs_pairs(g['box1'], g['box2'], sc_np(s1), sc_np(s2), num_discs=num_discs)
    np.testing.assert_array_equal(out, ref)                      # same [sin, cos] in, same bits out
    if num_discs != 5:
        gold = g[f'discs_{num_discs}']                           # the reference's numbers (its sin / cos come from torch-CPU)
        np.testing.assert_allclose(out, gold, atol=2e-6, rtol=0)
        assert ((out > 0) == (gold > 0)).mean() > 0.995
    with pytest.raises(AssertionError):
        collision_detection_with_discs(b1, b2, num_discs=4)
    with pytest.raises(RuntimeError):
        collision_detection_with_discs(b1, b2, num_discs=27)


def test_k2_box2corners_bit_exact(ops, oracle):
    g = load_golden('g2_boxes.npz')
    b = dev(g['rnd400_box1'])
    sc = ops.heading_sc(b[..., 4])
    np.testing.assert_array_equal(ops.box2corners(b, sc).cpu().numpy(), oracle.box2corners(g['rnd400_box1'], sc_np(sc)))


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_k2_scene_collision(ops, oracle, metric):
    g = load_golden('g2_scene_collision.npz')
    cases = [(boxes_of(g['state'], g['size']), g['present'], None, g['coll_' + metric]),
             (np.concatenate([boxes_of(g['state'], g['size']), boxes_of(g['npc_state'], g['npc_size'])], 1),
              np.concatenate([g['present'], g['npc_present']], 1), 8, g['coll_npc_' + metric]),
             (boxes_of(g['far_state'], g['far_size']), g['far_present'], None, g['far_coll_' + metric])]
    for boxes, present, nexp, gold in cases:
        bd = dev(boxes)
        sc = ops.metric_sc(torch.nan_to_num(bd, nan=0.0), metric)
        out, overlap, partner = ops.collision_forward(bd, sc, dev(present), nexp, metric, want_overlap=True, want_partner=True)
        out = out.cpu().numpy()
        ref = oracle.collision(boxes, present, n_exposed=nexp, metric=metric, sc=sc_np(sc))
        np.testing.assert_array_equal(out, ref)                          # same summation order -> bit-exact
        np.testing.assert_array_equal(out > 0, gold > 0)                 # collision mask vs the reference, bit-exact
        np.testing.assert_allclose(out, gold, atol=5e-7, rtol=0)
        # the new integer outputs are consistent with the overlaps they summarise
        ov, pa = overlap.cpu().numpy(), partner.cpu().numpy()
        assert ((ov != 0) == (pa >= 0)).all()
        assert ((out > 0) <= (ov != 0)).all()


def test_k2_collision_config2_shape_vs_oracle(ops, oracle):
    gen = np.random.default_rng(11)
    B, A = 24, 64
    centre = gen.uniform(0, 400, size=(B, 1, 2))
    xy = centre + gen.uniform(-40, 40, size=(B, A, 2))
    boxes = np.concatenate([xy, gen.uniform(4.0, 5.0, (B, A, 1)), gen.uniform(1.8, 2.2, (B, A, 1)), gen.uniform(-np.pi, np.pi, (B, A, 1))], -1).astype(np.float32)
    present = gen.uniform(size=(B, A)) < 0.9
    for metric in ('iou', 'discs'):
        bd = dev(boxes)
        sc = ops.metric_sc(bd, metric)
        out = ops.collision_forward(bd, sc, dev(present), None, metric)[0].cpu().numpy()
        ref = oracle.collision(boxes, present, metric=metric, sc=sc_np(sc))
        np.testing.assert_array_equal(out, ref)
        assert (out > 0).any() and not (out > 0).all()


# ---------------------------------------------------------------------------------------------------- K2b
def test_k2b_offroad_golden(ops, oracle):
    g = load_golden('g3_offroad.npz')
    m = make_map(ops, g['a_verts'], g['a_faces'], None, None, render=False)
    st, lw = dev(g['a_state']), dev(g['a_lenwid'])
    sc = ops.heading_sc(st[..., 2])
    for thr, key in ((0.5, 'a_off_t05'), (0.0, 'a_off_t0')):
        out = ops.offroad_forward(m, st, lw, sc, None, thr).cpu().numpy()
        np.testing.assert_array_equal(out, oracle.offroad(g['a_state'], g['a_lenwid'], g['a_verts'], g['a_faces'], thr, sc=sc_np(sc)))
        np.testing.assert_allclose(out, g[key], rtol=1e-6)
    # per-scene maps incl. collate-padded [0,0,0] faces
    for b in range(2):
        mb = make_map(ops, g['b_verts'][b], g['b_faces'][b], None, None, render=False)
        st, lw = dev(g['b_state'][b:b + 1]), dev(g['b_lenwid'][b:b + 1])
        sc = ops.heading_sc(st[..., 2])
        out = ops.offroad_forward(mb, st, lw, sc, dev(g['c_present'][b:b + 1]), 0.5).cpu().numpy()
        ref = oracle.offroad(g['b_state'][b:b + 1], g['b_lenwid'][b:b + 1], g['b_verts'][b], g['b_faces'][b], 0.5,
                             present=g['c_present'][b:b + 1], sc=sc_np(sc))
        np.testing.assert_array_equal(out, ref)
        np.testing.assert_allclose(out, g['c_sim_offroad'][b:b + 1], rtol=1e-6)


def test_k2b_offroad_town01_random(ops, oracle, town):
    m = make_map(ops, town['verts'], town['faces'], None, None, render=False)
    gen = np.random.default_rng(3)
    B, A = 4, 64
    road = town['verts'][town['vert_category'] == town['categories'].index('road')]
    xy = road[gen.integers(0, len(road), (B, A))] + gen.normal(0, 3.0, (B, A, 2))
    xy[0, :4] += 500.0                                   # far off the map: exercises the ring search to exhaustion
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    lw = np.concatenate([gen.uniform(4, 5, (B, A, 1)), gen.uniform(1.8, 2.2, (B, A, 1))], -1).astype(np.float32)
    sd = dev(state)
    sc = ops.heading_sc(sd[..., 2])
    out = ops.offroad_forward(m, sd, dev(lw), sc, None, 0.5).cpu().numpy()
    ref = oracle.offroad(state, lw, town['verts'], town['faces'], 0.5, sc=sc_np(sc))
    np.testing.assert_array_equal(out, ref)
    assert (out > 0).any() and (out == 0).any()


def test_k2b_candidate_lists_equal_the_ring_walk(ops, oracle, town, testing_lib):
    """Geometry-only maps answer the off-road query from per-cell nearest-face candidate lists; maps built without them (and points
    outside the grid) walk grid rings.  Both are the exact minimum over all faces: identical bits, at any distance from the road."""
    import os
    m = make_map(ops, town['verts'], town['faces'], None, None, render=False)
    assert m.info()['near_candidates'] > 0
    testing_lib.tds_testing_set_near_lists(0)      # this hook exists only in libtdship_testing.so
    m0 = make_map(ops, town['verts'], town['faces'], None, None, render=False)
    testing_lib.tds_testing_set_near_lists(1)
    assert m0.info()['near_candidates'] == 0
    gen = np.random.default_rng(11)
    lo, hi = town['verts'].min(0), town['verts'].max(0)
    B, A = 64, 64
    xy = gen.uniform(lo - 20.0, hi + 20.0, (B, A, 2))                   # anywhere over the map's bounding box and a little beyond
    road = town['verts'][town['vert_category'] == town['categories'].index('road')]
    xy[:16] = road[gen.integers(0, len(road), (16, A))] + gen.normal(0, 1.5, (16, A, 2))
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    state[-1, :4, :2] = [[lo[0], lo[1]], [hi[0], hi[1]], [lo[0] - 1e-3, hi[1] + 1e-3], [np.nan, 3.0]]      # grid corners, just outside, NaN
    lw = np.concatenate([gen.uniform(4, 5, (B, A, 1)), gen.uniform(1.8, 2.2, (B, A, 1))], -1).astype(np.float32)
    sd = dev(state)
    sc = ops.heading_sc(sd[..., 2])
    for thr in (0.5, 0.0, 30.0):
        a = ops.offroad_forward(m, sd, dev(lw), sc, None, thr)
        b = ops.offroad_forward(m0, sd, dev(lw), sc, None, thr)
        assert torch.equal(a, b)
    sub = slice(0, 20)
    ref = oracle.offroad(state[sub], lw[sub], town['verts'], town['faces'], 0.5, sc=sc_np(sc)[sub])
    np.testing.assert_array_equal(ops.offroad_forward(m, sd[sub], dev(lw)[sub], sc[sub], None, 0.5).cpu().numpy(), ref)
    far = ops.offroad_forward(m, sd, dev(lw), sc, None, 0.5)
    assert float(far[:-1].max()) > 40.0 ** 2                            # block centres: tens of metres from the nearest face


@pytest.mark.parametrize('cell', [1.0, 3.0, 8.0])
def test_k2b_candidate_lists_on_adversarial_meshes(ops, oracle, testing_lib, cell):
    """random triangle soups with huge faces, slivers (area < 5e-3: never "inside"), repeated vertices, collate-padded [0,0,0] faces and a
    face with a NaN vertex; queries inside the mesh, in the margin of the candidate grid and far beyond it"""
    gen = np.random.default_rng(int(cell * 10))
    for trial in range(3):
        V = 300
        verts = gen.uniform(0, 50, (V, 2)).astype(np.float32)
        faces = []
        for _ in range(200):
            a = gen.integers(0, V)
            near = np.argsort(((verts - verts[a]) ** 2).sum(1))[:12]
            faces.append([a, *gen.choice(near[1:], 2, replace=False)])
        faces += [[gen.integers(0, V), gen.integers(0, V), gen.integers(0, V)] for _ in range(6)]      # huge faces
        verts[10] = verts[11] + np.float32(1e-4)                                                            # slivers / near-degenerate edges
        faces += [[10, 11, 12], [11, 10, 11], [0, 0, 0], [0, 0, 0], [5, 5, 7]]
        if trial == 2:
            verts[20, 0] = np.nan                                                                           # never binned, never nearest
        faces = np.array(faces, np.int32)
        m = ops.StaticMap(verts, faces, device=DEV, cell_size=cell)
        testing_lib.tds_testing_set_near_lists(0)      # this hook exists only in libtdship_testing.so
        m0 = ops.StaticMap(verts, faces, device=DEV, cell_size=cell)
        testing_lib.tds_testing_set_near_lists(1)
        assert m.info()['near_candidates'] > 0 and m0.info()['near_candidates'] == 0
        B, A = 8, 64
        xy = gen.uniform(-90, 140, (B, A, 2))
        xy[:3] = gen.uniform(0, 50, (3, A, 2))
        state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
        lw = np.concatenate([gen.uniform(1, 6, (B, A, 1)), gen.uniform(1, 3, (B, A, 1))], -1).astype(np.float32)
        sd = dev(state)
        sc = ops.heading_sc(sd[..., 2])
        for thr in (0.5, 0.0):
            a = ops.offroad_forward(m, sd, dev(lw), sc, None, thr)
            assert torch.equal(a, ops.offroad_forward(m0, sd, dev(lw), sc, None, thr))
            if trial < 2:                                                                                   # the oracle has no NaN-vertex rule to compare with
                ref = oracle.offroad(state[:4], lw[:4], verts, faces, thr, sc=sc_np(sc)[:4])
                np.testing.assert_array_equal(a[:4].cpu().numpy(), ref)


def test_k2b_points_far_beyond_the_map_take_the_hierarchy(ops, oracle, town, testing_lib):
    """Agents that strayed beyond the margin the candidate lists cover (48 m around the mesh) are served by an ordered descent of a
    bounding-volume hierarchy over the faces (csrc/map.hip: nearest_face_d2_bvh; infractions.py:86-229 is the loss): the exact minimum over
    all faces, as the walk over grid rings it replaces there -- identical bits at 50 m, 1 km and 1 000 km from the map, forward and backward,
    with the oracle on a sample -- in a fraction of the time (the ring walk took 124 ms for a batch of 65 536 such agents)."""
    import time
    m = make_map(ops, town['verts'], town['faces'], None, None, render=False)          # lists + hierarchy
    testing_lib.tds_testing_set_near_lists(2)                                          # lists, no hierarchy: beyond them the walk over grid rings
    m2 = make_map(ops, town['verts'], town['faces'], None, None, render=False)
    testing_lib.tds_testing_set_near_lists(1)
    assert m.info()['near_candidates'] == m2.info()['near_candidates'] > 0 and m.info()['bytes'] > m2.info()['bytes']
    gen = np.random.default_rng(31)
    lo, hi = town['verts'].min(0), town['verts'].max(0)
    ctr, half = (lo + hi) / 2, (hi - lo) / 2
    B, A = 48, 64
    ang = gen.uniform(0, 2 * np.pi, (B, A))
    dist = np.exp(gen.uniform(np.log(50.0), np.log(1.0e6), (B, A)))                    # 50 m ... 1 000 km beyond the bounding box
    xy = ctr + np.stack([np.cos(ang), np.sin(ang)], -1) * (np.abs(half).max() + dist)[..., None]
    xy[:8] = gen.uniform(lo - 120.0, hi + 120.0, (8, A, 2))                            # around the edge of the lists' grid, inside and out
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    state[-1, :3, :2] = [[np.inf, 0.0], [np.nan, 3.0], [-3.0e38, 3.0e38]]
    lw = np.concatenate([gen.uniform(4, 5, (B, A, 1)), gen.uniform(1.8, 2.2, (B, A, 1))], -1).astype(np.float32)
    sd, lwd = dev(state), dev(lw)
    sc = ops.heading_sc(sd[..., 2])
    for thr in (0.5, 0.0, 1.0e4):
        assert torch.equal(ops.offroad_forward(m, sd, lwd, sc, None, thr), ops.offroad_forward(m2, sd, lwd, sc, None, thr))
    sub = slice(8, 14)
    ref = oracle.offroad(state[sub], lw[sub], town['verts'], town['faces'], 0.5, sc=sc_np(sc)[sub])
    np.testing.assert_array_equal(ops.offroad_forward(m, sd[sub], lwd[sub], sc[sub], None, 0.5).cpu().numpy(), ref)
    # gradients: the arg-min face's, from the same descent
    fin = slice(0, B - 1)
    grads = []
    for mm in (m, m2):
        s_ = sd[fin].clone().requires_grad_(True)
        l_ = lwd[fin].clone().requires_grad_(True)
        ops.offroad(mm, s_, l_, threshold=0.5).sum().backward()
        grads.append((s_.grad.clone(), l_.grad.clone()))
    # (far from the map many faces are at the SAME float32 distance: the hierarchy and the lists give the gradient of the one of lowest index,
    #  as torch.min does; the ring walk that of the first it meets -- the foot points differ by metres over kilometres)
    #  Judged against the size of the position gradient: the gradients of length and width are sums of the corners' with alternating signs.)
    scale = grads[1][0][..., :2].abs().amax(-1, keepdim=True).clamp_min(1e-6)
    for ga, gb in zip(grads[0], grads[1]):
        err = ((ga - gb).abs() / scale).max()
        print('gradient, hierarchy against grid rings: largest difference / position gradient', float(err))
        assert float(err) <= 1e-3 and bool(ga.abs().sum() > 0)
    # the time of a batch whose agents have ALL strayed 50 m ... 3 km beyond the map (what the hierarchy is for), and of one a thousand
    # kilometres away (there every face lies within the 0.2 % of the conservative bounds: no walk can leave any out)
    for label, d_lo, d_hi, factor in (('50 m .. 3 km', 50.0, 3.0e3, 0.25), ('1 000 km', 0.9e6, 1.1e6, 1.5)):
        ang = gen.uniform(0, 2 * np.pi, (1024, A))
        dd = np.exp(gen.uniform(np.log(d_lo), np.log(d_hi), (1024, A)))
        fxy = ctr + np.stack([np.cos(ang), np.sin(ang)], -1) * (np.abs(half).max() + dd)[..., None]
        far = dev(np.concatenate([fxy, gen.uniform(-np.pi, np.pi, (1024, A, 1)), np.zeros((1024, A, 1))], -1).astype(np.float32))
        lwf, scf = dev(np.tile(lw[20:21], (1024, 1, 1))), ops.heading_sc(far[..., 2])
        ms, outs = [], []
        for mm in (m, m2):
            ops.offroad_forward(mm, far, lwf, scf, None, 0.5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs.append(ops.offroad_forward(mm, far, lwf, scf, None, 0.5))
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
        print(f'off-road of 65 536 agents {label} beyond the map: hierarchy {ms[0]:.2f} ms, grid rings {ms[1]:.2f} ms')
        assert torch.equal(outs[0], outs[1]) and ms[0] < factor * ms[1]


# ---------------------------------------------------------------------------------------------------- K3
def oracle_static(oracle, verts, faces, vert_category, categories):
    return oracle.static_mesh_arrays(verts, faces, vert_category, categories)


def render_both(ops, oracle, smap, static, state, size, mask, cam_xy, cam_sc, fov, res, out_dtype=torch.float32):
    sv, sa, sf = static
    B, N = state.shape[:2]
    sd = dev(state)
    agent_sc = ops.heading_sc(sd[..., 2])
    tmpl = dev(oracle.actor_template(size))               # host-side template (mesh.py:911-996), bit-exact vs reference (G4)
    csc = dev(cam_sc)
    img = ops.raster_scene(smap, sd, agent_sc, tmpl, actor_keys(smap, B, N), dev(mask), dev(cam_xy), csc, fov, res, out_dtype)
    ref = oracle.render_scenes(state, size, mask, cam_xy, cam_sc, sv, sa, sf, fov, res, agent_sc=sc_np(agent_sc))
    return img.cpu().numpy(), ref


@pytest.mark.parametrize('bits', [True, False])
def test_k3_untrimmed_rendering_equals_the_oracle(ops, oracle, town, bits):
    """CV2RendererConfig(trim_mesh_before_rendering=False), cv2.py:15,32-41: faces without a vertex in the view are drawn too -- long
    triangles that cross the image.  A hand-made map of huge triangles (which the trim rule drops entirely) over a crop of Town01."""
    gen = np.random.default_rng(41)
    B, A, res, fov = 2, 5, 128, 35.0
    crop = np.linalg.norm(town['verts'][town['faces']].mean(1) - np.array([100.0, 60.0]), axis=1) < 60
    faces = town['faces'][crop]
    used, inv = np.unique(faces, return_inverse=True)
    verts, vcat = town['verts'][used], town['vert_category'][used]
    faces = inv.reshape(-1, 3).astype(np.int32)
    big = np.array([[-400, -300], [600, -250], [150, 700], [-350, 500], [700, 400], [90, 55]], np.float32) + np.float32(0.25)
    nv = len(verts)
    verts = np.concatenate([verts, big]).astype(np.float32)
    vcat = np.concatenate([vcat, np.full(len(big), town['categories'].index('road'), vcat.dtype)])
    faces = np.concatenate([np.array([[nv, nv + 1, nv + 2], [nv + 3, nv + 4, nv + 5]], np.int32), faces])
    smap = make_map(ops, verts, faces, vcat, town['categories'])
    static = oracle_static(oracle, verts, faces, vcat, town['categories'])
    xy = np.array([100.0, 60.0]) + gen.uniform(-30, 30, (B, A, 2))
    st = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    sz = np.tile(np.array([4.5, 2.0], np.float32), (B, A, 1))
    mask = np.ones((B, A, A), bool)
    cam_sc = sc_np(ops.heading_sc(dev(st)[..., 2]))
    sd = dev(st)
    args = (smap, sd, ops.heading_sc(sd[..., 2]), dev(oracle.actor_template(sz)), actor_keys(smap, B, A), dev(mask), dev(st[..., :2].copy()), dev(cam_sc), fov, res)
    ops.use_bitplanes = bits
    try:
        trimmed = ops.raster_scene(*args).cpu().numpy()
        untrimmed = ops.raster_scene(*args, trim=False).cpu().numpy()
    finally:
        ops.use_bitplanes = True
    sv, sa, sf = static
    ref_t = oracle.render_scenes(st, sz, mask, st[..., :2].copy(), cam_sc, sv, sa, sf, fov, res, agent_sc=cam_sc)
    oracle.set_trim_mesh(False)
    try:
        ref_u = oracle.render_scenes(st, sz, mask, st[..., :2].copy(), cam_sc, sv, sa, sf, fov, res, agent_sc=cam_sc)
    finally:
        oracle.set_trim_mesh(True)
    np.testing.assert_array_equal(trimmed, ref_t)
    np.testing.assert_array_equal(untrimmed, ref_u)
    assert (ref_u != ref_t).mean() > 0.05                         # the huge triangles fill the background only when nothing is trimmed


def test_k3_untrimmed_golden_scene(ops, oracle, town):
    """the scene of G15 (whose untrimmed call list the oracle reproduces from the reference, tests/test_oracle_golden.py): kernel pixels = oracle"""
    g = load_golden('g15_preraster_untrimmed.npz')
    st, sz, pr = g['state'], g['size'], g['present']
    B, A = st.shape[:2]
    smap = make_map(ops, g['road_verts'], g['road_faces'], g['road_vert_category'], town['categories'])
    static = oracle_static(oracle, g['road_verts'], g['road_faces'], g['road_vert_category'], town['categories'])
    mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A)))
    sd = dev(st)
    sc = ops.heading_sc(sd[..., 2])
    for trim in (False, True):
        img = ops.raster_scene(smap, sd, sc, dev(oracle.actor_template(sz)), actor_keys(smap, B, A), dev(mask), dev(st[..., :2].copy()), dev(g['cam_sc']),
                               float(g['fov']), int(g['res']), trim=trim).cpu().numpy()
        oracle.set_trim_mesh(trim)
        try:
            ref = oracle.render_scenes(st, sz, mask, st[..., :2].copy(), g['cam_sc'], *static, float(g['fov']), int(g['res']), agent_sc=sc_np(sc))
        finally:
            oracle.set_trim_mesh(True)
        np.testing.assert_array_equal(img, ref)


def test_k3_golden_scenes_bit_exact(ops, oracle, town, testing_lib):
    g = load_golden('g45_mesh_preraster.npz')
    for m in json.loads(str(g['g5_meta'])):
        n = m['name']
        st, sz, pr = g[f'g5_{n}_state'], g[f'g5_{n}_size'], g[f'g5_{n}_present']
        B, A = st.shape[:2]
        if m['road'] == 'town01':
            mesh = (town['verts'], town['faces'], town['vert_category'])
        elif m['road'] == 'crop':
            mesh = (g[f'g5_{n}_road_verts'], g[f'g5_{n}_road_faces'], g[f'g5_{n}_road_vert_category'])
        else:
            mesh = (np.zeros((0, 2), np.float32), np.zeros((0, 3), np.int32), np.zeros(0, np.uint8))
        smap = make_map(ops, *mesh, town['categories'])
        static = oracle_static(oracle, *mesh, town['categories']) if len(mesh[1]) else (np.zeros((0, 3), np.float32),) * 2 + (np.zeros((0, 3), np.int32),)
        mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A)))
        # {bit-plane kernel} + strip widths x {binned, fused} packed-key kernels
        # (with bits=True a strip width >= 32 cuts the bit-plane kernel's image into strips as well)
        for tw, ws, bits in ((0, True, True), (32, True, True), (64, True, True), (0, False, True), (64, False, True), (0, True, False), (0, False, False),
                             (64, True, False), (16, True, False), (16, False, False)):      # (ws False, bits True): no work queues, one workgroup per item
            testing_lib.tds_raster_set_strip_width(tw)
            ops.use_workspace, ops.use_bitplanes = ws, bits
            ops._workspaces.clear()
            try:
                img, ref = render_both(ops, oracle, smap, static, st, sz, mask, st[..., :2].copy(), g[f'g5_{n}_cam_sc'], m['fov'], m['res'])
            finally:
                testing_lib.tds_raster_set_strip_width(0)
                ops.use_workspace, ops.use_bitplanes = True, True
                ops._workspaces.clear()
            assert img.shape == tuple(m['out_shape'])
            bad = (img != ref)
            assert not bad.any(), f'{n} tw={tw} ws={ws} bits={bits}: {bad.sum()} of {bad.size} values differ in images {np.unique(np.nonzero(bad)[0:2], axis=1)[:, :8]}'
        assert ref.any()


def test_k3_split_form_equals_the_fused_kernel_and_the_oracle(ops, oracle, town, testing_lib):
    """Up to 144 x 144 (float32) / 208 x 208 (uint8) the bit-plane path runs as two kernels -- K3s lists every camera's faces, K3r rasterises
    the lists -- with a third launch of the fused kernel over the cameras whose list overflowed.  Every form must paint the same pixels:
    forced either way (debug flags 8192 / 16384 of the testing build), cut into narrow strips (LDS budget), with a workspace so small that
    most lists overflow, and with K3r's short path for small faces (the three vertices inside the image and in at most two -- from 112 x 112
    on: four -- rows, painted by their lane on the spot) switched off (32768): the two set-ups paint the same pixels."""
    g = load_golden('g45_mesh_preraster.npz')
    st, sz, pr = g['g5_town01_128_state'], g['g5_town01_128_size'], g['g5_town01_128_present']
    B, A = st.shape[:2]
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    static = oracle_static(oracle, town['verts'], town['faces'], town['vert_category'], town['categories'])
    mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A)))
    cam_sc = g['g5_town01_128_cam_sc']
    real_ws = ops._raster_workspace
    try:
        for res, fov in ((64, 35.0), (108, 120.0), (112, 35.0), (128, 35.0), (192, 60.0), (256, 35.0), (320, 80.0)):
            ref = None
            for dtype in (torch.float32, torch.uint8):
                for flags, lds_kb, small_ws in ((8192, 40, False), (16384, 40, False), (16384, 16, False), (16384, 52, False), (16384, 40, True),
                                                (16384 | 32768, 40, False), (16384 | 32768, 16, False)):
                    testing_lib.tds_raster_set_debug(flags)
                    testing_lib.tds_raster_set_list_lds(lds_kb)
                    ops._workspaces.clear()
                    if small_ws:
                        # room for 300 faces per camera: most Town01 views hold more, their cameras go to the fused kernel
                        n_img = B * A
                        nbytes = ((n_img + 1) * 4 + 255) // 256 * 256 + (n_img * 4 + 255) // 256 * 256 + n_img * 300 * 16
                        ops._raster_workspace = lambda d, n, r, *rest: torch.empty(nbytes, dtype=torch.uint8, device=d)
                    try:
                        img, r = render_both(ops, oracle, smap, static, st, sz, mask, st[..., :2].copy(), cam_sc, fov, res, dtype)
                    finally:
                        ops._raster_workspace = real_ws
                    ref = r if ref is None else ref
                    bad = img.astype(np.float32) != ref
                    assert not bad.any(), f'res {res} {dtype} flags {flags} lds {lds_kb} small workspace {small_ws}: {bad.sum()} values differ'
            assert ref.any()
    finally:
        testing_lib.tds_raster_set_debug(0)
        testing_lib.tds_raster_set_list_lds(40)
        ops._workspaces.clear()


def test_k3_per_camera_triangles_in_every_form(ops, oracle, town, testing_lib):
    """Per-camera triangles (waypoint discs: `extra_tri` / `extra_key` of tds_raster_scene) through every form of the bit-plane path: the fused
    kernel's third producer phase, K3s's own loop over them (round 6: scan_faces_kernel<SceneArgsEx>, 150 triangles per camera = three rounds of
    64), lists that overflow, the packed-key kernels; float32 and uint8.  Same pixels everywhere, and they differ from the scene without them."""
    g = load_golden('g45_mesh_preraster.npz')
    st, sz, pr = g['g5_town01_128_state'], g['g5_town01_128_size'], g['g5_town01_128_present']
    B, A = st.shape[:2]
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    mask = dev(np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A))))
    sd = dev(st)
    agent_sc, tmpl, keys = ops.heading_sc(sd[..., 2]), dev(oracle.actor_template(sz)), actor_keys(smap, B, A)
    gen = np.random.default_rng(3)
    K = 150
    centre = st[:, :, None, None, :2] + gen.uniform(-18, 18, (B, A, K, 1, 2))
    tri = dev((centre + gen.uniform(-1.5, 1.5, (B, A, K, 3, 2))).astype(np.float32))
    wkey = (smap.rank_of(LEVELS['vehicle']) << 24) | pack((139, 64, 0))           # a key of its own: one more bit plane
    ekey = torch.full((B, A, K), wkey, dtype=torch.int64).to(torch.int32).to(DEV)
    ekey[:, :, ::7] = 0                                                             # key 0: no triangle
    real_ws = ops._raster_workspace
    try:
        for res, fov in ((64, 35.0), (128, 35.0), (256, 35.0)):
            for dtype in (torch.float32, torch.uint8):
                ref = None
                for flags, small_ws, bits in ((8192, False, True), (16384, False, True), (16384, True, True), (0, False, True), (0, False, False)):
                    testing_lib.tds_raster_set_debug(flags)
                    ops._workspaces.clear()
                    if small_ws:
                        n_img = B * A
                        nbytes = ((n_img + 1) * 4 + 255) // 256 * 256 + (n_img * 4 + 255) // 256 * 256 + n_img * 300 * 20
                        ops._raster_workspace = lambda d, n, r, *rest: torch.empty(nbytes, dtype=torch.uint8, device=d)
                    ops.use_bitplanes = bits
                    try:
                        img = ops.raster_scene(smap, sd, agent_sc, tmpl, keys, mask, sd[..., :2].contiguous(), agent_sc, fov, res, dtype, extra_tri=tri, extra_key=ekey)
                    finally:
                        ops._raster_workspace = real_ws
                        ops.use_bitplanes = True
                    ref = img if ref is None else ref
                    assert torch.equal(img, ref), f'res {res} {dtype} flags {flags} small workspace {small_ws} bit planes {bits}: {(img != ref).sum().item()} values differ'
                plain = ops.raster_scene(smap, sd, agent_sc, tmpl, keys, mask, sd[..., :2].contiguous(), agent_sc, fov, res, dtype)
                assert (plain != ref).any()
    finally:
        testing_lib.tds_raster_set_debug(0)
        ops._workspaces.clear()


def test_k3_six_to_ten_keys_at_256_in_every_workgroup_shape(ops, oracle, town, testing_lib):
    """Two or three agent types (six / seven distinct keys): at 256 x 256 the bit planes no longer fit three workgroups per CU.  Round 6 renders the
    whole image in 8-wave workgroups, two per CU (raster.hip: raster_scene_impl; profiles/r06_more_keys.log); the alternatives stay reachable in
    the testing build -- the whole image in 4-wave workgroups (debug flag 262144), two half-image strips (524288).  Same pixels as the oracle in
    all three, float32 and uint8."""
    types = dict(vehicle=(4, (32, 74, 135)), bicycle=(5, (255, 150, 40)), pedestrian=(6, (255, 64, 180)), ego=(3, (255, 0, 0)), ground_truth=(9, (196, 188, 165)),
                 prediction=(10, (255, 155, 0)))
    levels = sorted(set(LEVEL_TABLE) | {float(z) for z, _ in types.values()}, reverse=True)
    cats = town['categories']
    cat = np.asarray(town['vert_category'])[town['faces'][:, 0]]
    smap = ops.StaticMap(town['verts'], town['faces'], np.array([LEVELS[cats[c]] for c in cat], np.float32),
                         np.array([pack(COLORS[cats[c]]) for c in cat], np.uint32), levels, device=DEV)
    static = oracle.static_mesh_arrays(town['verts'], town['faces'], town['vert_category'], cats, colors={**oracle.DEFAULT_COLORS, **COLORS},
                                       levels={**oracle.DEFAULT_LEVELS, **LEVELS})
    road = town['verts'][np.asarray(town['vert_category']) == cats.index('road')]
    gen = np.random.default_rng(9)
    B, A, res, fov = 2, 14, 256, 35.0
    try:
        # ... and nine / ten keys (five / six agent types): their planes exceed 64 KiB for a whole image -- strips of equal width (round 6; testing flag
        # 1048576: the widest that fit, 192 + 64 columns), for ten keys one strip more than necessary (2097152: not)
        for names in (['vehicle', 'pedestrian'], ['vehicle', 'bicycle', 'pedestrian'], ['vehicle', 'bicycle', 'pedestrian', 'ego', 'ground_truth'],
                      ['vehicle', 'bicycle', 'pedestrian', 'ego', 'ground_truth', 'prediction']):
            anchor = road[gen.integers(0, len(road), (B, 1))]
            state = np.concatenate([anchor + gen.uniform(-20, 20, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
            size = np.concatenate([gen.uniform(1.0, 8, (B, A, 1)), gen.uniform(0.6, 2.8, (B, A, 1))], -1).astype(np.float32)
            kind = gen.integers(0, len(names), (B, A))
            kind[:, :len(names)] = np.arange(len(names))                  # every type occurs
            mask = np.ascontiguousarray((gen.uniform(size=(B, 1, A)) < 0.9) & (gen.uniform(size=(B, A, A)) < 0.95))
            body = np.array([(smap.rank_of(types[n][0]) << 24) | pack(types[n][1]) for n in names], np.int64)
            dkey = (smap.rank_of(LEVELS['direction']) << 24) | pack(COLORS['direction'])
            keys = torch.from_numpy(np.stack([body[kind], np.full_like(kind, dkey)], -1)).to(torch.int32).to(DEV)
            lev = np.stack([np.array([types[n][0] for n in names], np.float32)[kind], np.full(kind.shape, LEVELS['direction'], np.float32)], -1)
            col = np.stack([np.array([types[n][1] for n in names], np.float32)[kind], np.broadcast_to(np.array(COLORS['direction'], np.float32), kind.shape + (3,))], -2) / np.float32(255.0)
            sd = dev(state)
            agent_sc = ops.heading_sc(sd[..., 2])
            ref = oracle.render_scenes(state, size, mask, state[..., :2].copy(), sc_np(agent_sc), *static, fov, res, agent_sc=sc_np(agent_sc),
                                       actor_levels=lev, actor_colors=col.astype(np.float32))
            assert ref.any()
            for flags in ((0, 262144, 524288) if len(names) <= 3 else (0, 1048576, 2097152)):
                testing_lib.tds_raster_set_debug(flags)
                for dtype in (torch.float32, torch.uint8):
                    img = ops.raster_scene(smap, sd, agent_sc, dev(oracle.actor_template(size)), keys, dev(mask), dev(state[..., :2].copy()), agent_sc, fov, res, dtype)
                    bad = img.cpu().numpy().astype(np.float32) != ref
                    assert not bad.any(), f'{len(names)} agent types, debug {flags}, {dtype}: {bad.sum()} values differ'
    finally:
        testing_lib.tds_raster_set_debug(0)


def test_k3_u8_mode_equals_f32(ops, oracle, town):
    g = load_golden('g45_mesh_preraster.npz')
    st, sz, pr = g['g5_town01_128_state'], g['g5_town01_128_size'], g['g5_town01_128_present']
    B, A = st.shape[:2]
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    static = oracle_static(oracle, town['verts'], town['faces'], town['vert_category'], town['categories'])
    mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A)))
    img8, ref = render_both(ops, oracle, smap, static, st, sz, mask, st[..., :2].copy(), g['g5_town01_128_cam_sc'], 35.0, 128, torch.uint8)
    np.testing.assert_array_equal(img8.astype(np.float32), ref)


def test_k3_random_town01_256_bit_exact(ops, oracle, town):
    """BASELINE config-2 image shape (256x256, fov 35) on the real Town01 mesh, random scenes incl. masked agent 0,
    visibility masks and cameras that look off the map."""
    gen = np.random.default_rng(17)
    B, A = 3, 12
    road = town['verts'][town['vert_category'] == town['categories'].index('road')]
    anchor = road[gen.integers(0, len(road), (B, 1))]
    xy = anchor + gen.uniform(-25, 25, (B, A, 2))
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 10, (B, A, 1))], -1).astype(np.float32)
    state[2, 3, :2] += 2000.0
    size = np.concatenate([gen.uniform(4, 5, (B, A, 1)), gen.uniform(1.8, 2.2, (B, A, 1))], -1).astype(np.float32)
    present = gen.uniform(size=(B, A)) < 0.8
    present[1, 0] = False
    mask = np.ascontiguousarray(present[:, None, :] & (gen.uniform(size=(B, A, A)) < 0.9))
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    static = oracle_static(oracle, town['verts'], town['faces'], town['vert_category'], town['categories'])
    sd = dev(state)
    cam_sc = sc_np(ops.heading_sc(sd[..., 2]))
    for ws, bits in ((True, True), (False, True), (True, False), (False, False)):      # (False, True): no workspace, hence no work queues -- the persistent kernel with one workgroup per camera
        ops.use_workspace, ops.use_bitplanes = ws, bits
        try:
            img, ref = render_both(ops, oracle, smap, static, state, size, mask, state[..., :2].copy(), cam_sc, 35.0, 256)
        finally:
            ops.use_workspace, ops.use_bitplanes = True, True
        bad = img != ref
        assert not bad.any(), f'workspace={ws} bitplanes={bits}: {bad.sum()} values differ'
    assert (ref > 0).mean() > 0.05


@pytest.mark.parametrize('res', [320, 512])
def test_k3_large_resolutions_use_strips(ops, oracle, town, res):
    """above 256x256 with five keys the bit planes of a whole image no longer fit 64 KiB: the bit-plane kernel renders the camera in
    strips (res 320 is not a multiple of 64 either); same pixels as the oracle"""
    gen = np.random.default_rng(23)
    B, A = 1, 5
    road = town['verts'][town['vert_category'] == town['categories'].index('road')]
    xy = road[gen.integers(0, len(road), (B, 1))] + gen.uniform(-12, 12, (B, A, 2))
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 10, (B, A, 1))], -1).astype(np.float32)
    size = np.tile(np.array([4.6, 2.0], np.float32), (B, A, 1))
    mask = np.ones((B, A, A), bool)
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    static = oracle_static(oracle, town['verts'], town['faces'], town['vert_category'], town['categories'])
    cam_sc = sc_np(ops.heading_sc(dev(state)[..., 2]))
    img, ref = render_both(ops, oracle, smap, static, state, size, mask, state[..., :2].copy(), cam_sc, 50.0, res)
    bad = img != ref
    assert not bad.any(), f'{bad.sum()} values differ'
    assert (ref > 0).mean() > 0.05


@pytest.mark.parametrize('res,fov', [(64, 30.0), (256, 35.0), (256, 12.0), (512, 25.0)])
def test_k3_triangle_soups_exercise_the_row_rule(ops, oracle, res, fov):
    """Random triangle soups instead of a road network: slivers, specks, faces larger than the view, everything overlapping, so that the row rule
    of the bit-plane kernel (outline edges merged into the rows; tests/fill_rows_model.c) meets y- and x-major edges of every length -- at 512
    pixels and a 25 m view also edges beyond its 100- and 147-row limits, which are walked exactly -- flat tops and bottoms, ties, faces that
    leave the image on every side; trimmed and untrimmed."""
    gen = np.random.default_rng(res * 1000 + int(fov))
    n = 2500
    c = gen.uniform(-30, 30, (n, 1, 2))
    kind = gen.integers(0, 5, n)
    scale = np.choose(kind, [0.15, 0.6, 3.0, 12.0, 40.0])[:, None, None]
    tri = c + gen.normal(0, 1, (n, 3, 2)) * scale
    sl = kind == 1                                                          # slivers: two vertices 0.1 m apart, the third several metres away
    tri[sl, 1] = tri[sl, 0] + gen.normal(0, 0.1, (sl.sum(), 2))
    tri[sl, 2] = tri[sl, 0] + gen.normal(0, 4.0, (sl.sum(), 2))
    ax = gen.integers(0, 4, n) == 0                                         # some axis-parallel edges (they stay so for the unrotated cameras)
    tri[ax, 1, 1] = tri[ax, 0, 1]
    verts = tri.reshape(-1, 2).astype(np.float32)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    cats = ['road', 'left_lane', 'right_lane']
    vc = np.repeat(gen.integers(0, 3, n), 3).astype(np.int64)
    smap = make_map(ops, verts, faces, vc, cats)
    static = oracle_static(oracle, verts, faces, vc, cats)
    B, A = 3, 4
    state = np.concatenate([gen.uniform(-25, 25, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    state[0, :, 2] = np.array([0.0, np.pi / 2, np.pi, -np.pi / 2])          # unrotated cameras: axis-parallel edges stay horizontal / vertical
    size = np.concatenate([gen.uniform(3.5, 6, (B, A, 1)), gen.uniform(1.6, 2.4, (B, A, 1))], -1).astype(np.float32)
    mask = gen.uniform(size=(B, A, A)) < 0.8
    cam_sc = sc_np(ops.heading_sc(dev(state)[..., 2]))
    sd = dev(state)
    args = (smap, sd, ops.heading_sc(sd[..., 2]), dev(oracle.actor_template(size)), actor_keys(smap, B, A), dev(mask), dev(state[..., :2].copy()), dev(cam_sc), fov, res)
    sv, sa, sf = static
    for trim in (True, False):
        img = ops.raster_scene(*args, trim=trim).cpu().numpy()
        oracle.set_trim_mesh(trim)
        try:
            ref = oracle.render_scenes(state, size, mask, state[..., :2].copy(), cam_sc, sv, sa, sf, fov, res, agent_sc=cam_sc)
        finally:
            oracle.set_trim_mesh(True)
        bad = img != ref
        assert not bad.any(), f'res {res} fov {fov} trim {trim}: {bad.sum()} values differ in {bad.any(axis=(2, 3, 4)).sum()} images'
        assert (ref > 0).mean() > 0.2


@pytest.mark.parametrize('res,fov', [(64, 35.0), (128, 60.0), (256, 35.0), (96, 12.0)])
def test_k3_paired_faces_of_every_kind(ops, oracle, res, fov):
    """The rendering grid pairs same-key faces that share an edge (tds_common.h: QuadEntry; the split form's scan kernel fetches, projects and
    trims a pair once) while every triangle is still drawn with the reference's per-face semantics (rendering/cv2.py:44-59: one
    cv2.fillConvexPoly per face, in its own vertex order).  A mesh of quads of every kind -- convex, concave (darts), folded over their
    diagonal, slivers, with a repeated point -- triangulated with either diagonal and any vertex order in both halves; neighbours of ANOTHER
    colour over the same edge (never paired); fans around a hub (odd numbers: the greedy pairing leaves lone faces); faces stacked three deep
    on one edge; duplicates.  64 .. 128 pixels take the split form (pairs), 256 the fused kernel (single triangles): both equal the oracle."""
    gen = np.random.default_rng(res * 77 + int(fov))
    V, F, C = [], [], []

    def add(pts, faces, cat):
        base = len(V)
        V.extend(pts)
        for f in faces:
            f = list(f)
            r = int(gen.integers(0, 3))
            f = f[r:] + f[:r]
            if gen.integers(0, 2):
                f = f[::-1]
            F.append([base + i for i in f])
            C.append(cat)
    for q in range(1400):
        c = gen.uniform(-30, 30, 2)
        kind = int(gen.integers(0, 7))
        ang = gen.uniform(0, 2 * np.pi)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        w, h = gen.uniform(0.1, 3.0), gen.uniform(0.3, 9.0)
        ring = np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) * 0.5
        if kind == 1:
            ring[2] = [0.05 * w, 0.1 * h]                                  # concave: a dart
        elif kind == 2:
            ring[[2, 3]] = ring[[3, 2]]                                    # folded over: a bow tie
        elif kind == 3:
            ring[:, 0] *= 0.03                                             # a sliver (lane marking)
        elif kind == 4:
            ring[3] = ring[0]                                              # a repeated point: one half is degenerate
        ring = ring + gen.normal(0, 0.05, ring.shape)
        pts = [tuple((R @ p + c).astype(np.float32)) for p in ring]
        diag = int(gen.integers(0, 2))
        faces = [(0, 1, 2), (0, 2, 3)] if diag == 0 else [(0, 1, 3), (1, 2, 3)]
        cat = int(gen.integers(0, 3))
        if kind == 5:
            add(pts, faces[:1], cat); add(pts, faces[1:], (cat + 1) % 3)   # the two halves in different colours (shared coordinates, never a pair)
        elif kind == 6:
            add(pts, faces + faces[:1], cat)                               # three faces on one edge, one of them a duplicate
        else:
            add(pts, faces, cat)
    for fan in range(60):                                                  # fans: 3 .. 9 triangles around a hub, consecutive ones share an edge
        c = gen.uniform(-30, 30, 2)
        n = int(gen.integers(3, 10))
        angs = np.sort(gen.uniform(0, 2 * np.pi, n + 1))
        rad = gen.uniform(0.5, 6.0, n + 1)
        pts = [tuple(c.astype(np.float32))] + [tuple((c + r * np.array([np.cos(t), np.sin(t)])).astype(np.float32)) for t, r in zip(angs, rad)]
        add(pts, [(0, i + 1, i + 2) for i in range(n)], int(gen.integers(0, 3)))
    verts = np.array(V, np.float32)
    faces = np.array(F, np.int32)
    # the colour belongs to the face's first vertex (cv2.py:58): give every face its own three vertices so that a face's colour is its own
    verts = verts[faces.reshape(-1)]
    vc = np.repeat(np.array(C, np.int64), 3)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    cats = ['road', 'left_lane', 'right_lane']
    smap = make_map(ops, verts, faces, vc, cats)
    info = smap.info()
    assert info['pairs'] > 800 and info['render_entries'] == info_entries_of(info), info
    static = oracle_static(oracle, verts, faces, vc, cats)
    B, A = 3, 4
    state = np.concatenate([gen.uniform(-25, 25, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    state[0, :, 2] = np.array([0.0, np.pi / 2, np.pi, -np.pi / 2])
    size = np.concatenate([gen.uniform(3.5, 6, (B, A, 1)), gen.uniform(1.6, 2.4, (B, A, 1))], -1).astype(np.float32)
    mask = gen.uniform(size=(B, A, A)) < 0.8
    cam_sc = sc_np(ops.heading_sc(dev(state)[..., 2]))
    sd = dev(state)
    args = (smap, sd, ops.heading_sc(sd[..., 2]), dev(oracle.actor_template(size)), actor_keys(smap, B, A), dev(mask), dev(state[..., :2].copy()), dev(cam_sc), fov, res)
    sv, sa, sf = static
    for trim in (True, False):
        for dtype in (torch.float32, torch.uint8):
            img = ops.raster_scene(*args, trim=trim, out_dtype=dtype).cpu().numpy().astype(np.float32)
            oracle.set_trim_mesh(trim)
            try:
                ref = oracle.render_scenes(state, size, mask, state[..., :2].copy(), cam_sc, sv, sa, sf, fov, res, agent_sc=cam_sc)
            finally:
                oracle.set_trim_mesh(True)
            bad = img != ref
            assert not bad.any(), f'res {res} fov {fov} trim {trim} {dtype}: {bad.sum()} values differ in {bad.any(axis=(2, 3, 4)).sum()} images'
            assert (ref > 0).mean() > 0.05


def info_entries_of(info):
    """(the rendering grid holds one entry per lone face or pair and cell: at most the triangle grid's, at least half of it)"""
    assert info['entries'] // 2 <= info['render_entries'] <= info['entries']
    return info['render_entries']


def test_k3_faces_far_outside_the_packed_coordinate_range(ops, oracle):
    """a ground quad of 1 km seen at 51 px / m puts its vertices 25 000 pixels away from the image: such faces cannot be packed into
    16-bit coordinates and take the sequential exact path (fill_generic / fill_generic_bits), on every kernel family"""
    verts = np.array([[-500, -500], [500, -500], [500, 500], [-500, 500], [3, 3], [400, 3.5], [400, 9]], np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.int32)
    vc = np.array([0, 0, 0, 0, 1, 1, 1], np.int64)
    cats = ['road', 'left_lane']
    smap = make_map(ops, verts, faces, vc, cats)
    static = oracle_static(oracle, verts, faces, vc, cats)
    state = np.array([[[1.0, 2.0, 0.4, 0.0], [4.0, -1.0, 2.0, 0.0]]], np.float32)
    size = np.array([[[4.5, 2.0], [4.5, 2.0]]], np.float32)
    mask = np.ones((1, 2, 2), bool)
    cam_sc = sc_np(ops.heading_sc(dev(state)[..., 2]))
    for ws, bits in ((True, True), (False, True), (True, False), (False, False)):      # (False, True): no workspace, hence no work queues -- the persistent kernel with one workgroup per camera
        ops.use_workspace, ops.use_bitplanes = ws, bits
        try:
            img, ref = render_both(ops, oracle, smap, static, state, size, mask, state[..., :2].copy(), cam_sc, 5.0, 256)
        finally:
            ops.use_workspace, ops.use_bitplanes = True, True
        assert not (img != ref).any(), f'workspace={ws} bitplanes={bits}: {(img != ref).sum()} values differ'
    assert len(np.unique(ref.reshape(-1, 3, 256 * 256).transpose(0, 2, 1).reshape(-1, 3), axis=0)) >= 3


def test_k3_generic_mesh_path(ops, oracle, town):
    """BirdviewRenderer.render_rgb_mesh on an explicit per-camera RGB mesh (the reference's own dataflow)."""
    g = load_golden('g45_mesh_preraster.npz')
    verts, faces, vc = g['g4_bg_verts'], g['g4_bg_faces'], g['g4_bg_vert_category']
    sv, sa, sf = oracle.static_mesh_arrays(verts, faces, vc, town['categories'])
    n = 5
    gen = np.random.default_rng(2)
    cam_xy = (np.array([100.0, 2.0]) + gen.uniform(-6, 6, (n, 2))).astype(np.float32)
    psi = torch.from_numpy(gen.uniform(-np.pi, np.pi, n).astype(np.float32)).to(DEV)
    cam_sc = ops.heading_sc(psi)
    V = np.broadcast_to(sv, (n,) + sv.shape).copy()
    Aat = np.broadcast_to(sa, (n,) + sa.shape).copy()
    Fc = np.broadcast_to(sf, (n,) + sf.shape).copy()
    levels = sorted(set(float(z) for z in sv[:, 2]), reverse=True)
    for res, fov in ((64, 35.0), (100, 20.0)):          # 100 is not a multiple of 16: scalar write-out path
        out = ops.raster_mesh(dev(V), dev(Aat), dev(Fc), dev(cam_xy), cam_sc, levels, 2.0 / fov, res).cpu().numpy()
        ref = oracle.render_rgb_mesh(V, Aat, Fc, cam_xy, sc_np(cam_sc), 2.0 / fov, res)          # n x H x W x 3
        np.testing.assert_array_equal(out, np.transpose(ref, (0, 3, 1, 2)))
        assert ref.any()


def test_k3_full_size_properties(ops, town):
    """BASELINE headline image size at a batch the oracle could not render in seconds: size-independent properties
    (SURVEY 8c): every value is one of the palette colours; u8 and f32 modes agree; masking every agent leaves only
    map colours + one possible stray dot per image; rendering is idempotent."""
    gen = np.random.default_rng(23)
    B, A = 16, 64
    road = town['verts'][town['vert_category'] == town['categories'].index('road')]
    anchor = road[gen.integers(0, len(road), (B, 1))]
    xy = anchor + gen.uniform(-30, 30, (B, A, 2))
    state = dev(np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 10, (B, A, 1))], -1).astype(np.float32))
    size = np.concatenate([gen.uniform(4, 5, (B, A, 1)), gen.uniform(1.8, 2.2, (B, A, 1))], -1).astype(np.float32)
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    from oracle import oracle as orc
    tmpl = dev(orc.actor_template(size))
    sc = ops.heading_sc(state[..., 2])
    mask = torch.ones(B, A, A, dtype=torch.bool, device=DEV)
    keys = actor_keys(smap, B, A)
    cam_xy = state[..., :2].contiguous()
    f = ops.raster_scene(smap, state, sc, tmpl, keys, mask, cam_xy, sc, 35.0, 256)
    u = ops.raster_scene(smap, state, sc, tmpl, keys, mask, cam_xy, sc, 35.0, 256, torch.uint8)
    assert f.shape == (B, A, 3, 256, 256)
    assert torch.equal(f, u.float())
    assert torch.equal(f, ops.raster_scene(smap, state, sc, tmpl, keys, mask, cam_xy, sc, 35.0, 256))
    packed = (u[:, :, 0].int() << 16) | (u[:, :, 1].int() << 8) | u[:, :, 2].int()
    palette = {0} | {pack(c) for c in COLORS.values()}
    assert set(torch.unique(packed).cpu().tolist()) <= palette
    # ego box is centred: the camera pixel carries the vehicle or its direction marker
    centre = packed[:, :, 128, 128].cpu()
    assert set(torch.unique(centre).tolist()) <= {pack(COLORS['vehicle']), pack(COLORS['direction'])}
    none = ops.raster_scene(smap, state, sc, tmpl, keys, torch.zeros_like(mask), cam_xy, sc, 35.0, 256, torch.uint8)
    pn = (none[:, :, 0].int() << 16) | (none[:, :, 1].int() << 8) | none[:, :, 2].int()
    veh = (pn == pack(COLORS['vehicle'])).flatten(2).sum(-1)
    assert int(veh.max()) <= 1 and not (pn == pack(COLORS['direction'])).any()


def test_k3_index_slices_errors_by_code(ops, town):
    """`_ops.raster_scene` retries without index slices only when the library reports a documented capacity (TDS_ELIMIT); a caller
    or ABI bug (a buffer that is too small: TDS_EINVAL) must surface.  Straight through the C ABI, then through the wrapper."""
    import ctypes
    from torchdrivesim_amd import _native as nat
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    cam_xy = dev(np.array([[[100.0, 200.0]]], np.float32))
    cam_sc = dev(np.array([[[0.0, 1.0]]], np.float32))
    out = torch.empty((1, 1, 3, 64, 64), dtype=torch.float32, device=DEV)
    small = torch.empty(4, dtype=torch.int32, device=DEV)
    aux = nat.RasterAux(index_slices=small.data_ptr(), index_slices_bytes=16)
    with pytest.raises(nat.TdsError) as ei:
        nat.call('tds_raster_scene', out.device, smap.handle, None, None, None, None, None, nat.dev_ptr(cam_xy, torch.float32, 'cam_xy'),
                 nat.dev_ptr(cam_sc, torch.float32, 'cam_sc'), 1, 1, 0, float(2.0 / 35.0), 64, nat.OUT_F32, nat.dev_ptr(out, torch.float32, 'out'),
                 None, 0, None, 0, 0, None, None, 0, ctypes.cast(ctypes.pointer(aux), ctypes.c_void_p), nat.stream_ptr(out.device))
    assert ei.value.code == nat.E_INVAL and 'too small' in str(ei.value)
    # the wrapper: a resolution that is no multiple of 4 has no slices (decided before the call), 20 distinct actor keys exceed the bit-plane
    # kernel (TDS_ELIMIT inside the call): both render, without slices
    st = dev(np.array([[[100.0, 200.0, 0.3, 1.0]] * 20], np.float32))
    tm = dev(np.tile(np.array([[2, 1], [-2, 1], [-2, -1], [2, -1], [2, 0.5], [2, -0.5], [1, 0]], np.float32), (1, 20, 1, 1)))
    keys = (torch.arange(1, 21, dtype=torch.int32, device=DEV)[None, :, None] * 0x10101 + (5 << 24)).expand(1, 20, 2).contiguous()
    img, sl, kt = ops.raster_scene(smap, st, ops.heading_sc(st[..., 2]), tm, keys, torch.ones(1, 1, 20, dtype=torch.bool, device=DEV), cam_xy, cam_sc, 35.0, 64,
                                   index_slices=True)
    assert sl is None and kt is None and img.shape == (1, 1, 3, 64, 64) and bool((img > 0).any())


@pytest.mark.parametrize('B,A', [(3, 5), (1, 1), (5, 13)])
def test_k3_camera_counts_that_do_not_divide_by_eight(ops, oracle, town, B, A):
    """The persistent launch cuts its work items into eight queues only when they divide evenly, else into one; K3s packs four cameras per
    workgroup.  Camera counts of 15, 1 and 65 at the resolutions of every form (fused persistent, fused one-item, split), both output types."""
    smap = make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories'])
    static = oracle_static(oracle, town['verts'], town['faces'], town['vert_category'], town['categories'])
    gen = np.random.default_rng(B * 100 + A)
    road = town['verts'][town['vert_category'] == town['categories'].index('road')]
    xy = road[gen.integers(0, len(road), (B, 1))] + gen.uniform(-20, 20, (B, A, 2))
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 10, (B, A, 1))], -1).astype(np.float32)
    size = np.concatenate([gen.uniform(3.5, 6, (B, A, 1)), gen.uniform(1.6, 2.4, (B, A, 1))], -1).astype(np.float32)
    mask = np.broadcast_to((gen.uniform(size=(B, A)) < 0.9)[:, None, :], (B, A, A)).copy()
    cam_sc = sc_np(ops.heading_sc(dev(state)[..., 2]))
    for res in (64, 128, 256, 320):
        for dtype in (torch.float32, torch.uint8):
            img, ref = render_both(ops, oracle, smap, static, state, size, mask, state[..., :2].copy(), cam_sc, 35.0, res, dtype)
            assert img.shape == (B, A, 3, res, res)
            np.testing.assert_array_equal(img.astype(np.float32), ref)
