"""
Pins the CPU oracle (oracle/tds_oracle.c) to golden vectors captured from the imported reference
by tools/gen_golden.py (fixtures G1-G5).  CPU only.
sin/cos are taken from torch, as the reference does (simulator.py:940, _iou_utils.py:290-291), so the
remaining arithmetic is IEEE basic operations and the comparison can be bit-exact.
"""
import json
from collections import Counter

import numpy as np
import pytest
import torch

from conftest import load_golden


def tsc(psi):
    a = torch.from_numpy(np.ascontiguousarray(psi, dtype=np.float32))
    return torch.stack([torch.sin(a), torch.cos(a)], -1).numpy()


def tsc_discs(box):
    t = torch.from_numpy(np.ascontiguousarray(box, dtype=np.float32))
    a = t[..., 4] + (np.pi / 2) * (t[..., 3] > t[..., 2])
    return torch.stack([torch.sin(a), torch.cos(a)], -1).numpy()


def boxes_of(state, size):
    return np.concatenate([state[..., :2], size, state[..., 2:3]], -1)


# ---------------------------------------------------------------- G1
def test_kinematics_match_reference(oracle):
    g = load_golden('g1_kinematic.npz')
    rel = lambda a, b: np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))
    s, a, lr = g['state'], g['action'], g['lr']
    assert rel(oracle.bicycle_step(s, a, lr), g['out_bicycle']) <= 1e-6
    assert rel(oracle.bicycle_step(s, a, lr, left_handed=True), g['out_bicycle_lh']) <= 1e-6
    assert rel(oracle.bicycle_step(s, a, lr, dt=0.25), g['out_bicycle_dt']) <= 1e-6
    assert rel(oracle.bicycle_step(s, a, lr, no_reversing=True), g['out_norev']) <= 1e-6
    two = oracle.bicycle_step(oracle.bicycle_step(s, a, lr), a[::-1], lr)
    assert rel(two, g['out_bicycle_2steps']) <= 1e-6
    assert rel(oracle.simple_step(s, g['action4']), g['out_simple']) <= 1e-6
    assert rel(oracle.simple_step(s, g['action4'], oriented=True), g['out_oriented']) <= 1e-6
    assert np.abs(oracle.bicycle_fit_action(g['future'], s) - g['fit_bicycle']).max() <= 1e-5
    assert np.abs(oracle.bicycle_fit_action(g['future'], s, left_handed=True) - g['fit_bicycle_lh']).max() <= 1e-5


def test_bicycle_by_displacement_matches_reference(oracle):
    """G1b: BicycleByDisplacement / BicycleByOrientedDisplacement (kinematic.py:526-587) = fit_action then step"""
    g = load_golden('g1b_bicycle_displacement.npz')
    rel = lambda a, b: np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))
    s, a, lr = g['state'], g['action'], g['lr']
    for name, kw in (('disp', {}), ('disp_max5', dict(max_dx=5.0, model_dt=0.2)), ('oriented', dict(oriented=True))):
        one = oracle.bicycle_by_displacement_step(s, a, lr, **kw)
        assert rel(one, g[f'out_{name}']) <= 2e-6, name
        two = oracle.bicycle_by_displacement_step(one, a[:, ::-1], lr, dt=0.05, **kw)          # dt of the step differs from the model's
        assert rel(two, g[f'out_{name}_2steps']) <= 2e-6, name


def test_bicycle_known_answer(oracle):
    # SURVEY 8c G1 known answer, kinematic.py:462-477
    out = oracle.bicycle_step(np.array([[[1, 2, 0.5, 3]]]), np.array([[[0.4, -0.2]]]), np.array([[1.5]]))
    np.testing.assert_allclose(out[0, 0], [1.31449008, 2.05912733, 0.43407637, 3.2], rtol=1e-6)
    np.testing.assert_array_equal(out, load_golden('g1_kinematic.npz')['kat_out'])


# ---------------------------------------------------------------- G2
def test_box2corners_bit_exact(oracle):
    g = load_golden('g2_boxes.npz')
    c = oracle.box2corners(g['cur_box1'], tsc(g['cur_box1'][..., 4]))
    np.testing.assert_array_equal(c, g['cur_corners1'])


@pytest.mark.parametrize('tag', ['cur', 'rnd0', 'rnd400'])
def test_iou_and_discs_bit_exact(oracle, tag):
    g = load_golden('g2_boxes.npz')
    b1, b2 = g[tag + '_box1'], g[tag + '_box2']
    iou, idx, nv, area = oracle.iou_pairs(b1, b2, tsc(b1[..., 4]), tsc(b2[..., 4]), debug=True)
    ref = g[tag + '_iou']
    assert ((iou == ref) | (np.isnan(iou) & np.isnan(ref))).all()
    np.testing.assert_array_equal(area, g[tag + '_area'])
    if tag != 'cur':
        np.testing.assert_array_equal(nv, g[tag + '_nvalid'])
        np.testing.assert_array_equal(idx, g[tag + '_idx'])      # sort_indices incl. padding rule
    d = oracle.discs_pairs(b1, b2, tsc_discs(b1), tsc_discs(b2))
    ref = g[tag + '_discs']
    assert ((d == ref) | (np.isnan(d) & np.isnan(ref))).all()


@pytest.mark.parametrize('num_discs', [3, 7, 9, 25])
def test_discs_with_other_disc_counts_bit_exact(oracle, num_discs):
    g = load_golden('g13_discs_n.npz')                            # collision_detection_with_discs(..., num_discs=k) of the reference
    b1, b2 = g['box1'], g['box2']
    d = oracle.discs_pairs(b1, b2, tsc_discs(b1), tsc_discs(b2), num_discs=num_discs)
    np.testing.assert_array_equal(d, g[f'discs_{num_discs}'])
    assert (d > 0).sum() > 100


def test_iou_known_answers(oracle):
    # SURVEY R3 known answers (base box (0,0,4,2,0))
    base = [0, 0, 4, 2, 0]
    others = [base, [2, 0, 4, 2, 0], [0, 0, 4, 2, np.pi / 2], [0, 0, 4, 2, np.pi / 4], [1, 0.5, 4, 2, np.radians(30)],
              [0, 0, 2, 1, 0.3], [4, 0, 4, 2, 0]]
    b1 = np.array([base] * len(others), np.float32)
    b2 = np.array(others, np.float32)
    iou = oracle.iou_pairs(b1, b2, tsc(b1[:, 4]), tsc(b2[:, 4]))
    np.testing.assert_allclose(iou, [1.0, 0.33333334, 0.33333334, 0.51742810, 0.43370697, 0.25, 0.0], atol=2e-7)


def test_sort_indices_example(oracle):
    # boxes (0,0,4,2,0)/(2,0,4,2,0) -> idx [0,5,6,3,0,8,8,8,8]
    b1 = np.array([[0, 0, 4, 2, 0]], np.float32)
    b2 = np.array([[2, 0, 4, 2, 0]], np.float32)
    _, idx, nv, _ = oracle.iou_pairs(b1, b2, tsc(b1[:, 4]), tsc(b2[:, 4]), debug=True)
    assert nv[0] == 4 and idx[0].tolist() == [0, 5, 6, 3, 0, 8, 8, 8, 8]


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_scene_collision(oracle, metric):
    g = load_golden('g2_scene_collision.npz')
    sc_of = (lambda b: tsc(b[..., 4])) if metric == 'iou' else tsc_discs
    cases = [(boxes_of(g['state'], g['size']), g['present'], None, g['coll_' + metric]),
             (np.concatenate([boxes_of(g['state'], g['size']), boxes_of(g['npc_state'], g['npc_size'])], 1),
              np.concatenate([g['present'], g['npc_present']], 1), 8, g['coll_npc_' + metric]),
             (boxes_of(g['far_state'], g['far_size']), g['far_present'], None, g['far_coll_' + metric])]
    for boxes, present, nexp, ref in cases:
        out = oracle.collision(boxes, present, n_exposed=nexp, metric=metric, sc=sc_of(np.nan_to_num(boxes)))
        # sum over agents is order dependent in fp32 (torch's vectorised reduction vs sequential): 4 ulp of O(1)
        np.testing.assert_allclose(out, ref, atol=5e-7, rtol=0)
        np.testing.assert_array_equal(out > 0, ref > 0)           # collision mask, bit-exact


def test_collision_nonpresent_quirk(oracle):
    # present=[F,T,T] for three boxes, two coincident -> [0,0,0]; [T,T,T] -> [1/3.. ] pattern
    boxes = np.array([[[0, 0, 4, 2, 0], [2, 0, 4, 2, 0], [30, 30, 4, 2, 0]]], np.float32)
    sc = tsc(boxes[..., 4])
    out_t = oracle.collision(boxes, np.array([[1, 1, 1]]), sc=sc)
    out_f = oracle.collision(boxes, np.array([[0, 1, 1]]), sc=sc)
    np.testing.assert_allclose(out_t[0], [1 / 3, 1 / 3, 0], atol=1e-6)
    np.testing.assert_allclose(out_f[0], [0, 0, 0], atol=1e-6)


# ---------------------------------------------------------------- G3
def test_offroad_bit_exact(oracle):
    g = load_golden('g3_offroad.npz')
    for thr, key in ((0.5, 'a_off_t05'), (0.0, 'a_off_t0')):
        o = oracle.offroad(g['a_state'], g['a_lenwid'], g['a_verts'], g['a_faces'], threshold=thr, sc=tsc(g['a_state'][..., 2]))
        np.testing.assert_array_equal(o, g[key])
    np.testing.assert_allclose(g['a_off_t05'][0, 0], 414.952, rtol=1e-6)    # SURVEY R4 probe value
    for thr, key in ((0.5, 'b_off_t05'), (0.0, 'b_off_t0')):
        o = oracle.offroad(g['b_state'], g['b_lenwid'], g['b_verts'], g['b_faces'], threshold=thr, sc=tsc(g['b_state'][..., 2]))
        np.testing.assert_array_equal(o, g[key])       # includes collate-padded [0,0,0] faces
    o = oracle.offroad(g['b_state'], g['b_lenwid'], g['b_verts'], g['b_faces'], threshold=0.5, sc=tsc(g['b_state'][..., 2]),
                       present=g['c_present'])
    np.testing.assert_array_equal(o, g['c_sim_offroad'])


# ---------------------------------------------------------------- G4 / G5
def test_actor_template_bit_exact(oracle):
    g = load_golden('g45_mesh_preraster.npz')
    t = oracle.actor_template(g['g4_size'])
    np.testing.assert_array_equal(t.reshape(2, -1, 2), g['g4_tmpl_verts'][..., :2])


def static_arrays(oracle, g, m):
    town = load_golden('town01_mesh.npz')
    cats = [str(c) for c in town['categories']]
    n = m['name']
    if m['road'] == 'town01':
        return oracle.static_mesh_arrays(town['verts'], town['faces'], town['vert_category'], cats)
    if m['road'] == 'crop':
        return oracle.static_mesh_arrays(g[f'g5_{n}_road_verts'], g[f'g5_{n}_road_faces'], g[f'g5_{n}_road_vert_category'], cats)
    return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)


def test_preraster_call_lists_match_reference(oracle):
    """Everything up to the OpenCV boundary (actor transform, camera shift, trim, z-order, projection,
    int truncation, colour quantisation) -- rendering/cv2.py:27-59 with a recording cv2."""
    g = load_golden('g45_mesh_preraster.npz')
    for m in json.loads(str(g['g5_meta'])):
        n = m['name']
        st, sz, pr = g[f'g5_{n}_state'], g[f'g5_{n}_size'], g[f'g5_{n}_present']
        B, A = st.shape[:2]
        sv, sa, sf = static_arrays(oracle, g, m)
        mask = np.broadcast_to(pr[:, None, :], (B, A, A))
        _, tris, cols, cnt = oracle.render_scenes(st, sz, mask, st[..., :2], g[f'g5_{n}_cam_sc'], sv, sa, sf, m['fov'], m['res'],
                                                  agent_sc=tsc(st[..., 2]), record=True, images=False)
        gt = g[f'g5_{n}_tris'].reshape(B * A, -1, 6)
        gc = g[f'g5_{n}_cols']
        for i in range(B * A):
            mine = [tuple(tris[i, k]) + tuple(cols[i, k]) for k in range(cnt[i])]
            ref = [tuple(gt[i, k]) + tuple(gc[i, k]) for k in range(gt.shape[1])]
            cm, cr = Counter(mine), Counter(ref)
            assert not (cm - cr), f'{n} image {i}: calls the reference never made'
            # the reference pads every image of a batch to the same face count with [0,0,0] faces
            # (mesh.py:314-317) = 1-pixel dots on an already-drawn vertex; nothing else may be extra
            for k in (cr - cm):
                assert k[0] == k[2] == k[4] and k[1] == k[3] == k[5], f'{n} image {i}: missing call {k}'
            # painter order: colour runs (= z levels) appear in the same order
            runs = lambda seq: [c for j, c in enumerate(seq) if j == 0 or c != seq[j - 1]]
            extra = dict(cr - cm)
            ref_wo_pad = []
            for k in ref:
                if extra.get(k, 0) > 0:
                    extra[k] -= 1
                else:
                    ref_wo_pad.append(k)
            assert runs([k[6:] for k in mine]) == runs([k[6:] for k in ref_wo_pad])


def test_untrimmed_call_lists_match_reference(oracle):
    """G15: CV2RendererConfig(trim_mesh_before_rendering=False) (rendering/cv2.py:15,32-41) -- the reference then hands EVERY face to
    fillConvexPoly (no padding: the call lists must be equal as multisets, painter-order runs included); with the rule on, the two huge
    triangles of this scene (no vertex in any view) are dropped."""
    g = load_golden('g15_preraster_untrimmed.npz')
    town = load_golden('town01_mesh.npz')
    cats = [str(c) for c in town['categories']]
    st, sz, pr = g['state'], g['size'], g['present']
    B, A = st.shape[:2]
    sv, sa, sf = oracle.static_mesh_arrays(g['road_verts'], g['road_faces'], g['road_vert_category'], cats)
    mask = np.broadcast_to(pr[:, None, :], (B, A, A))
    for name, trim in (('untrimmed', False), ('trimmed', True)):
        oracle.set_trim_mesh(trim)
        try:
            _, tris, cols, cnt = oracle.render_scenes(st, sz, mask, st[..., :2], g['cam_sc'], sv, sa, sf, float(g['fov']), int(g['res']),
                                                      agent_sc=tsc(st[..., 2]), record=True, images=False)
        finally:
            oracle.set_trim_mesh(True)
        gt, gc = g[f'{name}_tris'].reshape(B * A, -1, 6), g[f'{name}_cols']
        for i in range(B * A):
            mine = [tuple(tris[i, k]) + tuple(cols[i, k]) for k in range(cnt[i])]
            ref = [tuple(gt[i, k]) + tuple(gc[i, k]) for k in range(gt.shape[1])]
            cm, cr = Counter(mine), Counter(ref)
            assert not (cm - cr), f'{name} image {i}: calls the reference never made'
            for k in (cr - cm):                                  # trimmed lists are padded with [0,0,0] faces (mesh.py:314-317), untrimmed ones are not
                assert trim and k[0] == k[2] == k[4] and k[1] == k[3] == k[5], f'{name} image {i}: missing call {k}'
            runs = lambda seq: [c for j, c in enumerate(seq) if j == 0 or c != seq[j - 1]]
            assert runs([c[6:] for c in mine]) == runs([c[6:] for c in ref if not (trim and c[0] == c[2] == c[4] and c[1] == c[3] == c[5])] if trim else [c[6:] for c in ref]), name
        if not trim:
            assert all(cnt[i] == gt.shape[1] for i in range(B * A)) and gt.shape[1] == len(g['road_faces']) + 3 * A
