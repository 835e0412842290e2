"""CPU-only tests of the host-side mirror of the reference interface (no kernel is launched): shapes and bookkeeping that
the reference's own tests check (tests/simulator/test_simulator.py:90-160, tests/test_mesh.py), mesh file formats."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden


def make_sim(B=2, A=3, npc=0):
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig, NPCController
    km = KinematicBicycle()
    km.set_params(lr=torch.ones(B, A))
    km.set_state(torch.arange(B * A * 4, dtype=torch.float32).reshape(B, A, 4))
    ctrl = None
    if npc:
        ctrl = NPCController(npc_size=torch.ones(B, npc, 2), npc_state=torch.zeros(B, npc, 4), npc_present_mask=torch.ones(B, npc, dtype=torch.bool))
    return Simulator(BirdviewMesh.empty(batch_size=B), km, torch.ones(B, A, 2), torch.ones(B, A, dtype=torch.bool),
                     TorchDriveConfig(renderer=HipRendererConfig()), npc_controller=ctrl)


def test_accessor_shapes():
    sim = make_sim(2, 3, npc=2)
    assert sim.get_state().shape == (2, 3, 4) and sim.get_agent_size().shape == (2, 3, 2)
    assert sim.get_all_agent_state().shape == (2, 5, 4) and sim.get_all_agent_present_mask().shape == (2, 5)
    assert sim.get_all_agents_absolute().shape == (2, 5, 6)
    assert sim.get_all_agents_relative().shape == (2, 3, 4, 6)
    assert sim.get_all_agents_relative(exclude_self=False).shape == (2, 3, 5, 6)
    assert sim.compute_wrong_way().shape == (2, 3) and not sim.compute_wrong_way().any()
    assert sim.agent_count == 3 and sim.npc_count == 2 and sim.action_size == 2


def test_relative_excludes_self_like_boolean_indexing():
    sim = make_sim(2, 4, npc=1)
    full = sim.get_all_agents_relative(exclude_self=False)
    rel = sim.get_all_agents_relative(exclude_self=True)
    keep = torch.cat([~torch.eye(4, dtype=torch.bool), torch.ones(4, 1, dtype=torch.bool)], -1).flatten()
    ref = full.flatten(-3, -2)[..., keep, :].reshape(2, 4, 4, 6)          # the reference's formulation (simulator.py:774-781)
    assert torch.equal(rel, ref)


def test_copy_extend_select():
    sim = make_sim(2, 3)
    other = sim.copy()
    other.set_state(torch.zeros(2, 3, 4))
    assert not torch.equal(other.get_state(), sim.get_state())          # copy is independent (test_simulator.py:90-95)
    ext = sim.extend(3, in_place=False)
    assert ext.batch_size == 6 and ext.get_state().shape == (6, 3, 4) and sim.batch_size == 2
    assert torch.equal(ext.get_state()[0], ext.get_state()[2]) and torch.equal(ext.get_state()[0], sim.get_state()[0])
    sel = sim[[1]]
    assert sel.batch_size == 1 and torch.equal(sel.get_state()[0], sim.get_state()[1])
    sim.set_state(torch.ones(2, 3, 4), mask=torch.tensor([[True, False, False], [False, False, True]]))
    assert sim.get_state()[0, 0].tolist() == [1, 1, 1, 1] and sim.get_state()[0, 1].tolist() == [4, 5, 6, 7]


def test_lanelet_maps_and_lane_features_are_carried():
    from torchdrivesim_amd.simulator import Simulator
    sim = make_sim()
    # lane features are carried through the batch plumbing and the observation model (simulator.py:335,418,439,464,498,829)
    from torchdrivesim_amd.lanelet2 import LaneFeatures
    from torchdrivesim_amd.observation_noise import MapObservationNoiseFromLog
    lf = LaneFeatures(dense_lane_features=torch.arange(2 * 5 * 4.0).reshape(2, 5, 4), dense_lane_features_mask=torch.ones(2, 5, dtype=torch.bool))
    logged = LaneFeatures(dense_lane_features=torch.zeros(2, 5, 4))
    s3 = Simulator(sim.road_mesh, sim.kinematic_model, sim.agent_size, sim.present_mask, sim.cfg, lane_features=lf,
                   observation_noise_model=MapObservationNoiseFromLog(noisy_lane_features=[logged]))
    assert s3.get_noisy_lane_features() is logged                 # internal_time 0: from the log
    s3.internal_time = 1
    assert s3.get_noisy_lane_features() is s3.lane_features       # log exhausted: the simulator's own
    big = s3.extend(2, in_place=False)
    assert big.lane_features.dense_lane_features.shape == (4, 5, 4) and big.lane_features.sparse_lane_features is None
    assert torch.equal(big.lane_features.dense_lane_features[::2], lf.dense_lane_features)
    assert torch.equal(s3[[1]].lane_features.dense_lane_features, lf.dense_lane_features[1:])
    # lanelet maps are in scope (tests/test_lanelet2.py); scenes without one give zeros, as the reference
    s2 = Simulator(sim.road_mesh, sim.kinematic_model, sim.agent_size, sim.present_mask, sim.cfg, lanelet_map=[None, None])
    assert s2.compute_wrong_way().shape == (2, 3) and not s2.compute_wrong_way().any()


def test_kinematic_bookkeeping_and_fit_action():
    from torchdrivesim_amd.kinematic import KinematicBicycle, BicycleModel, SimpleKinematicModel, OrientedKinematicModel
    assert BicycleModel is KinematicBicycle
    g = load_golden('g1_kinematic.npz')
    km = KinematicBicycle()
    km.set_params(lr=torch.from_numpy(g['lr']))
    km.set_state(torch.from_numpy(g['state']))
    np.testing.assert_allclose(km.fit_action(torch.from_numpy(g['future'])).numpy(), g['fit_bicycle'], atol=1e-6)
    km2 = KinematicBicycle(left_handed=True)
    km2.set_params(lr=torch.from_numpy(g['lr']))
    km2.set_state(torch.from_numpy(g['state']))
    np.testing.assert_allclose(km2.fit_action(torch.from_numpy(g['future'])).numpy(), g['fit_bicycle_lh'], atol=1e-6)
    for cls, key in ((SimpleKinematicModel, 'fit_simple'), (OrientedKinematicModel, 'fit_oriented')):
        m = cls()
        m.set_state(torch.from_numpy(g['state']))
        np.testing.assert_allclose(m.fit_action(torch.from_numpy(g['future'])).numpy(), g[key], rtol=1e-5, atol=1e-5)
    c = km.copy()
    assert c is not km and c.get_state() is km.get_state() and c.lr is km.lr
    km.extend(2)
    assert km.get_state().shape == (6, 7, 4) and km.lr.shape == (6, 7)
    km.select_batch_elements([0, 3])
    assert km.get_state().shape == (2, 7, 4)


def test_mesh_json_format_and_fill_attr():
    from torchdrivesim_amd.mesh import BirdviewMesh, set_colors_with_defaults
    from torchdrivesim_amd.rendering import get_default_color_map, get_default_rendering_levels
    m = BirdviewMesh.load(os.path.join(GOLDEN, 'town01_crop_small_mesh.json'))       # written by the reference's save()
    g = load_golden('g45_mesh_preraster.npz')
    np.testing.assert_array_equal(m.verts[0].numpy(), g['g4_bg_verts'])
    np.testing.assert_array_equal(m.faces[0].numpy(), g['g4_bg_faces'])
    assert m.categories == ['right_lane', 'left_lane', 'road']
    rgb = set_colors_with_defaults(m.clone(), get_default_color_map(), get_default_rendering_levels())
    ref = g['g4_gen_bg_attrs_z']                                                        # reference fill_attr output
    np.testing.assert_array_equal(rgb.attrs[0].numpy(), ref[:, :3])
    np.testing.assert_array_equal(rgb.verts[0, :, 2].numpy(), ref[:, 3])


def test_mesh_roundtrip_expand_concat_collate(tmp_path):
    from torchdrivesim_amd.mesh import BirdviewMesh, BaseMesh
    m = BirdviewMesh.load(os.path.join(GOLDEN, 'town01_crop_small_mesh.json'))
    p = tmp_path / 'm.json'
    m.save(str(p))
    m2 = BirdviewMesh.load(str(p))
    assert torch.equal(m.verts, m2.verts) and torch.equal(m.faces, m2.faces) and torch.equal(m.vert_category, m2.vert_category)
    e = m.expand(3)
    assert e.batch_size == 3 and e.verts.shape[1:] == m.verts.shape[1:]
    cc = BirdviewMesh.concat([m, m])
    assert cc.verts_count == 2 * m.verts_count and int(cc.faces.max()) == 2 * m.verts_count - 1
    small = BaseMesh(verts=torch.zeros(1, 2, 2), faces=torch.zeros(1, 1, 3, dtype=torch.long))
    col = BaseMesh.collate([BaseMesh(m.verts, m.faces), small])
    assert col.batch_size == 2 and col.faces_count == m.faces_count and (col.faces[1, 1:] == 0).all()     # padded [0,0,0] faces
    sep = m.separate_by_category()
    assert set(sep) == {'right_lane', 'left_lane', 'road'} and sum(v.faces_count for v in sep.values()) >= m.faces_count


def test_generator_matches_reference_generate():
    """BirdviewRGBMeshGenerator.generate (generic dataflow) against G4 from the reference (mesh.py:1053-1157)."""
    from torchdrivesim_amd.mesh import BirdviewMesh, BirdviewRGBMeshGenerator
    from torchdrivesim_amd.rendering import get_default_color_map, get_default_rendering_levels
    g = load_golden('g45_mesh_preraster.npz')
    bg = BirdviewMesh.load(os.path.join(GOLDEN, 'town01_crop_small_mesh.json')).expand(2)
    gen = BirdviewRGBMeshGenerator(bg, get_default_color_map(), get_default_rendering_levels())
    size, state, present = torch.from_numpy(g['g4_size']), torch.from_numpy(g['g4_state']), torch.from_numpy(g['g4_present'])
    gen.initialize_actors_mesh(size, torch.zeros(2, 4, dtype=torch.long), ['vehicle'])
    np.testing.assert_array_equal(gen.actor_mesh.verts.numpy(), g['g4_tmpl_verts'])
    np.testing.assert_array_equal(gen.actor_mesh.faces.numpy(), g['g4_tmpl_faces'])
    np.testing.assert_array_equal(gen.actor_mesh.attrs.numpy(), g['g4_tmpl_attrs'])
    rgb = gen.generate(4, agent_state=state[:, None].expand(-1, 4, -1, -1), present_mask=present[:, None].expand(2, 4, 4))
    nv, nf = g['g4_bg_verts'].shape[0], g['g4_bg_faces'].shape[0]
    np.testing.assert_array_equal(rgb.verts[:, nv:].numpy(), g['g4_gen_actor_verts'])
    np.testing.assert_array_equal(rgb.faces[:, nf:].numpy(), g['g4_gen_actor_faces'])       # masked agents alias the first actor vertex
    np.testing.assert_array_equal(rgb.attrs[:, nv:].numpy(), g['g4_gen_actor_attrs'])


def test_compound_npc_controller_merges_by_owner():
    """simulator.py:206-278: every NPC is advanced by the controller that owns it; all controllers see the merged scene"""
    from torchdrivesim_amd.simulator import NPCController, CompoundNPCController

    class Mover(NPCController):
        def __init__(self, *a, dx=0.0, **k):
            super().__init__(*a, **k)
            self.dx = dx

        def advance_npcs(self, simulator):
            self.npc_state = self.npc_state + torch.tensor([self.dx, 0.0, 0.0, 0.0])

        def copy(self):
            c = super().copy()
            c.dx = self.dx
            return c

    B, n = 2, 3
    size = torch.ones(B, n, 2)
    state = torch.arange(B * n * 4, dtype=torch.float32).reshape(B, n, 4)
    present = torch.tensor([[True, False, True], [True, True, True]])
    a, b = Mover(size, state.clone(), present.clone(), dx=1.0), Mover(size * 2, state.clone() + 100, ~present, dx=-1.0)
    owner = torch.tensor([[0, 1, 0], [1, 1, 0]])
    comp = CompoundNPCController([a, b], owner)
    want = torch.where((owner == 0).unsqueeze(-1), state, state + 100)
    assert torch.equal(comp.get_npc_state(), want) and a.npc_state is comp.npc_state and b.npc_state is comp.npc_state
    assert torch.equal(comp.get_npc_present_mask(), torch.where(owner == 0, present, ~present))
    assert torch.equal(comp.get_npc_size()[..., 0], torch.where(owner == 0, 1.0, 2.0))
    comp.advance_npcs(None)
    moved = want + torch.where((owner == 0).unsqueeze(-1), torch.tensor([1.0, 0, 0, 0]), torch.tensor([-1.0, 0, 0, 0]))
    assert torch.equal(comp.get_npc_state(), moved)
    big = comp.extend(2, in_place=False)
    assert big.get_npc_state().shape == (4, n, 4) and torch.equal(big.controller_indices, owner.repeat_interleave(2, 0))
    assert torch.equal(big.get_npc_state()[::2], moved) and comp.get_npc_state().shape == (B, n, 4)
    sel = comp.select_batch_elements(torch.tensor([1]), in_place=False)
    assert torch.equal(sel.get_npc_state(), moved[1:]) and torch.equal(sel.controller_indices, owner[1:])


def test_replay_controller_follows_the_log_and_wraps(tmp_path):
    """behavior/replay.py: state(t) = log[..., t mod T, :]; batch plumbing carries the whole log"""
    from torchdrivesim_amd.behavior import ReplayController, interaction_replay, InitializationFailedError
    B, n, T = 2, 3, 4
    log = torch.arange(B * n * T * 4, dtype=torch.float32).reshape(B, n, T, 4)
    pres = torch.rand(B, n, T, generator=torch.Generator().manual_seed(1)) < 0.7
    import types
    rc = ReplayController(torch.ones(B, n, 2), log, pres)
    sim_of = lambda c: types.SimpleNamespace(npc_controller=c)                # all the spawn controller touches
    for t in range(1, 2 * T + 1):
        rc.advance_npcs(sim_of(rc))
        assert rc.time == t % T and torch.equal(rc.get_npc_state(), log[:, :, t % T]) and torch.equal(rc.get_npc_present_mask(), pres[:, :, t % T])
    big = rc.extend(3, in_place=False)
    assert big.npc_states.shape == (6, n, T, 4) and torch.equal(big.get_npc_state()[::3], rc.get_npc_state())
    big.advance_npcs(sim_of(big))
    assert torch.equal(big.get_npc_state()[::3], log[:, :, 1]) and rc.time == 0 and big.time == 1
    sel = rc.select_batch_elements([1], in_place=False)
    assert sel.npc_states.shape == (1, n, T, 4) and torch.equal(sel.get_npc_state(), log[1:, :, 0])
    assert torch.equal(ReplayController(torch.ones(B, n, 2), log).get_npc_present_mask(), torch.ones(B, n, dtype=torch.bool))
    # INTERACTION-format recording: two tracks, the second one appears late
    d = tmp_path / 'recorded_trackfiles' / 'loc'
    d.mkdir(parents=True)
    rows = ['track_id,frame_id,timestamp_ms,agent_type,x,y,vx,vy,psi_rad,length,width']
    rows += [f'1,{f},{100 * f},car,{10 + f},5,3,4,0.1,4.5,1.8' for f in range(1, 6)]
    rows += [f'2,{f},{100 * f},car,{20 + f},7,0,2,-0.2,4.0,2.0' for f in range(3, 6)]
    (d / 'vehicle_tracks_000.csv').write_text('\n'.join(rows))
    attrs, states, present = interaction_replay('loc', str(tmp_path), initial_frame=2, segment_length=4)
    assert attrs.shape == (1, 2, 3) and states.shape == (1, 2, 4, 4) and present.shape == (1, 2, 4)
    assert attrs[0].tolist() == [[4.5, 1.8, 1.4], [4.0, 2.0, 1.4]]
    assert present[0].tolist() == [[True] * 4, [False, True, True, True]]
    assert states[0, 0, 0].tolist() == [12.0, 5.0, 0.1, 5.0] and states[0, 1, 0].tolist() == [0.0] * 4 and states[0, 1, 1].tolist() == [23.0, 7.0, -0.2, 2.0]
    with pytest.raises(InitializationFailedError):
        interaction_replay('loc', str(tmp_path), initial_frame=4, segment_length=10)


def test_a_configuration_for_the_opencv_backend_selects_the_rasteriser():
    from torchdrivesim_amd.rendering import CV2RendererConfig, HipRenderer, RendererConfig, renderer_from_config
    r = renderer_from_config(CV2RendererConfig(left_handed_coordinates=True))
    assert isinstance(r, HipRenderer) and r.cfg.left_handed_coordinates and r.cfg.out_dtype == 'float32'
    assert isinstance(renderer_from_config(RendererConfig()), HipRenderer)          # backend 'default'
    assert r.trim and not renderer_from_config(CV2RendererConfig(trim_mesh_before_rendering=False)).trim       # cv2.py:15, honoured since round 2


def test_heading_cache_follows_the_live_state():
    """the [sin, cos] cache must never serve an earlier state.  States arrive as fresh tensors with version 0 whose
    storage the allocator may recycle, so the cache is keyed on tensor identity (and keeps that tensor alive)."""
    sim = make_sim(2, 3)
    ref = lambda st: torch.stack([torch.sin(st[..., 2]), torch.cos(st[..., 2])], -1)
    g = torch.Generator().manual_seed(0)
    assert torch.equal(sim._heading_sc(), ref(sim.get_state()))
    first = sim._heading_sc()
    assert sim._heading_sc() is first                                   # same tensor, same version: served from the cache
    for step in range(12):                                              # states replaced WITHOUT a query in between, buffers recycled
        buf = torch.empty(2, 3, 4)
        buf.copy_(torch.rand(2, 3, 4, generator=g) * 6 - 3)
        sim.kinematic_model.set_state(buf)
        del buf
        if step % 3 == 2:
            assert torch.equal(sim._heading_sc(), ref(sim.get_state()))
    st = sim.get_state()
    sc0 = sim._heading_sc().clone()
    st[..., 2] += 1.0                                                   # in-place update of the same tensor bumps its version
    assert not torch.equal(sim._heading_sc(), sc0) and torch.equal(sim._heading_sc(), ref(st))
    # scenes with NPCs build the concatenated state per call: never cached, always current
    sim2 = make_sim(2, 3, npc=2)
    a = sim2._heading_sc()
    sim2.kinematic_model.set_state(sim2.get_state() + 0.5)
    assert torch.equal(sim2._heading_sc(), ref(sim2.get_all_agent_state())) and not torch.equal(sim2._heading_sc()[:, :3], a[:, :3])


def test_scene_cache_is_keyed_on_its_sources():
    """the device scene cache (static map, actor templates, keys) is stamped on the identity of the long-lived
    tensors it is derived from, not on the addresses of torch.cat temporaries."""
    sim = make_sim(2, 3, npc=2)
    built = []
    sim.renderer.scene_maps = lambda *a, **k: built.append(1) or type('M', (), dict(rank_of=lambda self, z: 1))()
    sim._scene(); sim._scene(); sim._scene()
    assert len(built) == 1                                              # NPC scenes: cat temporaries differ per call, the cache still holds
    sim.npc_controller.npc_size = sim.npc_controller.npc_size.clone()   # replaced NPC sizes: rebuilt
    sim._scene()
    assert len(built) == 2
    sim.agent_size.mul_(1.5)                                            # in-place edit of the sizes: rebuilt
    s = sim._scene()
    assert len(built) == 3 and torch.allclose(s['tmpl'][:, :3].abs().amax((1, 2, 3)), torch.full((2,), 0.75))


def test_reserved_cu_layout_is_judged_from_where_the_streams_really_ran():
    """overlap_infractions = 'reserved' (Simulator, loop matched: examples/gym_env.py:83-126 of the reference) relies on the meaning of the bits of a
    CU mask, observed on one MI355X in SPX mode.  _ops.check_reserved_layout judges the places a probe kernel reported (XCC_ID << 16 | SE/SH/CU):
    the documented layout passes; a layout whose first 32 bits are the 32 CUs of ONE XCD, overlapping streams, and a 32-CU partition do not."""
    from torchdrivesim_amd import _ops
    cu_words = [se << 5 | sh << 4 | cu for se in range(4) for sh in range(1) for cu in range(8)]        # 32 CUs of an XCD as HW_ID bits 15..8
    every = [(x << 16) | w for x in range(8) for w in cu_words]
    assert len(set(every)) == 256
    # MI355X, SPX: mask bit i = CU i / 8 of XCD i % 8 -> four CUs of every XCD on the metric stream
    metric = [(x << 16) | cu_words[k] for x in range(8) for k in range(4)]
    raster = [p for p in every if p not in set(metric)]
    assert _ops.check_reserved_layout(raster, metric, 4, 256) == (True, 'ok')
    # a part whose mask bits run through one XCD first: all 32 reserved CUs on XCD 0
    shifted = [p for p in every if p >> 16 == 0]
    ok, why = _ops.check_reserved_layout([p for p in every if p >> 16 != 0], shifted, 4, 256)
    assert not ok and 'per XCD' in why
    # masks that are not honoured (both streams everywhere)
    ok, why = _ops.check_reserved_layout(every, metric, 4, 256)
    assert not ok and 'both streams' in why
    # a CPX partition: 32 CUs
    ok, why = _ops.check_reserved_layout(every[:28], every[28:32], 4, 32)
    assert not ok and '32 CUs' in why
    # CUs that neither stream reaches
    ok, why = _ops.check_reserved_layout(raster[:-8], metric, 4, 256)
    assert not ok and 'cover' in why


def test_reserved_probe_never_raises(monkeypatch):
    """reserved_layout_ok / Simulator._reserved_usable promise "never raises, falls back with ONE warning" -- also when creating or
    probing the CU-masked streams fails (here: no GPU at all, then a failing tds_stream_create), and a verdict taken under stream capture
    ('not verified') is not kept for good."""
    from torchdrivesim_amd import _ops
    monkeypatch.setattr(_ops, '_reserved_streams', {})
    ok, why = _ops.reserved_layout_ok(torch.device('cuda', 0))          # this container has no GPU: whatever fails, fails inside
    assert ok is False and isinstance(why, str) and why

    def boom(idx, per_xcd):
        raise RuntimeError('tds_stream_create failed (code -2): out of streams')
    monkeypatch.setattr(_ops, '_reserved_streams', {})
    monkeypatch.setattr(_ops, '_make_reserved_entry', boom)
    monkeypatch.setattr(torch.cuda, 'device', lambda idx: __import__('contextlib').nullcontext())
    ok, why = _ops.reserved_layout_ok(torch.device('cuda', 0))
    assert ok is False and 'out of streams' in why
    assert _ops.reserved_layout_ok(torch.device('cuda', 0)) == (ok, why)            # cached: the probe is not repeated on every render
    # a verdict taken under capture is re-judged outside capture
    calls = []
    monkeypatch.setattr(_ops, '_reserved_streams', {(0, 4): ('rs', 'ms', [], (True, 'not verified: created under stream capture'))})
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: False)
    monkeypatch.setattr(_ops, 'stream_places', lambda s: calls.append(s) or [])
    monkeypatch.setattr(_ops.nat, 'lib', lambda: type('L', (), dict(tds_device_cu_count=staticmethod(lambda idx, ref: 0)))())
    ok, why = _ops.reserved_layout_ok(torch.device('cuda', 0))
    assert calls == ['rs', 'ms'] and not why.startswith('not verified')


def test_map_cache_is_keyed_on_content_and_confirms_a_hit():
    """_ops.MapCache (round 6: one device map per DISTINCT mesh, shared by simulators, their copies, sub-batches and shards): a hit needs the same
    key AND the same bytes (the key carries hashes; the rows kept beside the map decide), the least recently used entries go beyond the capacity,
    a map the cache handed out is `shared` (close() leaves it to its other holders)."""
    from torchdrivesim_amd import _ops

    class FakeMap:
        _h = 1
        closed = False

        def _destroy(self):
            self.closed = True
        close = _ops.StaticMap.close

    cache = _ops.MapCache(capacity=2)
    rows = [torch.arange(6.0).reshape(2, 3), torch.arange(4)]
    built = []

    def build():
        built.append(FakeMap())
        return built[-1]

    a = cache.get(('render', 'cuda:0', (1, 2)), rows, build)
    assert cache.get(('render', 'cuda:0', (1, 2)), [r.clone() for r in rows], build) is a and len(built) == 1 and (cache.hits, cache.misses) == (1, 1)
    b = cache.get(('render', 'cuda:0', (1, 2)), [rows[0] + 1, rows[1]], build)          # the same hashes, other bytes: never served the wrong map
    assert b is not a and len(built) == 2
    a.close()
    assert a.shared and not a.closed                                                      # shared maps are not destroyed by one holder's close()
    cache.get(('render', 'cuda:0', (3, 4)), rows, build)
    cache.get(('offroad', 'cuda:0', (3, 4)), rows, build)
    assert len(cache._d) == 2 and len(built) == 4                                         # capacity 2: the oldest entries went
    cache.clear()
    assert len(cache._d) == 0
