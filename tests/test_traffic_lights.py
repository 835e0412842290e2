"""
Traffic-light programmes (torchdrivesim_amd/traffic_lights.py) against (a) the known answers of the reference's own
tests/test_traffic_light_controller.py on the reference's own data files and (b) a replay of the reference's controller on Town01's
programmes (tests/golden/g12_traffic_lights.json, tools/gen_golden.py:gen_traffic_lights).  Host logic: runs without a GPU.
"""
import json
import os
import random

import pytest
import torch

from torchdrivesim_amd.traffic_lights import (TrafficLightController, TrafficLightStateMachine, TrafficLightState, TrafficLightGroupState,
                                              current_light_state_tensor_from_controller)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
R, Y, G = TrafficLightState.red, TrafficLightState.yellow, TrafficLightState.green


@pytest.fixture
def controller():
    folder = os.path.join(GOLD, 'traffic_lights', 'machines')
    files = sorted(os.path.join(folder, f) for f in os.listdir(folder) if f.endswith('.json'))
    return TrafficLightController([TrafficLightStateMachine.from_json(f) for f in files])


def _lights(**kw):
    base = {k: R for k in ('4411', '3411', '4399', '3399', '3404', '4404', '3406', '3405', '3403', '4403')}
    base.update({k.lstrip('_'): v for k, v in kw.items()})
    return base


def test_reset(controller):
    controller.reset()
    assert all(t > 0 for t in controller.time_remaining)


def test_set_to(controller):
    controller.set_to([[2, 4], [0, 1]])
    assert controller.time_remaining == [4, 1]
    assert controller.current_state == _lights(_4411=G, _4399=Y, _3399=Y)
    controller.set_to([[2, 3], [1, 0]])
    assert controller.time_remaining == [3, 0]
    assert controller.current_state == _lights(_4411=G, _4399=Y, _3399=Y, _3406=G, _3405=G)


def test_tick(controller):
    controller.set_to([[0, 10], [0, 2]])
    controller.tick(1)
    assert controller.current_state == _lights()
    controller.tick(1)
    assert controller.current_state == _lights(_3406=G, _3405=G)


def test_json_round_trip(controller, tmp_path):
    ctl = TrafficLightController.from_json(os.path.join(GOLD, 'traffic_lights', 'intersection_controller.json'))
    assert ctl.get_number_of_light_groups() == len(ctl.traffic_fsms) > 0
    path = tmp_path / 'c.json'
    path.write_text(controller.to_json())
    again = TrafficLightController.from_json(str(path))
    assert [f.states for f in again.traffic_fsms] == [f.states for f in controller.traffic_fsms]
    (tmp_path / 'bad.json').write_text(json.dumps([[{'actor_states': {'1': 'purple'}, 'state': 0, 'duration': 1, 'next_state': 0}]]))
    with pytest.raises(ValueError):
        TrafficLightController.from_json(str(tmp_path / 'bad.json'))


def test_replay_of_the_reference_on_town01():
    g = json.load(open(os.path.join(GOLD, 'g12_traffic_lights.json')))
    random.seed(g['seed'])                                       # the reference resets every programme with random.randint
    ctl = TrafficLightController.from_json(os.path.join(GOLD, 'maps', 'carla_Town01', 'carla_Town01_traffic_light_controller.json'))
    assert json.loads(ctl.to_json()) == g['to_json']

    def check(want):
        assert ctl.state_per_machine == want['state_per_machine']
        assert ctl.time_remaining == want['time_remaining']      # the same float arithmetic in the same order: exact
        assert ctl.current_state_with_name == want['names']
        assert current_light_state_tensor_from_controller(ctl, g['ids']).tolist() == want['tensor']
    check(g['trace'][0])
    for (op, arg), want in zip(g['script'], g['trace'][1:]):
        getattr(ctl, op)(arg)
        check(want)
    assert len(g['script']) > 70


def test_tick_spanning_several_phases_and_map_config():
    mk = lambda i, d, n: TrafficLightGroupState({'7': [R, G, Y][i]}, i, d, n)
    fsm = TrafficLightStateMachine([mk(0, 2.0, 1), mk(1, 3.0, 2), mk(2, 1.0, 0)])
    fsm.set_to(0, 0.5)
    fsm.tick(4.0)                                                # 0.5 of red, all 3 of green, 0.5 into yellow
    assert fsm.current_state.sequence_number == 2 and fsm.time_remaining == 0.5 and fsm.duration == 1.0
    fsm.tick(0.5)                                                # ends exactly: red starts in full
    assert fsm.current_state.sequence_number == 0 and fsm.time_remaining == 2.0
    fsm.set_to(9, 50.0)                                          # clamped to the last phase and to its duration
    assert fsm.current_state.sequence_number == 2 and fsm.time_remaining == 1.0
    from torchdrivesim_amd.map import load_map_config
    from torchdrivesim_amd.traffic_controls import TrafficLightControl
    cfg = load_map_config(os.path.join(GOLD, 'maps', 'carla_Town01', 'metadata.json'))
    ctl = cfg.traffic_light_controller
    ids = [s.actor_id for s in cfg.stoplines if s.agent_type == 'traffic_light']
    t = current_light_state_tensor_from_controller(ctl, ids)
    assert t.shape == (len(ids),) and t.dtype == torch.int64
    names = TrafficLightControl._default_allowed_states()
    assert [names[i] for i in t.tolist()] == [ctl.current_state_with_name[str(i)] for i in ids]


def test_state_machine_known_answers_of_the_reference_tests():
    """tests/test_traffic_light_state_machine.py of the reference, on its intersection_1.json"""
    fsm = TrafficLightStateMachine.from_json(os.path.join(GOLD, 'traffic_lights', 'machines', 'intersection_1.json'))
    ids = ('4411', '3411', '4399', '3399')
    phase = lambda colours, seq, dur, nxt: TrafficLightGroupState(dict(zip(ids, colours)), seq, dur, nxt)
    fsm.reset()
    assert fsm.time_remaining >= 1
    for left in (3, 1):
        fsm.set_to(2, time_remaining=left)
        assert fsm.time_remaining == left and fsm.current_state == phase((G, R, Y, Y), 2, 5, 3)
    fsm.set_to(0, 1)
    fsm.tick(0.9)
    assert fsm.time_remaining <= 0.1 and fsm.current_state == phase((R, R, R, R), 0, 10, 1)
    fsm.tick(0.1)
    assert fsm.time_remaining == 10 and fsm.current_state == phase((G, R, G, G), 1, 10, 2)
    for dt, want, left in ((23, phase((G, R, Y, Y), 2, 5, 3), 2), (25, phase((G, G, R, R), 3, 10, 4), 10), (45, phase((R, R, R, R), 0, 10, 1), 5)):
        fsm.set_to(0, 10)
        fsm.tick(dt)
        assert fsm.current_state == want and fsm.time_remaining == left
    fsm.set_to(4, 3)
    assert fsm.get_current_actor_states() == dict(zip(ids, (Y, Y, R, R)))
    assert json.loads(fsm.to_json())[2]['next_state'] == '3'
