"""
Traffic controls: static rectangular stop lines with a discrete state per element (torchdrivesim/traffic_controls.py).
Same class surface as the reference (`BaseTrafficControl`, `TrafficLightControl`, `StopSignControl`, `YieldControl`); the one
computation on the hot path -- which agents overlap a red light's stop line -- runs on K2a's box-intersection kernel.
"""
from typing import List, Optional

import torch
from torch import Tensor

from torchdrivesim_amd import _ops


def _rear_boxes(boxes: Tensor, rear_factor: float) -> Tensor:
    """The rear `rear_factor` part of every box as a box of its own: same heading and width, length * rear_factor, centre
    moved back by length * (1 - rear_factor) / 2 (box2corners_with_rear_factor, _iou_utils.py:302-341).  boxes (...,5) [x,y,l,w,psi]."""
    x, y, l, w, psi = boxes.unbind(-1)
    back = (l * (1 - rear_factor)) / 2
    return torch.stack([x - back * torch.cos(psi), y - back * torch.sin(psi), l * rear_factor, w, psi], dim=-1)


class BaseTrafficControl:
    """
    pos: BxNx5 stop lines [x, y, length (thickness), width, orientation]; allowed_states: names of the discrete states;
    replay_states: BxNxT indices into allowed_states replayed by `step` while t < T; mask: BxN, False for padding
    (traffic_controls.py:12-156).
    """

    def __init__(self, pos: Tensor, allowed_states: Optional[List[str]] = None, replay_states: Optional[Tensor] = None,
                 mask: Optional[Tensor] = None):
        self.pos = pos
        self.allowed_states = list(allowed_states) if allowed_states is not None else self._default_allowed_states()
        self.replay_states = replay_states if replay_states is not None else \
            torch.zeros(pos.shape[:2] + (0,), dtype=torch.long, device=pos.device)
        self.mask = mask if mask is not None else torch.ones(pos.shape[:2], dtype=torch.bool, device=pos.device)
        self.state = self.replay_states[..., 0] if self.total_replay_time > 0 else \
            torch.zeros(pos.shape[:2], dtype=torch.long, device=pos.device)

    @classmethod
    def _default_allowed_states(cls) -> List[str]:
        return ['none']

    @property
    def total_replay_time(self) -> int:
        return self.replay_states.shape[-1]

    @property
    def corners(self) -> Tensor:
        """BxNx4x2 corners of the stop lines; padding elements are parked far away (traffic_controls.py:31-33)"""
        c = _box_corners(self.pos)
        m = self.mask.to(c.dtype)[..., None, None]
        return c * m + (1 - m) * -1000

    def _tensors(self):
        return ('pos', 'replay_states', 'mask', 'state')

    def copy(self):
        other = self.__class__(pos=self.pos.clone(), allowed_states=list(self.allowed_states), replay_states=self.replay_states.clone(),
                               mask=self.mask.clone())
        other.state = self.state.clone()
        return other

    def to(self, device):
        for n in self._tensors():
            setattr(self, n, getattr(self, n).to(device))
        return self

    def extend(self, n: int, in_place: bool = True):
        """repeat every batch element n times (B -> B*n, element-major), as the Simulator does for n cameras"""
        tgt = self if in_place else self.copy()
        for name in tgt._tensors():
            t = getattr(tgt, name)
            setattr(tgt, name, t.unsqueeze(1).expand((t.shape[0], n) + t.shape[1:]).reshape((n * t.shape[0],) + t.shape[1:]))
        return tgt

    def select_batch_elements(self, idx: Tensor, in_place: bool = True):
        tgt = self if in_place else self.copy()
        for name in tgt._tensors():
            setattr(tgt, name, getattr(tgt, name)[idx])
        return tgt

    def set_state(self, state: Tensor) -> None:
        self.state = state

    def compute_state(self, time: int) -> Tensor:
        """state at `time` beyond the replayed history: by default the current state is kept"""
        return self.state

    def step(self, time: int) -> None:
        self.set_state(self.replay_states[..., time] if time < self.total_replay_time else self.compute_state(time))

    def compute_violation(self, agent_state: Tensor) -> Tensor:
        """agent_state BxAx5 [x,y,length,width,orientation] -> BxA bool; the base class reports none"""
        return torch.zeros(agent_state.shape[:2], dtype=torch.bool, device=agent_state.device)


def _box_corners(pos: Tensor) -> Tensor:
    if pos.is_cuda:
        return _ops.box2corners(pos)
    # host-side plumbing (construction on the CPU before .to(device)): the same formula in torch (_iou_utils.py:270-299)
    x4 = torch.tensor([0.5, -0.5, -0.5, 0.5], dtype=pos.dtype) * pos[..., 2:3]
    y4 = torch.tensor([0.5, 0.5, -0.5, -0.5], dtype=pos.dtype) * pos[..., 3:4]
    s, c = torch.sin(pos[..., 4:5]), torch.cos(pos[..., 4:5])
    return torch.stack([x4 * c - y4 * s + pos[..., 0:1], x4 * s + y4 * c + pos[..., 1:2]], dim=-1)


class TrafficLightControl(BaseTrafficControl):
    """States ['red', 'yellow', 'green'].  An agent violates a light when the light is red and the rear tenth of the agent's box
    overlaps the stop line, i.e. it has driven (almost) completely over it (traffic_controls.py:159-194)."""
    violation_rear_factor = 0.1

    @classmethod
    def _default_allowed_states(cls) -> List[str]:
        return ['red', 'yellow', 'green']

    def compute_violation(self, agent_state: Tensor) -> Tensor:
        B, A = agent_state.shape[:2]
        N = self.pos.shape[1]
        if B == 0 or A == 0 or N == 0:
            return torch.zeros(B, A, dtype=torch.bool, device=agent_state.device)
        rear = _rear_boxes(agent_state, self.violation_rear_factor)                       # B x A x 5
        # padding stop lines are parked at (-1000, -1000) like the reference's corners (traffic_controls.py:31-33)
        key = (self.pos.data_ptr(), self.pos._version, tuple(self.pos.shape), self.mask.data_ptr(), self.mask._version, str(rear.device), A)
        cached = getattr(self, '_lines_cache', None)
        if cached is None or cached[0] != key:              # the stop lines do not move: their (B, A*N, 5) operand is built once
            m = self.mask.to(self.pos.dtype)[..., None]
            park = torch.tensor([-1000.0, -1000.0, 0.0, 0.0, 0.0], dtype=self.pos.dtype).to(self.pos.device)
            lines = (self.pos * m + (1 - m) * park).to(rear.device)
            cached = (key, lines[:, None, :, :].expand(B, A, N, 5).reshape(B, A * N, 5).contiguous())
            self._lines_cache = cached
        b1 = rear[:, :, None, :].expand(B, A, N, 5).reshape(B, A * N, 5)
        b2 = cached[1]
        overlap = _ops.pairwise_overlap(b1.contiguous(), b2.contiguous(), metric='iou').reshape(B, A, N) > 0
        red = (self.state.to(overlap.device) == self.allowed_states.index('red'))[:, None, :]
        return (overlap & red).any(dim=-1)


class YieldControl(BaseTrafficControl):
    """Yield sign: cross traffic has priority.  No violations are computed."""


class StopSignControl(BaseTrafficControl):
    """Stop sign.  No violations are computed."""
