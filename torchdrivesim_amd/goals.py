"""
Waypoint goals with the reference's surface (torchdrivesim/goals.py:11-217): every agent has `N` successive collections of `M`
waypoints; the collection `state[b, a]` is current, and it is ticked off as soon as the agent comes within `threshold` of any of its
(valid) waypoints.  Host-side torch bookkeeping on small tensors -- no kernel of its own; the waypoints of the current
collections are drawn by the K3 rasteriser as per-camera discs (`Simulator.render`, mesh.py:1120-1145 in the reference).
"""
from typing import Optional

import torch
from torch import Tensor


class WaypointGoal:
    """waypoints: B x A x N x M x 2 (x, y); mask: B x A x N x M bool, False = padding (default: all True)."""

    def __init__(self, waypoints: Tensor, mask: Optional[Tensor] = None):
        self.waypoints = waypoints
        self.mask = mask if mask is not None else torch.ones(waypoints.shape[:-1], dtype=torch.bool, device=waypoints.device)
        self.max_goal_idx = waypoints.shape[2]
        self.state = torch.zeros(waypoints.shape[:2] + (1,), dtype=torch.long, device=waypoints.device)      # B x A x 1

    # ---- the `count` collections starting at the current one, flattened to count*M; collections past the end are zeros / False
    def _window(self, count: int):
        idx = self.state + torch.arange(count, device=self.state.device).view(1, 1, -1)                       # B x A x count
        return idx.clamp(0, self.max_goal_idx - 1), idx < self.max_goal_idx

    def get_masks(self, count: int = 1) -> Tensor:
        """B x A x count*M (goals.py:33-69)"""
        idx, valid = self._window(count)
        M = self.mask.shape[3]
        got = torch.gather(self.mask, 2, idx[..., None].expand(-1, -1, -1, M)) & valid[..., None]
        return got.reshape(got.shape[:2] + (count * M,))

    def get_waypoints(self, count: int = 1) -> Tensor:
        """B x A x count*M x 2 (goals.py:71-107)"""
        idx, valid = self._window(count)
        M = self.waypoints.shape[3]
        got = torch.gather(self.waypoints, 2, idx[..., None, None].expand(-1, -1, -1, M, 2))
        got = torch.where(valid[..., None, None], got, torch.zeros_like(got))
        return got.reshape(got.shape[:2] + (count * M, 2))

    # ---- batch plumbing (goals.py:109-160)
    def copy(self):
        other = self.__class__(waypoints=self.waypoints.clone(), mask=self.mask.clone())
        other.state = self.state.clone()
        return other

    def to(self, device):
        self.waypoints, self.mask, self.state = self.waypoints.to(device), self.mask.to(device), self.state.to(device)
        return self

    def extend(self, n: int, in_place: bool = True):
        target = self if in_place else self.copy()
        rep = lambda x: x.unsqueeze(1).expand((x.shape[0], n) + x.shape[1:]).reshape((n * x.shape[0],) + x.shape[1:])
        target.waypoints, target.mask, target.state = rep(target.waypoints), rep(target.mask), rep(target.state)
        return target

    def select_batch_elements(self, idx, in_place: bool = True):
        target = self if in_place else self.copy()
        target.waypoints, target.mask, target.state = target.waypoints[idx], target.mask[idx], target.state[idx]
        return target

    def step(self, agent_states: Tensor, time: int = 0, threshold: float = 2.0) -> None:
        """Tick off the current collection of every agent that is within `threshold` of one of its valid waypoints: its valid
        entries become False and the state moves on, saturating at the last collection (goals.py:162-217)."""
        assert agent_states.shape[1] == self.waypoints.shape[1]
        wp, valid = self.get_waypoints(), self.get_masks()                               # B x A x M (x 2)
        dx, dy = agent_states[..., None, 0] - wp[..., 0], agent_states[..., None, 1] - wp[..., 1]
        near = ((dx ** 2) + (dy ** 2)) ** 0.5 <= threshold
        reached = (near & valid).any(dim=-1, keepdim=True)                               # B x A x 1
        idx = self.state[..., None].expand(-1, -1, -1, self.mask.shape[-1])              # B x A x 1 x M
        self.mask = self.mask.scatter(2, idx, (valid & ~reached).unsqueeze(2))           # padding entries stay False
        self.state = (self.state + reached).clamp(0, self.max_goal_idx - 1)
