"""
Log replay for non-playable agents (reference behavior/replay.py): the NPC states of every time step are given up front as
`(B, Npc, T, 4)` (+ presence `(B, Npc, T)`), resident on the device; advancing is a pair of views into them, so the NPC rows that
`Simulator.get_all_agent_state` concatenates behind the exposed agents (simulator.py:706-728) cost no copy per step.
"""
import os
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor

from torchdrivesim_amd.behavior.common import InitializationFailedError
from torchdrivesim_amd.simulator import NPCController, SpawnController, _enlarge


def interaction_replay(location, dataset_path, initial_frame=1, segment_length=40, recording=0):
    """A segment of an INTERACTION-dataset recording as `(attributes (1,N,3) [length, width, rear offset], states (1,N,T,4),
    present (1,N,T))` (replay.py:13-47); tracks absent in a frame are padded with zeros and marked not present."""
    import pandas as pd
    path = os.path.join(dataset_path, 'recorded_trackfiles', location, 'vehicle_tracks_{:03d}.csv'.format(recording))
    df = pd.read_csv(path)
    final_frame = initial_frame + segment_length - 1
    frames_on_file = set(df.frame_id.unique())
    for frame in (initial_frame, final_frame):
        if frame not in frames_on_file:
            raise InitializationFailedError(f'Frame {frame} not available in {path}')
    df = df[(df.frame_id >= initial_frame) & (df.frame_id <= final_frame)].sort_values(['track_id', 'frame_id'])
    tracks, frames = sorted(df.track_id.unique()), sorted(df.frame_id.unique())
    row = {t: i for i, t in enumerate(tracks)}
    col = {f: j for j, f in enumerate(frames)}
    states = np.zeros((len(tracks), len(frames), 4), np.float64)
    present = np.zeros((len(tracks), len(frames)), bool)
    i, j = df.track_id.map(row).to_numpy(), df.frame_id.map(col).to_numpy()
    states[i, j] = np.stack([df.x, df.y, df.psi_rad, np.sqrt(df.vx ** 2 + df.vy ** 2)], -1)
    present[i, j] = True
    sums = np.zeros((len(tracks), 2))
    np.add.at(sums, i, df[['length', 'width']].to_numpy(np.float64))
    attrs = np.concatenate([sums / np.bincount(i, minlength=len(tracks))[:, None], np.full((len(tracks), 1), 1.4)], -1)
    return torch.from_numpy(attrs).unsqueeze(0), torch.from_numpy(states).unsqueeze(0), torch.from_numpy(present).unsqueeze(0)


class ReplayController(NPCController):
    """NPCs that follow a log, wrapping around at its end (replay.py:50-107)."""

    def __init__(self, npc_size, npc_states, npc_present_masks: Optional[Tensor] = None, time: int = 0, npc_types: Optional[Tensor] = None,
                 agent_type_names: Optional[List[str]] = None, spawn_controller: Optional[SpawnController] = None):
        self.time = time
        self.npc_states = npc_states
        self.npc_present_masks = npc_present_masks if npc_present_masks is not None else torch.ones_like(npc_states[..., 0], dtype=torch.bool)
        super().__init__(npc_size, self.npc_states[..., time, :], self.npc_present_masks[..., time], npc_types, agent_type_names, spawn_controller)

    def advance_npcs(self, simulator) -> None:
        self.time = (self.time + 1) % self.npc_states.shape[-2]
        self.npc_state = self.npc_states[..., self.time, :]
        self.npc_present_mask = self.npc_present_masks[..., self.time]
        self.spawn_despawn_npcs(simulator)

    def _map(self, f):
        self.npc_states, self.npc_present_masks = f(self.npc_states), f(self.npc_present_masks)
        return super()._map(f)

    def copy(self):
        other = self.__class__(self.npc_size, self.npc_states, self.npc_present_masks, self.time, self.npc_types, self.agent_type_names,
                               self.spawn_controller.copy())
        other.npc_state, other.npc_present_mask = self.npc_state.clone(), self.npc_present_mask.clone()     # despawned NPCs stay despawned
        return other
