"""
Kinematic models with the reference's plugin surface (torchdrivesim/kinematic.py:20-157): `step` replaces the state
tensor with a new one computed by the K1 HIP kernels (torchdrivesim_amd/csrc/kinematic.hip) through autograd Functions,
so gradients chain through time exactly like the reference's torch graph.  `fit_action` and the bookkeeping methods are
host-side torch ops (not on the hot path).
"""
from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from torchdrivesim_amd import _ops
from torchdrivesim_amd.utils import rotate


def _enlarge(x: Tensor, n: int) -> Tensor:
    return x.unsqueeze(1).expand((x.shape[0], n) + x.shape[1:]).reshape((n * x.shape[0],) + x.shape[1:])


class KinematicModel(ABC):
    """Batch-mode kinematic model; subclasses define `action_size`, `step` and `fit_action` (kinematic.py:20-157)."""
    state_size: int = 4      #: x, y, orientation, speed
    action_size: int = 4

    def __init__(self, dt: float = 0.1):
        self.dt = dt
        self.state = None

    @property
    def batch_size(self) -> int:
        return self.get_state()[..., 0].numel()

    @abstractmethod
    def step(self, action: Tensor, dt: Optional[float] = None) -> None:
        raise NotImplementedError

    @abstractmethod
    def fit_action(self, future_state: Tensor, current_state: Optional[Tensor] = None, dt: Optional[float] = None) -> Tensor:
        raise NotImplementedError

    def copy(self, other=None):
        """Shallow copy: tensors are shared, the object is new (kinematic.py:67-76)."""
        if other is None:
            other = self.__class__(dt=self.dt)
        other.set_params(**self.get_params())
        other.set_state(self.get_state())
        return other

    def to(self, device):
        if self.state is not None:
            self.state = self.state.to(device)
        self.map_param(lambda x: x.to(device))
        return self

    def set_state(self, state: Tensor) -> None:
        self.state = state

    def get_state(self) -> Tensor:
        return self.state

    def get_params(self) -> Dict[str, Tensor]:
        return dict()

    def set_params(self, **kwargs) -> None:
        pass

    def flattening(self, batch_shape) -> None:
        pass

    def unflattening(self, batch_shape) -> None:
        pass

    def map_param(self, f) -> None:
        pass

    def normalize_action(self, action: Tensor) -> Tensor:
        return action

    def denormalize_action(self, action: Tensor) -> Tensor:
        return action

    @staticmethod
    def pack_state(x: Tensor, y: Tensor, psi: Tensor, speed: Tensor) -> Tensor:
        return torch.stack([x, y, psi, speed], dim=-1)

    @staticmethod
    def unpack_state(state: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        return state[..., 0], state[..., 1], state[..., 2], state[..., 3]

    def extend(self, n: int):
        self.map_param(lambda x: _enlarge(x, n))
        self.set_state(_enlarge(self.get_state(), n))

    def select_batch_elements(self, idx):
        self.map_param(lambda x: x[idx])
        self.set_state(self.get_state()[idx])


class TeleportingKinematicModel(KinematicModel):
    """The action is the next state (kinematic.py:317-325)."""

    def step(self, action, dt=None):
        self.set_state(action)

    def fit_action(self, future_state, current_state=None, dt=None):
        return future_state


class SimpleKinematicModel(KinematicModel):
    """Action = time derivative of the state in units of the constructor arguments (kinematic.py:328-376)."""

    def __init__(self, max_dx=20, max_dpsi=10 * np.pi, max_dv=5, dt=0.1):
        super().__init__(dt=dt)
        self.max_dx, self.max_dpsi, self.max_dv = max_dx, max_dpsi, max_dv
        self._normalization_factor = torch.tensor([max_dx, max_dx, max_dpsi, max_dv])
    _oriented = False

    def copy(self, other=None):
        if other is None:
            other = self.__class__(max_dx=self.max_dx, max_dv=self.max_dv, dt=self.dt)
        other._normalization_factor = self._normalization_factor.clone()    # behaviour follows the tensor
        return super().copy(other)

    def to(self, device):
        super().to(device)
        self._normalization_factor = self._normalization_factor.to(device)
        return self

    def normalize_action(self, action):
        return action / self._normalization_factor.to(action.device)

    def denormalize_action(self, action):
        return action * self._normalization_factor.to(action.device)

    def step(self, action, dt=None):
        dt = self.dt if dt is None else dt
        assert action.shape[-1] == self.action_size
        norm = [float(x) for x in self._normalization_factor.detach().cpu()]
        self.set_state(_ops.simple_step(self.get_state(), action, dt=dt, norm=norm, oriented=self._oriented))

    def fit_action(self, future_state, current_state=None, dt=None):
        dt = self.dt if dt is None else dt
        if current_state is None:
            current_state = self.get_state()
        return self.normalize_action((future_state - current_state) / dt)


class OrientedKinematicModel(SimpleKinematicModel):
    """Like SimpleKinematicModel but the action frame rotates with the agent (kinematic.py:379-397)."""
    _oriented = True

    def fit_action(self, future_state, current_state=None, dt=None):
        parent = super().fit_action(future_state, current_state=current_state, dt=dt)
        if current_state is None:
            current_state = self.get_state()
        xy = rotate(parent[..., :2], -current_state[..., 2:3])
        return torch.cat([xy, parent[..., 2:]], dim=-1)


class KinematicBicycle(KinematicModel):
    """Kinematic bicycle with steering applied at the geometric centre; parameter `lr` = distance centre -> rear axle;
    action = (acceleration, steering) normalised by the constructor arguments (kinematic.py:400-506)."""
    action_size: int = 2
    _no_reversing = False

    def __init__(self, max_acceleration=5, max_steering=np.pi / 2, dt=0.1, left_handed=False):
        super().__init__(dt=dt)
        self.max_acceleration = max_acceleration
        self.max_steering = max_steering
        self.left_handed = left_handed
        self._normalization_factor = torch.tensor([self.max_acceleration, self.max_steering])
        self.lr = None

    def copy(self, other=None):
        if other is None:
            other = self.__class__(max_acceleration=self.max_acceleration, dt=self.dt, left_handed=self.left_handed)
        other._normalization_factor = self._normalization_factor.clone()
        return super().copy(other)

    def to(self, device):
        super().to(device)
        self._normalization_factor = self._normalization_factor.to(device)
        return self

    def get_params(self):
        params = super().get_params()
        params['lr'] = self.lr
        return params

    def set_params(self, **kwargs):
        assert 'lr' in kwargs
        self.lr = kwargs['lr']

    def flattening(self, batch_shape):
        assert self.lr is not None
        self.lr = self.lr.reshape((int(np.prod(batch_shape)),))

    def unflattening(self, batch_shape):
        assert self.lr is not None
        self.lr = self.lr.reshape(batch_shape)

    def map_param(self, f):
        assert self.lr is not None
        self.lr = f(self.lr)

    def normalize_action(self, action):
        return action / self._normalization_factor.to(action.device)

    def denormalize_action(self, action):
        return action * self._normalization_factor.to(action.device)

    def step(self, action, dt=None):
        assert action.shape[-1] == 2, 'The bicycle model takes as input only acceleration and steering'
        dt = self.dt if dt is None else dt
        nf = self._normalization_factor.detach().cpu()
        state = self.get_state()
        lr = self.lr.expand(state.shape[:-1]) if self.lr.shape != state.shape[:-1] else self.lr
        self.set_state(_ops.bicycle_step(state, action, lr, dt=dt, max_acc=float(nf[0]), max_steer=float(nf[1]),
                                         left_handed=self.left_handed, no_reversing=self._no_reversing))

    def fit_action(self, future_state, current_state=None, dt=None):
        dt = self.dt if dt is None else dt
        f_x, f_y, _, _ = self.unpack_state(future_state)
        c_x, c_y, c_psi, c_v = self.unpack_state(current_state if current_state is not None else self.get_state())
        vx, vy = (f_x - c_x) / dt, (f_y - c_y) / dt
        speed = torch.sqrt(vx ** 2 + vy ** 2)
        # steering is taken modulo 2 pi and forced to 0 when the agent does not move (kinematic.py:491-495)
        beta = torch.atan2(vy, vx) - c_psi * torch.sign(torch.abs(speed))
        beta = torch.remainder(beta + np.pi, 2 * np.pi) - np.pi
        reversing = torch.sign(torch.cos(beta)) == -1
        v = speed * torch.where(reversing, -1, 1)
        beta = torch.where(reversing, beta - np.pi * torch.sign(beta), beta)
        a = (v - c_v) / dt
        if self.left_handed:
            beta = -beta
        return self.normalize_action(torch.stack([a, beta], dim=-1))


#: the north star's name for the bicycle model
BicycleModel = KinematicBicycle


class BicycleNoReversing(KinematicBicycle):
    """Bicycle that comes to a full stop instead of reversing (kinematic.py:509-523)."""
    _no_reversing = True


class BicycleByDisplacement(KinematicBicycle):
    """Bicycle driven by a directed velocity (kinematic.py:526-567): the action (dx, dy) * max_dx is turned into the bicycle action that
    reaches `xy + (dx, dy) dt` (KinematicBicycle.fit_action) and stepped by K1."""

    def __init__(self, max_dx=20, dt=0.1):
        super().__init__(dt=dt)
        self.max_dx = max_dx
        self._xy_normalization_tensor = torch.tensor([self.max_dx, self.max_dx])

    def copy(self, other=None):
        if other is None:
            other = self.__class__(max_dx=self.max_dx, dt=self.dt)
        other._xy_normalization_tensor = self._xy_normalization_tensor.clone()
        return super().copy(other)

    def to(self, device):
        super().to(device)
        self._xy_normalization_tensor = self._xy_normalization_tensor.to(device)
        return self

    def step(self, action, dt=None):
        assert action.shape[-1] == 2        # x and y displacement
        self.step_from_xy(action[..., :2], dt=dt)

    def step_from_xy(self, xy, dt=None):
        dt = self.dt if dt is None else dt
        action = xy * self._xy_normalization_tensor.to(xy.device).to(xy.dtype)
        dx, dy = action[..., 0], action[..., 1]
        x, y, psi, v = self.unpack_state(self.get_state())
        # the bicycle fit ignores psi and v of the target; it is made with the model's OWN dt whatever `dt` is (kinematic.py:556)
        bicycle_action = KinematicBicycle.fit_action(self, self.pack_state(x + dx * dt, y + dy * dt, psi, v))
        KinematicBicycle.step(self, bicycle_action, dt=dt)

    def fit_action(self, future_state, current_state=None, dt=None):
        dt = self.dt if dt is None else dt
        xf, yf, _, _ = self.unpack_state(future_state)
        xp, yp, _, _ = self.unpack_state(self.get_state() if current_state is None else current_state)
        action = torch.stack([(xf - xp) / dt, (yf - yp) / dt], dim=-1)
        return action / self._xy_normalization_tensor.to(action.device)


class BicycleByOrientedDisplacement(BicycleByDisplacement):
    """BicycleByDisplacement with the displacement given in the agent's own frame (kinematic.py:570-587)."""

    def step_from_xy(self, xy, dt=None):
        psi = self.get_state()[..., 2:3]
        super().step_from_xy(rotate(xy, psi), dt=dt)

    def fit_action(self, future_state, current_state=None, dt=None):
        action = super().fit_action(future_state, current_state=current_state, dt=dt)
        if current_state is None:
            current_state = self.get_state()
        return rotate(action[..., :2], -current_state[..., 2:3])


class UnicycleModel(KinematicModel):
    """Unicycle named by the north star (absent from the reference): action = (acceleration, yaw rate),
    v += a dt; x += v cos(psi) dt; y += v sin(psi) dt; psi += w dt."""
    action_size: int = 2

    def __init__(self, max_acceleration=5, max_yaw_rate=1.0, dt=0.1):
        super().__init__(dt=dt)
        self.max_acceleration, self.max_yaw_rate = max_acceleration, max_yaw_rate
        self._normalization_factor = torch.tensor([float(max_acceleration), float(max_yaw_rate)])

    def copy(self, other=None):
        if other is None:
            other = self.__class__(max_acceleration=self.max_acceleration, max_yaw_rate=self.max_yaw_rate, dt=self.dt)
        return super().copy(other)

    def to(self, device):
        super().to(device)
        self._normalization_factor = self._normalization_factor.to(device)
        return self

    def normalize_action(self, action):
        return action / self._normalization_factor.to(action.device)

    def denormalize_action(self, action):
        return action * self._normalization_factor.to(action.device)

    def step(self, action, dt=None):
        assert action.shape[-1] == 2
        dt = self.dt if dt is None else dt
        self.set_state(_ops.unicycle_step(self.get_state(), action, dt=dt, max_acc=self.max_acceleration, max_yaw_rate=self.max_yaw_rate))

    def fit_action(self, future_state, current_state=None, dt=None):
        dt = self.dt if dt is None else dt
        cur = current_state if current_state is not None else self.get_state()
        a = (future_state[..., 3] - cur[..., 3]) / dt
        w = (future_state[..., 2] - cur[..., 2]) / dt
        return self.normalize_action(torch.stack([a, w], dim=-1))


class CompoundKinematicModel(KinematicModel):
    """Host-side dispatch over several models by splitting the batch (kinematic.py:160-314)."""

    def __init__(self, models: List[KinematicModel], model_assignments: Tensor, dt: float = 0.1):
        super().__init__(dt=dt)
        self.models = models
        self.model_assignments = model_assignments
        self.state_size = max(m.state_size for m in models)
        self.action_size = max(m.action_size for m in models)
        sizes = [m.batch_size for m in models]
        picked = [int((self.batch_assignments == i).sum()) for i in range(len(models))]
        if sizes != picked:
            raise ValueError(f'Batch sizes of models do not match how many elements are assigned to them: {sizes} vs {picked}')
        self.get_params()       # duplicate parameter names are an error

    @property
    def batch_assignments(self) -> Tensor:
        return self.model_assignments.flatten()

    @property
    def batch_size(self) -> int:
        return len(self.batch_assignments)

    @property
    def batch_shape(self):
        return self.model_assignments.shape

    def _split(self, x: Tensor, width_of) -> List[Tensor]:
        flat = x.flatten(0, -2)
        return [flat[self.batch_assignments == i, :width_of(m)] for i, m in enumerate(self.models)]

    def _merge(self, parts: List[Tensor], width: int, shape) -> Tensor:
        padded = [torch.nn.functional.pad(p, (0, width - p.shape[-1])) for p in parts]
        flat = torch.zeros_like(torch.cat(padded, dim=0))
        for i, p in enumerate(padded):
            flat[self.batch_assignments == i] = p
        return flat.reshape(tuple(shape) + (width,))

    def step(self, action: Tensor, dt: Optional[float] = None) -> None:
        for m, a in zip(self.models, self._split(action, lambda mm: mm.action_size)):
            m.step(a, dt=dt)

    def fit_action(self, future_state, current_state=None, dt=None):
        if current_state is None:
            current_state = self.get_state()
        fs, cs = self._split(future_state, lambda m: m.state_size), self._split(current_state, lambda m: m.state_size)
        acts = [m.fit_action(f, c, dt=dt) for m, f, c in zip(self.models, fs, cs)]
        return self._merge(acts, self.action_size, future_state.shape[:-1])

    def copy(self, other=None):
        if other is None:
            other = self.__class__(models=[m.copy() for m in self.models], model_assignments=self.model_assignments, dt=self.dt)
        other.set_params(**self.get_params())
        other.set_state(self.get_state())
        return other

    def to(self, device):
        for m in self.models:
            m.to(device)
        self.model_assignments = self.model_assignments.to(device)
        return self

    def extend(self, n: int):
        state = self.get_state()
        self.model_assignments = _enlarge(self.model_assignments, n)
        self.map_param(lambda x: _enlarge(x, n))
        self.set_state(_enlarge(state, n))

    def select_batch_elements(self, idx):
        self.model_assignments = self.model_assignments[idx]

    def set_state(self, state: Tensor) -> None:
        for m, s in zip(self.models, self._split(state, lambda mm: mm.state_size)):
            m.set_state(s)

    def get_state(self) -> Tensor:
        return self._merge([m.get_state() for m in self.models], self.state_size, self.batch_shape)

    def get_params(self) -> Dict[str, Tensor]:
        per_model = [m.get_params() for m in self.models]
        names = [k for p in per_model for k in p]
        dup = {k for k in names if names.count(k) > 1}
        if dup:
            raise ValueError(f'Duplicate parameter names in CompoundKinematicModel: {dup}')
        out = {}
        for i, params in enumerate(per_model):
            for k, v in params.items():
                full = torch.zeros((self.batch_size,) + v.shape[1:], dtype=v.dtype, device=v.device)
                full[self.batch_assignments == i] = v
                out[k] = full.reshape(self.batch_shape)
        return out

    def set_params(self, **kwargs) -> None:
        for i, m in enumerate(self.models):
            own = m.get_params()
            m.set_params(**{k: v.flatten()[self.batch_assignments == i] for k, v in kwargs.items() if k in own})

    def flattening(self, batch_shape) -> None:
        for m in self.models:
            m.flattening(batch_shape)

    def unflattening(self, batch_shape) -> None:
        for m in self.models:
            m.unflattening(batch_shape)

    def map_param(self, f) -> None:
        for m in self.models:
            m.map_param(f)

    def normalize_action(self, action: Tensor) -> Tensor:
        parts = [m.normalize_action(a) for m, a in zip(self.models, self._split(action, lambda mm: mm.action_size))]
        return self._merge(parts, self.action_size, action.shape[:-1])

    def denormalize_action(self, action: Tensor) -> Tensor:
        parts = [m.denormalize_action(a) for m, a in zip(self.models, self._split(action, lambda mm: mm.action_size))]
        return self._merge(parts, self.action_size, action.shape[:-1])
