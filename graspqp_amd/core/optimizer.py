"""MALA* optimiser with the reference's class surface (core/optimizer.py:152-347) on the HIP propose/accept kernels.

State handling follows the reference exactly (pinned by tests/golden/mala_*.npz): the proposal is built from the
current pose and its gradient, rejected rows get pose / contact indices / gradient restored.
"""

import torch

from .. import _C


class MalaStar:
    def __init__(self, hand_model, switch_possibility=0.5, starting_temperature=18, temperature_decay=0.95,
                 annealing_period=30, step_size=0.005, stepsize_period=50, mu=0.98, device="cuda", global_ema=False,
                 clip_grad=False, batch_size=-1, generator=None):
        self.hand_model = hand_model
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.switch_possibility = float(switch_possibility)
        self.starting_temperature = float(starting_temperature)
        self.temperature_decay = float(temperature_decay)
        self.annealing_period = int(annealing_period)
        self.step_size = float(step_size)
        self.step_size_period = int(stepsize_period)
        self.mu = float(mu)
        self.clip_grad = bool(clip_grad)
        B, D = hand_model.hand_pose.shape
        self.step = torch.zeros(B, dtype=torch.long, device=self.device)
        self.ema_grad_hand_pose = torch.zeros(B, D, dtype=torch.float, device=self.device)
        self._g2 = torch.empty(D, device=self.device)
        self.generator = generator
        self.old_hand_pose = None
        self.old_contact_point_indices = None
        self.old_grad_hand_pose = None
        self.last_draws = None
        # reference quirk after a re-initialisation (set_parameters(env_mask=...) makes hand_pose a leaf, hand_model.py:
        # 846-851): in the NEXT iteration autograd accumulates the new gradient in place into the tensor kept as
        # old_grad_hand_pose, so rejected rows get old + new gradient back (pinned by tests/golden/mala_ext_*.npz)
        self._leaf_pose_pending = False
        self._old_grad_accumulates = False

    def _draw(self, B, n):
        g = self.generator
        u = torch.rand(B, n, dtype=torch.float, device=self.device, generator=g)
        # full-size randint + select instead of the reference's data-dependent size (removes a host sync)
        new_idx = torch.randint(self.hand_model.n_contact_candidates, (B, n), device=self.device, generator=g)
        return u, new_idx

    def try_step(self, draws=None):
        hm = self.hand_model
        hp = hm.hand_pose.detach().contiguous()
        grad = hm.hand_pose.grad
        grad = torch.zeros_like(hp) if grad is None else grad.contiguous()
        idx = hm.contact_point_indices.contiguous()
        B, D = hp.shape
        n = idx.shape[1]
        u_switch, new_idx = self._draw(B, n) if draws is None else draws
        self.last_draws = (u_switch, new_idx)
        pose_out = torch.empty_like(hp)
        idx_out = torch.empty_like(idx)
        s = torch.empty(B, device=self.device)
        _C.call("gq_mala_propose", _C.f32(hp), _C.f32(grad), _C.i64(idx), _C.f32(u_switch.contiguous()),
                _C.i64(new_idx.contiguous()), B, D, n, self.step_size, self.step_size_period, self.temperature_decay,
                self.mu, self.switch_possibility, int(self.clip_grad), _C.f32(self.ema_grad_hand_pose), _C.i64(self.step),
                _C.f32(pose_out), _C.i64(idx_out), _C.f32(s), _C.f32(self._g2), None, 0, None, _C.stream_ptr())
        self.old_hand_pose = hp
        self.old_contact_point_indices = idx
        self.old_grad_hand_pose = grad
        self._old_grad_accumulates, self._leaf_pose_pending = self._leaf_pose_pending, False
        hm.set_parameters(pose_out.requires_grad_(), idx_out, _known_finite=True)
        return s

    def reset_envs(self, mask):
        self.step[mask] = 0
        self.ema_grad_hand_pose[mask] = 0
        self.old_hand_pose[mask] = self.hand_model.hand_pose.detach()[mask]
        self.old_contact_point_indices[mask] = self.hand_model.contact_point_indices[mask]
        self.old_grad_hand_pose[mask] = 0 * self.old_grad_hand_pose[mask]
        self._leaf_pose_pending = True

    def accept_step(self, energy, new_energy, reset_mask=None, z_score=None, z_score_threshold=2.0, u_accept=None):
        """Returns (accept (B,) bool, temperature (B,)).  ``energy`` is updated in place for accepted rows
        (the reference does that one line later, fit.py:454)."""
        hm = self.hand_model
        B, D = hm.hand_pose.shape
        n = hm.contact_point_indices.shape[1]
        if u_accept is None:
            u_accept = torch.rand(B, dtype=torch.float, device=self.device, generator=self.generator)
        pose_new = hm.hand_pose.detach().contiguous()
        grad_new = hm.hand_pose.grad
        grad_new = torch.zeros_like(pose_new) if grad_new is None else grad_new.contiguous()
        idx_new = hm.contact_point_indices.contiguous()
        # accepted state buffers start from the old state and receive accepted rows
        pose = self.old_hand_pose.clone()
        idx = self.old_contact_point_indices.clone()
        grad = self.old_grad_hand_pose.clone()
        if self._old_grad_accumulates:
            grad += grad_new
        accept = torch.empty(B, dtype=torch.uint8, device=self.device)
        T = torch.empty(B, device=self.device)
        rm = None if reset_mask is None else reset_mask.to(torch.uint8).contiguous()
        _C.call("gq_mala_accept", _C.f32(new_energy.detach().contiguous()), _C.f32(u_accept.contiguous()),
                _C.f32(None if z_score is None else z_score.detach().contiguous()), _C.u8(rm), _C.i64(self.step),
                _C.f32(pose_new), _C.i64(idx_new), _C.f32(grad_new), B, D, n, self.starting_temperature,
                self.temperature_decay, self.annealing_period, _C.f32(energy), _C.f32(pose), _C.i64(idx), _C.f32(grad),
                _C.u8(accept), _C.f32(T), 0, None, None, _C.stream_ptr())
        # accepted state becomes the model state (optimizer.py:325-338): pose / indices / gradient of rejected rows
        # are the old ones; kinematics are refreshed for all rows like the reference's trailing fk() call
        hp = pose.requires_grad_()
        hm.hand_pose = hp
        hm.global_translation = hp[:, 0:3]
        with torch.no_grad():
            hm._set_contact_idxs(idx)
        hp.grad = grad
        return accept.bool(), T

    def zero_grad(self):
        if self.hand_model.hand_pose.grad is not None:
            self.hand_model.hand_pose.grad.data.zero_()


class AnnealingDexGraspNet(MalaStar):
    """DexGraspNet's annealing optimiser with the reference's surface (core/optimizer.py:11-149): the same RMS-normalised
    proposal and Metropolis test as MalaStar on the same HIP kernels, but one global step counter, no gradient clipping,
    no z-score in the temperature, and re-initialisations are not tracked (``reset_envs`` is a no-op, ``accept_step``
    ignores ``reset_mask`` / ``z_score``)."""

    def __init__(self, hand_model, switch_possibility=0.5, starting_temperature=18, temperature_decay=0.95,
                 annealing_period=30, step_size=0.005, stepsize_period=50, mu=0.98, device="cuda", generator=None, **kwargs):
        super().__init__(hand_model, switch_possibility, starting_temperature, temperature_decay, annealing_period, step_size,
                         stepsize_period, mu, device, clip_grad=False, generator=generator)

    def reset_envs(self, mask):
        pass

    def accept_step(self, energy, new_energy, *args, u_accept=None, **kwargs):
        return super().accept_step(energy, new_energy, None, None, u_accept=u_accept)
