"""Batched lander environment (device-resident) + the single-env facade the reference's loops use.

Stands where the reference calls `gym.make('LunarLanderContinuous-v2')`, `env.reset()`,
`env.step(a)` and `env.action_space.sample()` (example/dsac.py:78-79,99,102,127).  gym/Box2D are
not available; the dynamics are this build's own Box2D-style model (csrc/env.hip; specification
and caveats in oracle/env_oracle.py's header and DESIGN.md §env).
"""
import ctypes

import numpy as np
import torch

from . import _lib


class VecLunarLander:
    """n environments stepped by one kernel launch; all tensors stay on the device."""

    obs_dim, act_dim, act_high = 8, 2, 1.0

    def __init__(self, n_envs, seed=0, max_ep_len=1000, device=None):
        _lib.require_gpu()
        self._lib = _lib.load()
        self.n, self.seed, self.max_ep_len = int(n_envs), int(seed) & 0xFFFFFFFF, int(max_ep_len)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_env_create(ctypes.byref(h), self.device.index, self.n, self.seed, self.max_ep_len))
        self._h = h
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=self.device)
        self.obs = e(self.n, 8)        # observation to act on next
        self.obs2 = e(self.n, 8)       # o2 of the last transition
        self.rew, self.done = e(self.n), e(self.n)
        self.ended = e(self.n, dt=torch.uint8)
        self._sample_ctr = 0
        self.reset()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_env_destroy(h)

    def reset(self, mask=None):
        """env.reset() for every env (or those with mask != 0); returns obs[n, 8] (device)."""
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self._lib.ddrl_env_reset(self._h, _lib.dptr(m), _lib.dptr(self.obs), _lib.stream_ptr()))
        return self.obs

    def step(self, act):
        """env.step(a) for every env + episode bookkeeping of example/dsac.py:102-127.
        Returns (obs2, rew, done_as_stored, next_obs, ended) — device tensors owned by the env."""
        act = act.to(device=self.device, dtype=torch.float32).contiguous()
        _lib.check(self._lib.ddrl_env_step(self._h, _lib.dptr(act), _lib.dptr(self.obs2), _lib.dptr(self.rew),
                                           _lib.dptr(self.done), _lib.dptr(self.obs), _lib.dptr(self.ended),
                                           _lib.stream_ptr()))
        return self.obs2, self.rew, self.done, self.obs, self.ended

    def step_wrapped(self, act, act_noise=0.0, obs_noise=0.0, reward_scale=1.0, action_repeat=3, limit_steps=None):
        """env.step through `Wrapper` (algos/sac1/hyperparams.py:107-134) + the n-step rollout's episode
        bookkeeping (algos/sac1/sac_ray.py:212-258).  Same outputs as step(); done is the raw d."""
        act = act.to(device=self.device, dtype=torch.float32).contiguous()
        limit = int(limit_steps if limit_steps is not None else self.max_ep_len)
        _lib.check(self._lib.ddrl_env_step_wrapped(self._h, _lib.dptr(act), float(act_noise), float(obs_noise), float(reward_scale),
                                                   int(action_repeat), limit, _lib.dptr(self.obs2), _lib.dptr(self.rew),
                                                   _lib.dptr(self.done), _lib.dptr(self.obs), _lib.dptr(self.ended), _lib.stream_ptr()))
        return self.obs2, self.rew, self.done, self.obs, self.ended

    def sample_actions(self, out=None):
        """env.action_space.sample() for every env: U[-1, 1) from the counter generator."""
        out = out if out is not None else torch.empty(self.n, 2, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_uniform_fill(_lib.dptr(out), self.n * 2, -1.0, 1.0, self.seed ^ 0x5EED5EED,
                                               self._sample_ctr, _lib.stream_ptr()))
        self._sample_ctr += self.n * 2
        return out

    def stats(self):
        """(finished episodes, sum of returns, sum of lengths) since the last call."""
        ep, ln, rs = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(self._lib.ddrl_env_stats(self._h, ctypes.byref(ep), ctypes.byref(rs), ctypes.byref(ln),
                                            _lib.stream_ptr()))
        return int(ep.value), float(rs.value), int(ln.value)

    def get_state(self):
        s = torch.empty(_lib.DDRL_ENV_STATE_FIELDS, self.n, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_env_get_state(self._h, _lib.dptr(s), _lib.stream_ptr()))
        return s

    def set_state(self, s):
        s = s.to(device=self.device, dtype=torch.float32).contiguous()
        _lib.check(self._lib.ddrl_env_set_state(self._h, _lib.dptr(s), _lib.stream_ptr()))


class _ActionSpace:
    def __init__(self, env):
        self._env = env
        self.high = np.array([1.0, 1.0], dtype=np.float32)
        self.low = -self.high
        self.shape = (2,)

    def sample(self):
        return self._env._vec.sample_actions()[0].cpu().numpy()


class _ObsSpace:
    shape = (8,)


class LunarLander:
    """One environment with the gym call shape (reset() -> o; step(a) -> (o2, r, d, info)) for the
    reference-style loops (worker_rollout with num_envs == 1, worker_test).  `d` is True at the
    time limit too, as gym's TimeLimit wrapper reports it."""

    def __init__(self, seed=0, max_ep_len=1000):
        self._vec = VecLunarLander(1, seed=seed, max_ep_len=max_ep_len)
        self.action_space = _ActionSpace(self)
        self.observation_space = _ObsSpace()
        self._fresh = True  # the kernel resets finished envs itself

    def reset(self):
        if not self._fresh:
            self._vec.reset()
        self._fresh = False
        return self._vec.obs[0].cpu().numpy().astype(np.float64)

    def step(self, a):
        act = torch.as_tensor(np.asarray(a, np.float32).reshape(1, 2))
        o2, r, d, _, ended = self._vec.step(act)
        ended = bool(ended[0].item())
        self._fresh = ended
        return o2[0].cpu().numpy().astype(np.float64), float(r[0].item()), ended, {}


class Wrapper(object):
    """algos/sac1/hyperparams.py:107-134 on a host env: uniform action noise added IN PLACE to the caller's action (`action +=`),
    the action repeated `action_repeat` times with the rewards summed and scaled — except that a terminal inside the repeat returns
    reward 0.0, and action_repeat == 1 returns the first step's raw reward and un-noised observation (the reference's branches, kept
    as they are).  The reference hard-codes BipedalWalker's sizes (24 observations, 4 actions) for the noise; here they follow the
    arrays.  `rng`: np.random (the reference) or a seeded RandomState."""

    def __init__(self, env, obs_noise, act_noise, reward_scale, action_repeat=3, rng=None):
        self._env = env
        self.action_repeat = action_repeat
        self.act_noise = act_noise
        self.obs_noise = obs_noise
        self.reward_scale = reward_scale
        self._rng = np.random if rng is None else rng

    def __getattr__(self, name):
        return getattr(self._env, name)

    def _noisy(self, obs):
        obs = np.asarray(obs)
        return obs + self.obs_noise * (-2 * self._rng.random_sample(obs.shape[-1]) + 1)

    def reset(self):
        return self._noisy(self._env.reset())

    def step(self, action):
        action += self.act_noise * (-2 * self._rng.random_sample(np.asarray(action).shape[-1]) + 1)
        r = 0.0
        for _ in range(self.action_repeat):
            obs_, reward_, done_, info_ = self._env.step(action)
            r = r + reward_
            if done_ and self.action_repeat != 1:
                return self._noisy(obs_), 0.0, done_, info_
            if self.action_repeat == 1:
                return obs_, r, done_, info_
        return self._noisy(obs_), self.reward_scale * r, done_, info_


def make(env_name="LunarLanderContinuous-v2", **kw):
    """gym.make stand-in (example/dsac.py:78)."""
    if "LunarLander" not in env_name:
        raise ValueError("only the LunarLanderContinuous-v2 stand-in is built (SURVEY §8(a) A7): %r" % env_name)
    return LunarLander(**kw)
