"""ORACLE — TEST INFRASTRUCTURE ONLY.

NumPy (float32, vectorised over envs) restatement of the build's batched lander environment
(csrc/env.hip).  It stands where the reference calls gym's LunarLanderContinuous-v2
(example/dsac.py:78-79,102,127; algos/sac1/sac1.py:162,185,213).

PARITY UNPINNED against the reference's environment: gym and Box2D are third-party, absent from
/root/reference and from this image, versions unpinned (env id implies gym ~0.10-0.17), and the
reference holds no fixtures for env transitions.  What is kept from the published gym
lunar_lander.py interface: observation layout and normalisation, action semantics (main engine
fires iff a0 > 0 with power (clip(a0,0,1)+1)/2; side engines iff |a1| > 0.5 with power
clip(|a1|,0.5,1)), engine impulses with dispersion noise, shaping reward
(-100 dist -100 speed -100 |angle| +10 per leg contact, minus 0.30/0.03 fuel costs), -100 on crash
or leaving the viewport, +100 when the lander comes to rest, FPS 50, SCALE 30, gravity -10,
terrain of 11 smoothed chunks with a flat helipad, random initial impulse.
What is this build's OWN model ("Box2D-style"): one rigid body (mass 4.82, inertia
0.84) with two rigid leg contact points, sequential-impulse contacts (4 iterations, speculative
margin, Baumgarte 0.2, friction 0.5) against the piecewise-linear terrain with a vertical contact
normal, hull-vertex crash test, rest detection after 25 slow steps on both legs.

The arithmetic is written so that the HIP kernel reproduces it BIT-EXACTLY: float32 only, one
rounding per operation (no FMA contraction), sqrt/divide correctly rounded, sin/cos from the shared
polynomial `sincos32` below instead of libm, integers via the counter hash of noise_oracle.
Episode bookkeeping follows example/dsac.py:102-127 (time limit is not a terminal for the stored
done flag, but ends the episode).
"""
import numpy as np

from .noise_oracle import hash3, mix32, u01

F = np.float32
NF = 32  # DDRL_ENV_STATE_FIELDS
(X, Y, VX, VY, ANG, OM, C1, C2, PREV, HASP, EPLEN, EPRET, SLEEP, EPI, T0) = range(15)
PSTEP = 25

FPS = F(50.0)
DT = F(0.02)
SCALE = F(30.0)
W = F(20.0)
H = F(13.333333)
HELIPAD_Y = F(3.3333333)
LEG_DOWN = F(0.6)
GRAV = F(-10.0)
INV_M = F(0.20746888)
INV_I = F(1.1904762)
MAIN_POWER = F(13.0)
SIDE_POWER = F(0.6)
SIDE_AWAY = F(0.4)      # SIDE_ENGINE_AWAY / SCALE
SIDE_H = F(0.46666667)  # SIDE_ENGINE_HEIGHT / SCALE
LEGX = F(0.6666667)
LEGY = F(-0.8666667)
HULL = [(-0.46666667, 0.56666666), (-0.56666666, 0.0), (-0.56666666, -0.33333334),
        (0.56666666, -0.33333334), (0.56666666, 0.0), (0.46666667, 0.56666666)]
NIT = 4
SLOP = F(0.005)
BAUM = F(0.2)
MU = F(0.5)
SLEEP_V2 = F(0.0025)
SLEEP_W = F(0.05)
SLEEP_STEPS = F(25.0)
RESET_NOISE_STREAM = 0xFFFFFFEF
RESET_STREAM = np.uint32(0xFFFFFFF0)


def sincos32(x):
    """Cody-Waite reduction by pi/2 + cephes sinf/cosf polynomials, plain float32 ops only."""
    x = np.asarray(x, dtype=F)
    k = np.rint(x * F(0.63661975))
    r = ((x - k * F(1.5703125)) - k * F(4.837513e-4)) - k * F(7.54979e-8)
    r2 = r * r
    s = r + (r * r2) * (F(-1.6666654611e-1) + r2 * (F(8.3321608736e-3) + r2 * F(-1.9515295891e-4)))
    c = (F(1.0) - F(0.5) * r2) + (r2 * r2) * (F(4.166664568298827e-2) + r2 * (F(-1.388731625493765e-3) + r2 * F(2.443315711809948e-5)))
    q = k.astype(np.int32) & 3
    sin = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cos = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sin.astype(F), cos.astype(F)


def _ground(S, px):
    """Piecewise-linear terrain height at world x (chunk width W/(CHUNKS-1) = 2)."""
    idx = np.clip(np.floor(px * F(0.5)), F(0.0), F(9.0)).astype(np.int64)
    cols = np.arange(S.shape[1])
    h0 = S[T0 + idx, cols]
    h1 = S[T0 + idx + 1, cols]
    t = (px - F(2.0) * idx.astype(F)) * F(0.5)
    return (h0 + (h1 - h0) * t).astype(F)


class LanderOracle:
    def __init__(self, n_envs, seed=0, max_ep_len=1000):
        self.n, self.seed, self.max_ep_len = int(n_envs), int(seed) & 0xFFFFFFFF, int(max_ep_len)
        self.S = np.zeros((NF, self.n), dtype=F)
        self.env_ids = np.arange(self.n, dtype=np.uint32)
        self.episodes, self.ret_sum, self.len_sum = 0, 0.0, 0
        self.reset()

    # -- RNG: one stream per (env, episode), indexed by (physics step, j) -----------------------
    def _rng(self, step, j):
        ep = self.S[EPI].astype(np.uint32)
        seed_e = np.uint32(self.seed) ^ mix32(ep)
        step = np.asarray(step, dtype=np.uint32) + np.zeros(self.n, np.uint32)
        b = step * np.uint32(16) + np.uint32(j)
        # per-env seed differs, so hash each env with its own seed word
        h = mix32(seed_e ^ np.uint32(0x9E3779B9))
        h = mix32(h + self.env_ids * np.uint32(0x85EBCA6B) + np.uint32(0x27D4EB2F))
        h = mix32(h ^ (b * np.uint32(0xC2B2AE35) + np.uint32(0x165667B1)))
        return u01(h)

    def obs(self):
        S = self.S
        o = np.empty((self.n, 8), dtype=F)
        o[:, 0] = (S[X] - F(10.0)) / F(10.0)
        o[:, 1] = (S[Y] - (HELIPAD_Y + LEG_DOWN)) / F(6.6666665)
        o[:, 2] = S[VX] * F(10.0) / FPS
        o[:, 3] = S[VY] * F(6.6666665) / FPS
        o[:, 4] = S[ANG]
        o[:, 5] = F(20.0) * S[OM] / FPS
        o[:, 6] = S[C1]
        o[:, 7] = S[C2]
        return o

    # -- one physics step for every env; returns (reward, done_env) -----------------------------
    def _physics(self, act):
        S = self.S
        a0 = np.clip(act[:, 0].astype(F), F(-1.0), F(1.0))
        a1 = np.clip(act[:, 1].astype(F), F(-1.0), F(1.0))
        step = S[PSTEP].astype(np.uint32)
        d0 = (self._rng(step, 0) * F(2.0) - F(1.0)) / SCALE
        d1 = (self._rng(step, 1) * F(2.0) - F(1.0)) / SCALE
        sn, cs = sincos32(S[ANG])
        tip0, tip1 = sn, cs
        side0, side1 = -tip1, tip0
        vx, vy, om = S[VX].copy(), S[VY].copy(), S[OM].copy()
        # main engine (impulse through the centre of mass: no torque)
        fire_m = a0 > F(0.0)
        m_power = np.where(fire_m, (np.clip(a0, F(0.0), F(1.0)) + F(1.0)) * F(0.5), F(0.0)).astype(F)
        ox = tip0 * (F(0.13333334) + F(2.0) * d0) + side0 * d1
        oy = -tip1 * (F(0.13333334) + F(2.0) * d0) - side1 * d1
        vx = vx + (-ox * MAIN_POWER * m_power) * INV_M
        vy = vy + (-oy * MAIN_POWER * m_power) * INV_M
        # side engines
        fire_s = np.abs(a1) > F(0.5)
        direction = np.where(a1 < F(0.0), F(-1.0), F(1.0)).astype(F)
        s_power = np.where(fire_s, np.clip(np.abs(a1), F(0.5), F(1.0)), F(0.0)).astype(F)
        arm = F(3.0) * d1 + direction * SIDE_AWAY
        ox = tip0 * d0 + side0 * arm
        oy = -tip1 * d0 - side1 * arm
        px_ = -ox * SIDE_POWER * s_power
        py_ = -oy * SIDE_POWER * s_power
        rx = ox - tip0 * F(0.56666666)
        ry = oy + tip1 * SIDE_H
        vx = vx + px_ * INV_M
        vy = vy + py_ * INV_M
        om = om + (rx * py_ - ry * px_) * INV_I
        # gravity
        vy = vy + GRAV * DT
        # leg contacts: sequential impulses, vertical normal, speculative margin
        x, y = S[X], S[Y]
        accn = [np.zeros(self.n, F), np.zeros(self.n, F)]
        acct = [np.zeros(self.n, F), np.zeros(self.n, F)]
        geo = []
        for leg in range(2):
            pbx = -LEGX if leg == 0 else LEGX
            rx = cs * pbx - sn * LEGY
            ry = sn * pbx + cs * LEGY
            gap = (y + ry) - _ground(S, x + rx)
            vmin = np.where(gap >= F(0.0), -gap / DT, BAUM * np.maximum(-gap - SLOP, F(0.0)) / DT).astype(F)
            kn = INV_M + (rx * rx) * INV_I
            kt = INV_M + (ry * ry) * INV_I
            geo.append((rx, ry, vmin, kn, kt))
        for _ in range(NIT):
            for leg in range(2):
                rx, ry, vmin, kn, kt = geo[leg]
                vn = vy + om * rx
                lam = -(vn - vmin) / kn
                new = np.maximum(accn[leg] + lam, F(0.0))
                dl = new - accn[leg]
                accn[leg] = new
                vy = vy + dl * INV_M
                om = om + (rx * dl) * INV_I
                vt = vx - om * ry
                lam = -vt / kt
                lim = MU * accn[leg]
                new = np.minimum(np.maximum(acct[leg] + lam, -lim), lim)
                dl = new - acct[leg]
                acct[leg] = new
                vx = vx + dl * INV_M
                om = om - (ry * dl) * INV_I
        # integrate
        x = x + vx * DT
        y = y + vy * DT
        ang = S[ANG] + om * DT
        S[X], S[Y], S[VX], S[VY], S[ANG], S[OM] = x, y, vx, vy, ang, om
        S[C1] = (accn[0] > F(0.0)).astype(F)
        S[C2] = (accn[1] > F(0.0)).astype(F)
        S[PSTEP] = S[PSTEP] + F(1.0)
        # crash test on the hull vertices at the new pose
        sn, cs = sincos32(ang)
        crash = np.zeros(self.n, dtype=bool)
        for hx, hy in HULL:
            hx, hy = F(hx), F(hy)
            wx = x + (cs * hx - sn * hy)
            wy = y + (sn * hx + cs * hy)
            crash |= wy < _ground(S, wx)
        # rest detection
        slow = ((vx * vx + vy * vy) < SLEEP_V2) & (np.abs(om) < SLEEP_W) & (S[C1] > F(0.0)) & (S[C2] > F(0.0))
        S[SLEEP] = np.where(slow, S[SLEEP] + F(1.0), F(0.0)).astype(F)
        asleep = S[SLEEP] >= SLEEP_STEPS
        # observation, shaping, reward (gym lunar_lander.py semantics)
        o = self.obs()
        shaping = ((F(-100.0) * np.sqrt(o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1])
                    - F(100.0) * np.sqrt(o[:, 2] * o[:, 2] + o[:, 3] * o[:, 3]))
                   - F(100.0) * np.abs(o[:, 4])) + F(10.0) * o[:, 6] + F(10.0) * o[:, 7]
        shaping = shaping.astype(F)
        rew = np.where(S[HASP] > F(0.0), shaping - S[PREV], F(0.0)).astype(F)
        S[PREV], S[HASP] = shaping, F(1.0)
        rew = (rew - m_power * F(0.30)) - s_power * F(0.03)
        out = crash | (np.abs(o[:, 0]) >= F(1.0))
        rew = np.where(out, F(-100.0), np.where(asleep, F(100.0), rew)).astype(F)
        return rew, (out | asleep), o

    # -- env.reset() for masked envs (gym: build terrain, random initial impulse, one no-op step)
    def reset(self, mask=None):
        S = self.S
        m = np.ones(self.n, bool) if mask is None else np.asarray(mask).astype(bool)
        keep = S.copy()
        hs = [self._rng(RESET_STREAM, j) * F(6.6666665) for j in range(12)]
        for j in range(3, 8):
            hs[j] = np.full(self.n, HELIPAD_Y, F)
        for i in range(11):
            S[T0 + i] = F(0.33) * ((hs[(i - 1) % 12] + hs[i]) + hs[i + 1])
        fx = (self._rng(RESET_STREAM, 12) * F(2.0) - F(1.0)) * F(1000.0)
        fy = (self._rng(RESET_STREAM, 13) * F(2.0) - F(1.0)) * F(1000.0)
        S[X], S[Y] = F(10.0), H
        S[VX], S[VY] = (fx * DT) * INV_M, (fy * DT) * INV_M
        for f in (ANG, OM, C1, C2, PREV, HASP, EPLEN, EPRET, SLEEP, PSTEP):
            S[f] = F(0.0)
        self._physics(np.zeros((self.n, 2), F))
        S[:, ~m] = keep[:, ~m]
        return self.obs()

    # -- env.step(a) + the worker's episode bookkeeping (example/dsac.py:102-127) ----------------
    def step(self, act):
        S = self.S
        rew, done_env, obs2 = self._physics(np.asarray(act, dtype=F).reshape(self.n, 2))
        S[EPLEN] = S[EPLEN] + F(1.0)
        S[EPRET] = S[EPRET] + rew
        limit = S[EPLEN] >= F(self.max_ep_len)
        done_store = np.where(limit, F(0.0), done_env.astype(F)).astype(F)  # dsac.py:109
        ended = done_env | limit                                           # dsac.py:118
        self.episodes += int(ended.sum())
        self.ret_sum += float(S[EPRET][ended].astype(np.float64).sum())
        self.len_sum += int(S[EPLEN][ended].sum())
        S[EPI] = np.where(ended, S[EPI] + F(1.0), S[EPI]).astype(F)
        next_obs = obs2.copy()
        if ended.any():
            next_obs = self.reset(ended)
        return obs2, rew, done_store, next_obs, ended.astype(np.uint8)

    # -- Wrapper.step (algos/sac1/hyperparams.py:123-134) + the n-step rollout's bookkeeping
    #    (algos/sac1/sac_ray.py:212-258); mirrors k_env_step_wrapped statement by statement ----------
    def step_wrapped(self, act, act_noise, obs_noise, reward_scale, repeat, limit_steps):
        S = self.S
        act = np.asarray(act, dtype=F).reshape(self.n, 2)
        st0 = S[PSTEP].astype(np.uint32)
        a = np.empty_like(act)
        a[:, 0] = act[:, 0] + F(act_noise) * (F(-2.0) * self._rng(st0, 2) + F(1.0))
        a[:, 1] = act[:, 1] + F(act_noise) * (F(-2.0) * self._rng(st0, 3) + F(1.0))
        act[...] = a   # hyperparams.py:124 `action += ...` mutates the caller's array (a float32 ndarray is updated in place)
        r = np.zeros(self.n, F)
        rew = np.zeros(self.n, F)
        obs = np.zeros((self.n, 8), F)
        done = np.zeros(self.n, bool)
        noisy = np.ones(self.n, bool)
        active = np.ones(self.n, bool)
        for _ in range(int(repeat)):
            if not active.any():
                break
            keep = S.copy()
            rk, dk, ok = self._physics(a)
            S[:, ~active] = keep[:, ~active]
            r = np.where(active, (r + rk).astype(F), r)
            obs[active] = ok[active]
            done = np.where(active, dk, done)
            brk_done = active & dk & (repeat != 1)
            rew = np.where(brk_done, F(0.0), rew)
            brk_one = active & ~brk_done & (repeat == 1)
            rew = np.where(brk_one, r, rew)
            noisy = noisy & ~brk_one
            cont = active & ~brk_done & ~brk_one
            rew = np.where(cont, (F(reward_scale) * r).astype(F), rew)
            active = cont
        st1 = S[PSTEP].astype(np.uint32)
        for j in range(8):
            nz = (obs[:, j] + F(obs_noise) * (F(-2.0) * self._rng(st1, 4 + j) + F(1.0))).astype(F)
            obs[:, j] = np.where(noisy, nz, obs[:, j])
        S[EPLEN] = S[EPLEN] + F(1.0)
        S[EPRET] = (S[EPRET] + rew).astype(F)
        ended = done | (S[EPLEN] >= F(limit_steps))
        self.episodes += int(ended.sum())
        self.ret_sum += float(S[EPRET][ended].astype(np.float64).sum())
        self.len_sum += int(S[EPLEN][ended].sum())
        S[EPI] = np.where(ended, S[EPI] + F(1.0), S[EPI]).astype(F)
        next_obs = obs.copy()
        if ended.any():
            ro = self.reset(ended)
            for j in range(8):
                nz = (ro[:, j] + F(obs_noise) * (F(-2.0) * self._rng(RESET_NOISE_STREAM, j) + F(1.0))).astype(F)
                next_obs[:, j] = np.where(ended, nz, next_obs[:, j])
        return obs, rew.astype(F), done.astype(F), next_obs, ended.astype(np.uint8)

    def stats(self):
        out = (self.episodes, self.ret_sum, self.len_sum)
        self.episodes, self.ret_sum, self.len_sum = 0, 0.0, 0
        return out
