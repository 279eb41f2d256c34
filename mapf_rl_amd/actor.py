"""Vectorised actor loop: the reference's `Actor.run` (reference worker.py:368-428) for E lock-step
environments on one GPU -- policy inference, epsilon-greedy, environment step, per-episode local buffers and
the flush into the device replay all stay on the device; nothing crosses a process boundary.

Per iteration (= worker.py:376-414 for every environment at once):
  model.step_batch -> (only agent 0 explores: quirk Q7) -> env.step -> record agent 0's (q, a, r), the next
  observation (bit-packed straight from the env kernel), agent 0's post-communication hidden state and the
  comm mask -> environments whose episode ended (done, or 256 steps) compute their initial priorities
  (LocalBuffer.finish, buffer.py:153-179), are appended to the replay ring (GlobalBuffer.add) and restart
  on a fresh scenario (Actor.reset, worker.py:422-428).

Quirk Q8 (worker.py:399): on a time-out the reference runs one more model.step on the STALE observation only
to obtain `comm_mask` for the last buffer row (its q value is never used by the priorities, buffer.py:173).
The stale positions equal those of the previous step, so the same row is obtained here by repeating the
previous comm mask -- no extra inference.

No host synchronisation in `step()`: which environments finished stays a device mask -- the replay append
(`GlobalBuffer.add_finished` = mapf_replay_add_many: priorities, ring slots, rows, sum tree), the scenario reset
(`mapf_reset_envs` takes the mask) and the local-buffer rewind are all launched unconditionally and do nothing for
environments that are still running.  Episode outcomes for the curriculum go through a small device log that is read
when statistics are asked for (`drain_outcomes`)."""
import contextlib
import ctypes

import numpy as np
import torch

from . import environment as _envmod
from ._lib import check, lib
from .environment import VecEnvironment, generate_scenarios
from .streams import role_stream as _role_stream


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

MAX_STEPS = 256      # config.max_steps
FORWARD_GAMMA = 0.99


def _wait_for_default_stream(device):
    """An actor stepped on its own stream (train.py --overlap-actors) copies the learner's weights: behind everything the learner's
    stream (the default one) has queued, i.e. behind its last optimizer step."""
    if torch.device(device).type == "cuda":
        cur, src = torch.cuda.current_stream(device), torch.cuda.default_stream(device)
        if cur != src:
            cur.wait_stream(src)


def epsilon_ladder(num_envs, num_actors=16, base=0.4, alpha=7.0):
    """train.py:25: eps_i = 0.4 ** (1 + 7 i / 15); environment e plays the role of actor e mod 16."""
    i = torch.arange(num_envs) % num_actors
    return base ** (1 + i.double() / (num_actors - 1) * alpha)


def pack_comm_device(comm, cw):
    """bool [E, N, N] -> int32 [E, N, cw] (bit j of word j/32)."""
    E, N, _ = comm.shape
    pad = torch.zeros((E, N, cw * 32), dtype=torch.int64, device=comm.device)
    pad[:, :, :N] = comm
    w = (pad.view(E, N, cw, 32) << torch.arange(32, device=comm.device)).sum(-1)
    return (w & 0xFFFFFFFF).to(torch.int64).where(w < 2 ** 31, w - 2 ** 32).to(torch.int32)


class VecActor:
    REUSE_LATENTS = True  # encode only the agents whose observation changed since the previous step (fused.LatentCache: exact)
    FUSED_TAIL = True     # exploration .. episode flush of an iteration as one library call (mapf_actor_iteration_tail)

    STAGE_AHEAD = True    # the next scenario of every environment is drawn beside the policy's forward (mapf_stage_next); an episode end swaps it in

    def __init__(self, env: VecEnvironment, model, buffer, epsilons=None, max_steps=MAX_STEPS, seed=0, density=-1.0,
                 keep_flushed=False, on_device_reset=True, weights_period=None, stage_ahead=None):
        """weights_period (reference config.actor_update_steps = 400, worker.py:416-420: an actor pulls the learner's weights every
        400 steps and acts on that snapshot in between): None = act on `model` itself (always the newest weights); an integer =
        keep an own copy of the network, refreshed from `model` every that many steps -- the reference's semantics, and what keeps
        the latents of unchanged observations reusable from step to step while the learner updates `model`."""
        self.env, self.buffer = env, buffer
        self.source_model, self.weights_period = model, weights_period
        if weights_period is not None:
            from copy import deepcopy

            model = deepcopy(model)
            for p in model.parameters():
                p.requires_grad_(False)
        self.model = model
        self._since_pull = 0
        from .fused import LatentCache

        self.latents = LatentCache() if (self.REUSE_LATENTS and env.device.type == "cuda") else None
        E, N = env.num_envs, env.num_agents
        d = env.device
        self.E, self.N, self.device, self.max_steps, self.density = E, N, d, max_steps, density
        assert buffer is None or buffer.max_agents >= N, "replay rows must hold at least the environment's agents"
        self.eps = (epsilon_ladder(E) if epsilons is None else torch.as_tensor(epsilons, dtype=torch.float64).expand(E)).to(d)
        self.actor_ids = (torch.arange(E) % 16).tolist()
        self.explore_seed, self._explore_counter = (seed * 0x9E3779B1 + 12345) & 0xFFFFFFFFFFFFFFFF, 0
        self.last_policy_actions = self._act8 = self._state = self._state_key = None
        self.scenario_seed = seed * 1000003 + 17
        # Scenario generation off the critical path (round 5; reference worker.py:422-428 draws at the episode's end): the scenario of
        # (seed, environment, reset count) does not depend on WHEN it is drawn, so with one fixed seed per actor the next one of every
        # environment is staged on a second stream beside the policy's forward pass, and the reset behind env.step is a copy.
        # (Rounds 1-4 passed a new seed every iteration, which tied a scenario to the iteration its episode ended in.)
        stage = self.STAGE_AHEAD if stage_ahead is None else stage_ahead
        self._fixed_seed = bool(stage and on_device_reset and env.device.type == "cuda")
        self._stage_stream = _role_stream(env.device, "actor_stage") if self._fixed_seed else None  # (None with _fixed_seed: the same scenarios, drawn at the reset -- tests)
        self._stage_ev = None
        self.RD = env.obs_bits_row_dwords
        # replay rows are laid out for A = buffer.max_agents >= N agents: the N-agent bit row is a prefix of the
        # A-agent row (bit a*486 + ...), so local rows are simply allocated at the replay's width, zero padded
        self.A = N if buffer is None else buffer.max_agents
        self.RDA = self.RD if buffer is None else buffer.row_dwords
        self.CW = (self.A + 31) // 32
        R = max_steps + 1
        self.lb_obs = torch.zeros((E, R, self.RDA), dtype=torch.int32, device=d)
        self.lb_comm = torch.zeros((E, R, self.A, self.CW), dtype=torch.int32, device=d)
        self.lb_act = torch.zeros((E, max_steps), dtype=torch.uint8, device=d)
        self.lb_rew = torch.zeros((E, max_steps), dtype=torch.float16, device=d)
        self.lb_hid = torch.zeros((E, max_steps, 256), dtype=torch.float16, device=d)
        self.lb_q = torch.zeros((E, max_steps, 5), dtype=torch.float32, device=d)
        self.t = torch.zeros(E, dtype=torch.int64, device=d)
        self.bits = torch.zeros((E, self.RD), dtype=torch.int32, device=d)
        self.ar = torch.arange(E, device=d)
        self.hidden = None
        self.env_steps = 0
        self.finished = torch.zeros(E, dtype=torch.uint8, device=d)
        self.counters = torch.zeros(2, dtype=torch.int64, device=d)  # {episodes finished, outcomes logged}
        self.stat_mask = torch.tensor([a >= 10 for a in self.actor_ids], dtype=torch.uint8, device=d)
        self.stat_log = torch.zeros(self.STAT_LOG, dtype=torch.uint8, device=d)
        self._stat_read = 0
        if buffer is not None:
            buffer.register_actor(self)
        self.keep_flushed = keep_flushed
        self.on_device_reset = on_device_reset
        self.flushed = []
        self._begin()

    def _begin(self):
        self.obs, self.pos = self.env.observe(obs_bits_out=self.bits)
        self.lb_obs[:, 0, :self.RD] = self.bits
        self.t.zero_()
        self.hidden = None

    # ------------------------------------------------------------------ one lock-step iteration
    @torch.no_grad()
    def step(self, actions_override=None):
        """One lock-step iteration.  `actions_override` (int tensor [E, N], tests only): the joint action to execute instead of
        the policy's (teacher forcing along a recorded trajectory); the policy's own greedy actions stay in `last_policy_actions`.
        Returns `self.finished` (u8 [E]: the environments whose episode ended in this step -- the actor's own buffer, valid until the
        next step)."""
        comm, comm_packed = self.policy_inputs()
        self.pull_weights()
        self.stage_scenarios()
        actions, q, hidden, comm = self.model.step_batch(self.obs, self.pos, self.hidden, comm, cache=self.latents)
        return self.act(actions, q, hidden, comm, comm_packed, actions_override)

    def stage_scenarios(self):
        """Next scenarios onto the staging stream: behind everything this stream has queued (the last iteration's reset moved the
        environments' epochs), beside what it queues next (the policy's forward); `act` waits for it in front of its reset."""
        if self._stage_stream is None:
            return
        cur = torch.cuda.current_stream(self.device)
        self._stage_stream.wait_stream(cur)
        with torch.cuda.stream(self._stage_stream):
            self.env.stage_next(self.density, self.scenario_seed)
            ev = torch.cuda.Event()
            ev.record(self._stage_stream)
        self._stage_ev = ev

    def _reset_seed(self):
        """The seed of this iteration's reset: fixed while scenarios are staged ahead, a new one per iteration otherwise."""
        if self._fixed_seed:
            if self._stage_ev is None and self._stage_stream is not None:
                # a reset that step() did not prepare (act() called directly, or a staging call that failed before its event): stage here,
                # in front of the reset -- once a handle has staged, a reset of the same (density, seed) without a staged scenario is
                # the sticky MAPF_ERR_NOT_READY (advisor, round 5: the contract was implicit in step())
                self.stage_scenarios()
            if self._stage_ev is not None:
                torch.cuda.current_stream(self.device).wait_event(self._stage_ev)
                self._stage_ev = None
            return self.scenario_seed
        self.scenario_seed += 1
        return self.scenario_seed

    def _tail_state(self, actions):
        """The mapf_actor_state struct of this actor (built once; rebuilt when a buffer it names was replaced)."""
        E, N = self.E, self.N
        if self._act8 is None or self.last_policy_actions is None or self.last_policy_actions.shape != actions.shape:
            self.last_policy_actions = torch.empty_like(actions)
            self._act8 = torch.empty((E, N), dtype=torch.int8, device=self.device)
            self._state = None
        key = (self.env.obs.data_ptr(), self.env.pos.data_ptr(), self.last_policy_actions.data_ptr(), self._act8.data_ptr())
        if self._state is None or self._state_key != key:
            from ._lib import ActorState

            e = self.env
            st = ActorState(E, N, self.max_steps, self.RD, self.RDA, self.A, self.STAT_LOG, 0)
            for k, t in (("lb_q", self.lb_q), ("lb_act", self.lb_act), ("lb_rew", self.lb_rew), ("lb_hid", self.lb_hid), ("lb_comm", self.lb_comm),
                         ("lb_obs", self.lb_obs), ("t", self.t), ("finished", self.finished), ("obs_bits", self.bits), ("stat_mask", self.stat_mask),
                         ("stat_log", self.stat_log), ("counters", self.counters), ("eps", self.eps), ("policy_actions", self.last_policy_actions),
                         ("act8", self._act8), ("obs", e.obs), ("pos", e.pos), ("reward_class", e.reward_class), ("reward", e.reward), ("done", e.done)):
                assert t.is_contiguous()
                setattr(st, k, t.data_ptr())
            self._state, self._state_key = st, key
        return self._state

    def pull_weights(self):
        """worker.py:416-420: with `weights_period` the actor acts on its own snapshot, refreshed every that many steps."""
        if self.weights_period is not None:
            if self._since_pull >= self.weights_period:
                _wait_for_default_stream(self.device)
                self.model.load_state_dict(self.source_model.state_dict())
                self._since_pull = 0
            self._since_pull += 1

    def policy_inputs(self):
        """(comm mask bool [E, N, N], the replay's packed comm rows) of the current positions (reference model.py:195-208)."""
        if self.pos.dtype == torch.int16 and self.N <= 128:  # mask + the replay's packed comm row from one kernel
            from .fused import comm_mask

            return comm_mask(self.pos, packed_words=self.CW)
        return None, None

    @torch.no_grad()
    def act(self, actions, q, hidden, comm, comm_packed, actions_override=None):
        """Everything of an iteration behind the policy's forward (worker.py:380-414): exploration, environment step, recording,
        episode flush.  actions int64 [E, N], q f32 [E, N, 5], hidden bf16 [E*N, 256] as Network.step_batch returns them."""
        E, N, d = self.E, self.N, self.device
        st = _stream(d)
        if actions_override is None and not self.keep_flushed and self.on_device_reset and comm_packed is not None and self.FUSED_TAIL:
            # the whole tail of the iteration as ONE call into the library (include/mapf_replay.h: mapf_actor_iteration_tail): the same
            # eight launches as below, without eight trips through the interpreter and ctypes -- the loop is host-bound at
            # curriculum shapes
            assert hidden.dtype == torch.bfloat16 and hidden.is_contiguous() and q.is_contiguous() and q.dtype == torch.float32
            actions = actions.contiguous()
            state = self._tail_state(actions)
            rseed = self._reset_seed()
            buf = self.buffer
            with (buf.lock if buf is not None else contextlib.nullcontext()):
                check(lib.mapf_actor_iteration_tail(ctypes.byref(state), self.env._h, None if buf is None else buf._h, _ptr(actions), _ptr(q), _ptr(hidden),
                                                    _ptr(comm_packed), self.explore_seed, self._explore_counter, float(self.density),
                                                    int(rseed) & 0xFFFFFFFFFFFFFFFF, st), "mapf_actor_iteration_tail")
            self._explore_counter += 1
            self.obs, self.pos = self.env.obs, self.env.pos
            self.hidden = hidden
            self.env_steps += E
            return self.finished
        if actions_override is None:
            # worker.py:380-382: only agent 0 of an environment explores -- exploration, the greedy copy and the int8 joint action in
            # one launch (csrc/mapf_actor.hip: actor_explore_kernel, counter-based generator)
            actions = actions.contiguous()
            if self._act8 is None or self.last_policy_actions is None or self.last_policy_actions.shape != actions.shape:
                self.last_policy_actions = torch.empty_like(actions)
                self._act8 = torch.empty((E, N), dtype=torch.int8, device=d)
            act8 = self._act8
            check(lib.mapf_actor_explore(E, N, _ptr(actions), _ptr(self.last_policy_actions), _ptr(act8), _ptr(self.eps), self.explore_seed,
                                         self._explore_counter, st), "mapf_actor_explore")
            self._explore_counter += 1
        else:  # tests: teacher forcing along a recorded trajectory
            self.last_policy_actions = actions.clone()
            actions = torch.as_tensor(actions_override).to(d, torch.int64).view(E, N)
            act8 = actions.to(torch.int8).contiguous()
        obs, pos, reward, done, _ = self.env.step(act8, obs_bits_out=self.bits)
        if comm_packed is None:
            comm_packed = pack_comm_device(comm, self.CW)
        assert hidden.dtype == torch.bfloat16 and hidden.is_contiguous() and q.is_contiguous() and q.dtype == torch.float32
        actions = actions.contiguous()
        # worker.py:388 -> buffer.py:140-151, the episode-end test (worker.py:390) and the last comm row of finished episodes
        # (Q8): one launch (csrc/mapf_actor.hip); self.t becomes the episode length so far, self.finished the end-of-episode mask
        st = _stream(d)
        check(lib.mapf_actor_record(E, N, self.max_steps, self.RD, self.RDA, self.A, _ptr(q), _ptr(actions), _ptr(reward), _ptr(hidden),
                                    _ptr(comm_packed), _ptr(self.bits), _ptr(done), _ptr(self.t), _ptr(self.lb_q), _ptr(self.lb_act),
                                    _ptr(self.lb_rew), _ptr(self.lb_hid), _ptr(self.lb_comm), _ptr(self.lb_obs), _ptr(self.finished), st),
              "mapf_actor_record")
        self.hidden = hidden
        self.env_steps += E
        self._flush(done)
        return self.finished

    # ------------------------------------------------------------------ episode end
    def _finish_priorities(self, ids, sizes):
        """LocalBuffer.finish (buffer.py:170-177) for the listed environments, f64 on the device, as torch ops: the independent
        statement of what mapf_replay_add_many computes (tests; `keep_flushed` records)."""
        S = self.max_steps
        q = self.lb_q[ids].double()                                   # [n, S, 5]
        rew = self.lb_rew[ids].double()                               # [n, S] (f16 values)
        steps = torch.arange(S, device=self.device)
        valid = steps[None, :] < sizes[:, None]
        rew = torch.where(valid, rew, torch.zeros_like(rew))
        nxt = torch.cat([rew[:, 1:], torch.zeros_like(rew[:, :1])], dim=1)
        ret = rew + FORWARD_GAMMA * nxt + q.max(-1).values            # np.convolve(ret, [0.99, 1], 'valid') + q_max
        qa = q.gather(-1, self.lb_act[ids].long().unsqueeze(-1)).squeeze(-1)
        td = (ret - qa).abs()
        td = torch.where(valid, td, torch.zeros_like(td))             # zeros past the episode end
        if S < 256:                                                   # a replay slot always has 256 leaves (worker.py:87,94)
            td = torch.cat([td, torch.zeros((td.shape[0], 256 - S), dtype=td.dtype, device=td.device)], dim=1)
        return td.contiguous()

    def _flush(self, done):
        """self.finished: u8 [E] device mask (episode over, set by mapf_actor_record); done: u8 [E] (all agents on their goals).
        Nothing here reads the device: every launch is unconditional and returns at once for environments that are still running."""
        finished = self.finished
        E, N, d = self.E, self.N, self.device
        sizes = self.t
        st = _stream(d)
        if self.keep_flushed:  # tests: host-side copies of every finished episode (synchronises)
            ids = finished.bool().nonzero().view(-1)
            td = self._finish_priorities(ids, sizes[ids])
            for k, e in enumerate(ids.tolist()):
                size = int(sizes[e])
                self.flushed.append(dict(env=e, size=size, done=bool(done[e]), obs=self.lb_obs[e, :size + 1].clone(),
                                         comm=self.lb_comm[e, :size + 1].clone(), act=self.lb_act[e, :size].clone(),
                                         rew=self.lb_rew[e, :size].clone(), hid=self.lb_hid[e, :size].clone(),
                                         q=self.lb_q[e, :size].clone(), td=td[k].clone()))
        if self.buffer is not None:
            self.buffer.add_finished(N, self.finished, sizes, done, self.lb_obs, self.lb_comm, self.lb_act, self.lb_rew, self.lb_hid, self.lb_q)
        # episode counter + curriculum outcomes (worker.py:74-82: actors with id >= 10) in episode order into the device log
        check(lib.mapf_actor_log(E, _ptr(self.finished), _ptr(done), _ptr(self.stat_mask), _ptr(self.stat_log), self.STAT_LOG,
                                 _ptr(self.counters), st), "mapf_actor_log")
        # Actor.reset (worker.py:422-428): fresh scenario, recurrent state cleared
        rseed = self._reset_seed()
        if self.on_device_reset:   # mapf_reset_envs: generation + placement + navi (or the staged scenario's hand-over), only where the mask is set
            self.env.reset_envs(self.finished, self.density, rseed)
        else:                      # host generator + partial load (synchronises)
            ids_h = finished.bool().nonzero().view(-1).tolist()
            if ids_h:
                maps, agents, goals, _ = generate_scenarios(len(ids_h), self.env.map_length, N, self.density, rseed)
                self.env.load_envs(ids_h, maps, agents, goals)
        # obs / pos / bits already hold what env.step wrote for the environments that go on: only the reset ones are re-observed
        self.obs, self.pos = self.env.observe(obs_bits_out=self.bits, mask=self.finished if self.on_device_reset else None)
        check(lib.mapf_actor_rewind(E, N, self.max_steps, self.RD, self.RDA, _ptr(self.finished), _ptr(self.bits), _ptr(self.t),
                                    _ptr(self.lb_obs), _ptr(self.hidden), st), "mapf_actor_rewind")

    STAT_LOG = 4096  # outcomes kept between two reads (a level's window is the last 200)

    def drain_outcomes(self):
        """Feeds the episode outcomes logged since the last call to the curriculum's level table, in episode order
        (worker.py:74-82).  Reads the device log: synchronises; called when statistics are asked for."""
        if self.buffer is None:
            return 0
        n = int(self.counters[1])
        k = min(n - self._stat_read, self.STAT_LOG)
        if k > 0:
            log = self.stat_log.cpu().numpy()
            key = (self.N, self.env.map_length)
            for i in range(n - k, n):
                self.buffer.levels.record(key, bool(log[i % self.STAT_LOG]))
        self._stat_read = n
        return k

    @property
    def episodes(self):
        return int(self.counters[0])

    def run(self, num_iterations):
        for _ in range(num_iterations):
            self.step()
        return self.env_steps
