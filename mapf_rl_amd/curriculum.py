"""Curriculum of (num_agents, map_length) levels: the reference's adaptive training schedule
(reference worker.py:74-82 statistics, :205-226 level promotion, :237-250 stop criterion; config.py:49-52;
Environment.reset(level) environment.py:148-151).

`LevelTable` owns the schedule's state: per active level a sliding window of the last 200 episode outcomes, the
promotion rule (a full window with >= pass_rate successes opens the level with one more agent and the level with a
5 cells longer map side, and retires the level unless its map is already the longest) and the stop criterion (every
agent count has passed on the longest map).  `GlobalBuffer` holds one and exposes it under the reference's names
(`stat_dict`, `get_level`, `check_done`, and the level lines of `stats`).  With several ranks the windows are pooled
(`pooled_counts`: one small all-reduce) so that every rank takes the same promotion and stop decisions.

The reference's 16 actors each draw a random level per episode.  Here every active level owns a
`VecEnvironment` + `VecActor` of `envs_per_level` lock-step environments (`CurriculumActors`); all of them record
into ONE device replay whose rows are laid out for `config.max_num_agetns` agents (smaller levels are a zero-padded
prefix of the row, see actor.py).  `sync_levels()` follows the level table after every `stats()` call: levels that
appeared get actors, levels that were promoted away are retired."""
import os

import torch

from .actor import VecActor
from .environment import VecEnvironment, generate_scenarios


class LevelTable:
    WINDOW = 200   # episodes per level window (worker.py:77,211)
    MAP_STEP = 5   # worker.py:217

    def __init__(self, init_set=(1, 10), max_agents=6, max_map_length=40, pass_rate=0.9, fixed=False):
        self.max_agents, self.max_map_length, self.pass_rate = max_agents, max_map_length, pass_rate
        self.fixed = fixed  # fixed-level training (train.py --agents/--map): outcomes are recorded, levels never change
        self.windows = {tuple(init_set): []}

    @property
    def levels(self):
        return list(self.windows.keys())

    def record(self, key, success):
        """One finished episode of level `key` (worker.py:74-82); episodes of retired levels are ignored."""
        w = self.windows.get(tuple(key))
        if w is not None:
            if len(w) >= self.WINDOW:
                del w[0]
            w.append(bool(success))

    def counts(self):
        return {k: (sum(w), len(w)) for k, w in self.windows.items()}

    def _passed(self, cnt, full):
        return cnt[1] >= full and cnt[0] >= full * self.pass_rate

    def advance(self, counts=None, full=None):
        """Applies the promotion rule (worker.py:211-224) to `counts` {level: (successes, episodes)} -- this table's own
        windows by default, the pooled windows of all ranks in a multi-GPU run (full = WINDOW * world).  Returns the lines the
        reference prints per level."""
        counts = self.counts() if counts is None else counts
        full = self.WINDOW if full is None else full
        lines = []
        for key in self.levels:
            ok, n = counts.get(key, (0, 0))
            lines.append("{}: {}/{}".format(key, ok, n))
            if self.fixed or not self._passed((ok, n), full):
                continue
            more_agents, longer_map = (key[0] + 1, key[1]), (key[0], key[1] + self.MAP_STEP)
            if more_agents[0] <= self.max_agents:
                self.windows.setdefault(more_agents, [])
            if key[1] < self.max_map_length:  # the longest map is never retired: the stop criterion reads its window
                self.windows.setdefault(longer_map, [])
                del self.windows[key]
        return lines

    def done(self, counts=None, full=None):
        """worker.py:237-250: every agent count 1..max_agents has a full, passing window on the longest map."""
        counts = self.counts() if counts is None else counts
        full = self.WINDOW if full is None else full
        return all(self._passed(counts.get((n, self.max_map_length), (0, 0)), full) for n in range(1, self.max_agents + 1))

    # ---- several ranks: pooled windows -> identical decisions everywhere
    def _grid(self):
        """Every level the schedule can reach from the current table, in a fixed order (identical on all ranks)."""
        a0, m0 = min(k[0] for k in self.windows), min(k[1] for k in self.windows)
        reach = {(a, m) for a in range(a0, max(self.max_agents, a0) + 1)
                 for m in range(m0, max(self.max_map_length, m0) + 1, self.MAP_STEP)}
        return sorted(reach | set(self.windows))

    def pooled_counts(self, device, group=None):
        """Sum of every rank's (successes, episodes) per level: one all-reduce of a small integer tensor.  Every rank holds the
        same level set (same initial table, same pooled decisions), so the enumeration agrees."""
        import torch.distributed as dist

        grid, own = self._grid(), self.counts()
        t = torch.tensor([own.get(k, (0, 0)) for k in grid], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        vals = t.tolist()
        return {k: tuple(v) for k, v in zip(grid, vals)}


class CurriculumActors:
    """One vectorised actor per active level, stepped TOGETHER: the levels' observations live back to back in one buffer, so an
    iteration is one change detection + one encoder launch + one input-projection GEMM + one Q head for all levels
    (`Network.step_levels`), and per level only what depends on its shape: comm mask, recurrence, environment step, recording,
    episode flush.  (Round 2 stepped the levels one after the other: ~31 launches x up to 13 levels on small batches.)
    `weights_period` (reference config.actor_update_steps = 400, worker.py:416-420): the actors act on a snapshot of `model`
    refreshed every that many iterations; None = on `model` itself."""

    BATCHED = True
    # (the environment variables: A/B and diagnostic runs of train.py)
    MERGED = os.environ.get("MAPF_ACTOR_MERGED", "1") != "0"   # every per-level launch of an iteration merged over all levels (`_iteration`)
    GRAPH = os.environ.get("MAPF_ACTOR_GRAPH", "1") != "0"     # ... and the whole iteration replayed from a captured HIP graph
    PACK_RECURRENCE = True  # 16 // N environments of a level per workgroup of the policy recurrence (the weight stream is shared)

    def __init__(self, model, buffer, envs_per_level=256, device=None, seed=0, max_steps=256, reward_fn=None, weights_period=None):
        self.source_model, self.buffer, self.weights_period = model, buffer, weights_period
        if weights_period is not None:
            from copy import deepcopy

            model = deepcopy(model)
            for p in model.parameters():
                p.requires_grad_(False)
        self.model = model
        self._since_pull = 0
        self.envs_per_level, self.seed, self.max_steps, self.reward_fn = envs_per_level, seed, max_steps, reward_fn
        self.device = buffer.device if device is None else torch.device(device)
        self.actors = {}
        self.retired_env_steps = 0
        self.obs_all = self.latents = self.multi = self._graph = self._cap_stream = None
        self._warm = self.graph_replays = 0
        self.tick = torch.zeros(1, dtype=torch.int64, device=self.device)  # iteration counter on the device (exploration / scenario streams)
        self.policy_override = None
        self.sync_levels()

    def set_policy_override(self, fn):
        """bench.py / tests (the counterpart of VecActor.step(actions_override=...)): `fn(obs_all u8 [rows, 6, 9, 9]) -> int64 [rows]` gives
        the action every agent row executes instead of the network's greedy one (the network's forward runs all the same; agent 0 of an
        environment still explores on top, worker.py:380-382).  Capturable torch operations only -- it becomes part of the replayed
        iteration; None = the policy's own actions again."""
        self.policy_override = fn
        self._graph = None
        self._warm = 0

    def _make(self, key):
        n, L = key
        E = self.envs_per_level
        env = VecEnvironment(E, L, n, reward_fn=self.reward_fn, device=self.device)
        maps, agents, goals, _ = generate_scenarios(E, L, n, -1.0, seed=self.seed * 7919 + n * 131 + L)
        env.load(maps, agents, goals)
        # (stage_ahead off: the levels' resets are one merged launch inside the captured iteration, environment.MultiEnvironment)
        return VecActor(env, self.model, self.buffer, max_steps=self.max_steps, seed=self.seed + n * 1000 + L, density=-1.0, stage_ahead=False)

    def sync_levels(self):
        """Create actors for new levels, drop actors of levels no longer in GlobalBuffer.level (worker.py:224)."""
        want = [tuple(k) for k in self.buffer.get_level()]
        changed = False
        for key in want:
            if key not in self.actors:
                self.actors[key] = self._make(key)
                changed = True
        for key in list(self.actors):
            if key not in want:
                self.retired_env_steps += self.actors[key].env_steps
                del self.actors[key]
                changed = True
        if changed:
            self._layout()
        return want

    def _layout(self):
        """Everything the levels' iteration reads or writes back to back in ONE tensor per kind -- observations, positions, recurrent
        states, Q-values, actions, rewards, comm masks / packed rows, bit-packed observation rows, the per-environment episode
        state and the local buffers (all at the replay's row width) -- with every level's actor / environment handle holding
        views; one latent cache over the observations; a per-environment table {agents, first agent row, offset of the comm mask,
        offset of the bit row}; and the set that steps / resets / re-observes all levels by one launch each
        (environment.MultiEnvironment).  An iteration is then ~30 launches whatever the number of levels (`_iteration`)."""
        from .fused import LatentCache

        acts = list(self.actors.values())
        d = self.device
        rows, Et = sum(a.E * a.N for a in acts), sum(a.E for a in acts)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=d)
        self.obs_all = torch.empty((max(rows, 1), 6, 9, 9), dtype=torch.uint8, device=d)
        self.hidden_all, self.hidden_new = z((max(rows, 1), 256), torch.bfloat16), z((max(rows, 1), 256), torch.bfloat16)
        self.latents = LatentCache() if (VecActor.REUSE_LATENTS and d.type == "cuda") else None
        self.multi = self.envtab = None
        self._graph = None
        self._warm = 0
        if not acts:
            return
        a0 = acts[0]
        S, A, CW, RDA = a0.max_steps, a0.A, a0.CW, a0.RDA
        assert all((a.max_steps, a.A, a.CW, a.RDA) == (S, A, CW, RDA) for a in acts)
        per_row = dict(pos=((2,), torch.int16), reward=((), torch.float32), reward_class=((), torch.int8), act8=((), torch.int8),
                       policy=((), torch.int64), comm_packed=((CW,), torch.int32))
        per_env = dict(done=((), torch.uint8), finished=((), torch.uint8), t=((), torch.int64), eps=((), torch.float64), stat_mask=((), torch.uint8),
                       lb_obs=((S + 1, RDA), torch.int32), lb_comm=((S + 1, A, CW), torch.int32), lb_act=((S,), torch.uint8), lb_rew=((S,), torch.float16),
                       lb_hid=((S, 256), torch.float16), lb_q=((S, 5), torch.float32))
        self.cat = {k: z((rows,) + sh, dt) for k, (sh, dt) in per_row.items()}
        self.cat.update({k: z((Et,) + sh, dt) for k, (sh, dt) in per_env.items()})
        self.cat["q"], self.cat["act"] = z((rows, 5), torch.float32), z((rows,), torch.int64)
        self.cat["bits"] = z((sum(a.E * a.RD for a in acts),), torch.int32)
        self.cat["comm"] = z((sum(a.E * a.N * a.N for a in acts),), torch.uint8)
        self.cat["nag"] = torch.cat([torch.full((a.E,), a.N, dtype=torch.int32) for a in acts]).to(d)
        tab, aux, self._views = [], [], []
        r0 = e0 = c0 = b0 = 0
        for a in acts:
            E, N = a.E, a.N
            rs, es = slice(r0, r0 + E * N), slice(e0, e0 + E)

            def home(obj, name, view):  # the tensor moves into its slice of the shared buffer
                old = getattr(obj, name)
                if old is not None:
                    view.copy_(old.reshape(view.shape))
                setattr(obj, name, view)

            home(a.env, "obs", self.obs_all[rs].view(E, N, 6, 9, 9))
            home(a.env, "pos", self.cat["pos"][rs].view(E, N, 2))
            home(a.env, "reward", self.cat["reward"][rs].view(E, N))
            home(a.env, "reward_class", self.cat["reward_class"][rs].view(E, N))
            home(a.env, "done", self.cat["done"][es])
            a.obs, a.pos = a.env.obs, a.env.pos
            home(a, "hidden", self.hidden_all[rs])   # (None = episode start everywhere = the zero state, model.py:186-189)
            home(a, "_act8", self.cat["act8"][rs].view(E, N))
            home(a, "last_policy_actions", self.cat["policy"][rs].view(E, N))
            home(a, "bits", self.cat["bits"][b0:b0 + E * a.RD].view(E, a.RD))
            for k in ("finished", "t", "eps", "stat_mask", "lb_obs", "lb_comm", "lb_act", "lb_rew", "lb_hid", "lb_q"):
                home(a, k, self.cat[k][es])
            a.latents = None               # the shared cache takes over
            a._state = None                # (the fused per-level tail's struct names the old buffers)
            e = torch.arange(E, dtype=torch.int64)
            tab.append(torch.stack([torch.full((E,), N, dtype=torch.int64), r0 + e * N, c0 + e * N * N, b0 + e * a.RD], dim=1))
            aux.append(torch.stack([torch.full((E,), a.explore_seed - (1 << 64) if a.explore_seed >= (1 << 63) else a.explore_seed, dtype=torch.int64),
                                    torch.full((E,), a._explore_counter, dtype=torch.int64), e], dim=1))
            self._views.append((self.cat["comm"][c0:c0 + E * N * N], self.cat["comm_packed"][rs].view(E, N, CW), rs))
            r0, e0, c0, b0 = r0 + E * N, e0 + E, c0 + E * N * N, b0 + E * a.RD
        self.level_start = [0]
        for a in acts:
            self.level_start.append(self.level_start[-1] + a.E)
        if self.MERGED and self.BATCHED and d.type == "cuda" and all(a.on_device_reset and not a.keep_flushed for a in acts):
            from ._lib import ERR_UNSUPPORTED, MapfError
            from .environment import MultiEnvironment

            try:
                # (the streams a level's own actor would use: scenario_seed + 1, + 2, ... per iteration; exploration: counter 0, 1, ...)
                self.tick.zero_()
                for a in acts:
                    a._explore_base = a._explore_counter
                self.multi = MultiEnvironment([a.env for a in acts], [a._act8 for a in acts], [a.bits for a in acts], [a.finished for a in acts],
                                              reset_seeds=[a.scenario_seed + 1 for a in acts])
                if all(a.N <= 16 for a in acts):
                    self.envtab = torch.cat(tab).to(torch.int32).contiguous().to(d)
                    self.aux = torch.cat(aux).contiguous().to(d)
                    # the recurrence's own table: 16 // N consecutive environments of a level per workgroup (one agent tile of rows;
                    # a step streams the weights once per workgroup, include/mapf_dqn.h: mapf_recurrent_infer_multi)
                    rt, r0, c0 = [], 0, 0
                    for a in acts:
                        k = max(1, 16 // a.N) if self.PACK_RECURRENCE else 1
                        for e in range(0, a.E, k):
                            m = min(k, a.E - e)
                            rt.append((m * a.N, r0 + e * a.N, c0 + e * a.N * a.N, a.N))
                        r0, c0 = r0 + a.E * a.N, c0 + a.E * a.N * a.N
                    self.rtab = torch.tensor(rt, dtype=torch.int32).contiguous().to(d)
            except MapfError as ex:
                if ex.status != ERR_UNSUPPORTED:
                    raise

    # ---- one iteration of all levels (worker.py:376-414 for every level at once) ----
    def _iteration(self):
        if self.envtab is None:
            return self._iteration_per_level()
        import ctypes

        from ._lib import check, lib
        from .actor import _ptr, _stream

        acts = list(self.actors.values())
        c, a0, st = self.cat, acts[0], _stream(self.device)
        Et, n = self.level_start[-1], len(acts)
        # communication masks of all levels (reference model.py:195-208) + the replay's packed rows: one launch
        check(lib.mapf_comm_mask_multi(_ptr(c["pos"]), Et, _ptr(self.envtab), 4, 3, _ptr(c["comm"]), _ptr(c["comm_packed"]), a0.CW, st), "mapf_comm_mask_multi")
        # policy: change detection, encoder on the changed rows, projection GEMM, ONE recurrence launch, Q head, arg-max
        self.model.step_levels([(a.E, a.N, a.pos, a.hidden, None) for a in acts], self.obs_all, self.latents, hidden_out=self.hidden_new,
                               packed_inplace=True, merged=(self.rtab, c["comm"], self.hidden_all), q_out=c["q"], act_out=c["act"])
        if self.policy_override is not None:
            c["act"].copy_(self.policy_override(self.obs_all))
        # worker.py:380-382: agent 0 of every environment explores (its level's own stream: seed, base counter + tick, index in the level)
        check(lib.mapf_actor_explore_multi(Et, _ptr(self.envtab), _ptr(self.aux), _ptr(c["act"]), _ptr(c["policy"]), _ptr(c["act8"]), _ptr(c["eps"]),
                                           _ptr(self.tick), st), "mapf_actor_explore_multi")
        self.multi.step()
        check(lib.mapf_actor_record_multi(Et, a0.max_steps, a0.RDA, a0.A, _ptr(self.envtab), _ptr(c["q"]), _ptr(c["act"]), _ptr(c["reward"]),
                                          _ptr(self.hidden_new), _ptr(c["comm_packed"]), _ptr(c["bits"]), _ptr(c["done"]), _ptr(c["t"]), _ptr(c["lb_q"]),
                                          _ptr(c["lb_act"]), _ptr(c["lb_rew"]), _ptr(c["lb_hid"]), _ptr(c["lb_comm"]), _ptr(c["lb_obs"]), _ptr(c["finished"]), st),
              "mapf_actor_record_multi")
        if self.buffer is not None:  # every finished episode of every level into the replay, in level / environment order
            self.buffer.add_finished_env(c["nag"], c["finished"], c["t"], c["done"], c["lb_obs"], c["lb_comm"], c["lb_act"], c["lb_rew"], c["lb_hid"], c["lb_q"])
        check(lib.mapf_actor_log_multi(n, (ctypes.c_int32 * (n + 1))(*self.level_start), (ctypes.c_void_p * n)(*[a.stat_log.data_ptr() for a in acts]),
                                       (ctypes.c_void_p * n)(*[a.counters.data_ptr() for a in acts]), a0.STAT_LOG, _ptr(c["finished"]), _ptr(c["done"]),
                                       _ptr(c["stat_mask"]), st), "mapf_actor_log_multi")
        self.multi.reset(a0.density, self.tick)   # Actor.reset (worker.py:422-428) of every finished episode
        self.multi.observe_masked()
        check(lib.mapf_actor_rewind_multi(Et, a0.max_steps, a0.RDA, _ptr(self.envtab), _ptr(c["finished"]), _ptr(c["bits"]), _ptr(c["t"]), _ptr(c["lb_obs"]),
                                          _ptr(self.hidden_new), _ptr(self.hidden_all), _ptr(self.tick), st), "mapf_actor_rewind_multi")
        # (the same launch hands the new hidden states on as the next iteration's input and counts the iteration)

    def _iteration_per_level(self):
        """Merged environment launches, everything else per level (a level of more than 16 agents is among them)."""
        from ._lib import check, lib
        from .actor import _ptr, _stream
        from .fused import comm_mask

        assert self.policy_override is None, "policy_override: merged launches only (levels of <= 16 agents)"
        acts = list(self.actors.values())
        inputs = [comm_mask(a.pos, packed_words=a.CW, out_mask=cm, out_packed=pk) for a, (cm, pk, _) in zip(acts, self._views)]
        outs = self.model.step_levels([(a.E, a.N, a.pos, a.hidden, cm) for a, (cm, _) in zip(acts, inputs)], self.obs_all, self.latents,
                                      hidden_out=self.hidden_new, packed_inplace=True)
        st = _stream(self.device)
        for a, (actions, q, hidden, _) in zip(acts, outs):
            # worker.py:380-382: only agent 0 explores; the draw is a function of (seed, iteration counter on the device, environment)
            check(lib.mapf_actor_explore_dev(a.E, a.N, _ptr(actions), _ptr(a.last_policy_actions), _ptr(a._act8), _ptr(a.eps), a.explore_seed,
                                             a._explore_base, _ptr(self.tick), st), "mapf_actor_explore_dev")
        self.multi.step()
        for a, (cm, packed), (actions, q, hidden, _) in zip(acts, inputs, outs):
            e = a.env
            check(lib.mapf_actor_record(a.E, a.N, a.max_steps, a.RD, a.RDA, a.A, _ptr(q), _ptr(actions), _ptr(e.reward), _ptr(hidden), _ptr(packed),
                                        _ptr(a.bits), _ptr(e.done), _ptr(a.t), _ptr(a.lb_q), _ptr(a.lb_act), _ptr(a.lb_rew), _ptr(a.lb_hid), _ptr(a.lb_comm),
                                        _ptr(a.lb_obs), _ptr(a.finished), st), "mapf_actor_record")
            if a.buffer is not None:
                a.buffer.add_finished(a.N, a.finished, a.t, e.done, a.lb_obs, a.lb_comm, a.lb_act, a.lb_rew, a.lb_hid, a.lb_q)
            check(lib.mapf_actor_log(a.E, _ptr(a.finished), _ptr(e.done), _ptr(a.stat_mask), _ptr(a.stat_log), a.STAT_LOG, _ptr(a.counters), st),
                  "mapf_actor_log")
        self.multi.reset(acts[0].density, self.tick)   # Actor.reset (worker.py:422-428) of every finished episode
        self.multi.observe_masked()
        for a, (actions, q, hidden, _) in zip(acts, outs):
            check(lib.mapf_actor_rewind(a.E, a.N, a.max_steps, a.RD, a.RDA, _ptr(a.finished), _ptr(a.bits), _ptr(a.t), _ptr(a.lb_obs), _ptr(hidden), st),
                  "mapf_actor_rewind")
        self.hidden_all.copy_(self.hidden_new)
        self.tick.add_(1)

    def _capture(self):
        """The launches of one iteration as a HIP graph (replayed until the level set changes).  No gc / cache flush around the
        capture (torch.cuda.graph does both); this stream is idle when it starts."""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        cur.synchronize()
        from .fused import capture_mode, no_gc_during_capture

        if self._cap_stream is None:
            # (no library warm-up: the captured iteration contains no library GEMM since round 5 -- every product in it is a kernel of
            # this library, tests/test_curriculum_gpu.py replays it from a fresh process)
            from .streams import role_stream

            self._cap_stream = role_stream(dev, "capture_actors")
        g = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.stream(self._cap_stream):
            g.capture_begin(capture_error_mode=capture_mode())
            try:
                self._iteration()
            finally:
                g.capture_end()
        return g

    _graph_wkey = None

    def _weights_key(self):
        """Identity of the weights the acting network holds now: its epoch (bumped by the fused optimizer step) and every
        parameter's (address, version) -- what the packed-weight caches of fused.py key on."""
        m = self.model
        if self._wparams is None or self._wparams[0] is not m:
            self._wparams = (m, list(m.parameters()))
        return (getattr(m, "weights_epoch", 0),) + tuple((p.data_ptr(), p._version) for p in self._wparams[1])

    _wparams = None

    def step(self):
        acts = list(self.actors.values())
        pulled = False
        if self.weights_period is not None:  # worker.py:416-420
            if self._since_pull >= self.weights_period:
                from .actor import _wait_for_default_stream

                _wait_for_default_stream(self.device)
                self.model.load_state_dict(self.source_model.state_dict())
                self._since_pull = 0
                pulled = True
            self._since_pull += 1
        if not self.BATCHED:
            for a in acts:
                a.step()
            return
        if self.multi is not None:
            # merged environment launches; and, once the allocator and the libraries have seen the iteration twice, its launches
            # replayed from a graph.  After a weight pull one iteration is issued directly: it re-packs the weight images (in place)
            # and re-encodes every observation (fused.LatentCache), which the captured sequence does not contain.
            # The same after ANY other change of the acting network's weights (weights_period None: the actors act on the learner's
            # module itself, which its optimizer step rewrites): the captured launches read the packed images and cached latents of
            # the last directly issued iteration, so a replay would mix them with the live head parameters.
            wkey = self._weights_key()
            if self.GRAPH and not pulled and self._warm >= 2 and wkey == self._graph_wkey:
                if self._graph is None:
                    self._graph = self._capture()
                self._graph.replay()
                self.graph_replays += 1
            else:
                self._iteration()
                self._warm += 1
                self._graph_wkey = wkey
            for a in acts:  # (host mirrors of what moved on the device: a later re-layout continues the levels' own streams)
                a.env_steps += a.E
                a._explore_counter += 1
                a.scenario_seed += 1
            return
        inputs = [a.policy_inputs() for a in acts]
        outs = self.model.step_levels([(a.E, a.N, a.pos, a.hidden, cm) for a, (cm, _) in zip(acts, inputs)], self.obs_all, self.latents)
        for a, (cm, packed), (actions, q, hidden, _) in zip(acts, inputs, outs):
            a.act(actions, q, hidden, cm, packed)

    @property
    def env_steps(self):
        return self.retired_env_steps + sum(a.env_steps for a in self.actors.values())

    @property
    def episodes(self):
        return sum(a.episodes for a in self.actors.values())
