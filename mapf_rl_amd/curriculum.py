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
    def __init__(self, model, buffer, envs_per_level=256, device=None, seed=0, max_steps=256, reward_fn=None):
        self.model, self.buffer = model, buffer
        self.envs_per_level, self.seed, self.max_steps, self.reward_fn = envs_per_level, seed, max_steps, reward_fn
        self.device = buffer.device if device is None else torch.device(device)
        self.actors = {}
        self.retired_env_steps = 0
        self.sync_levels()

    def _make(self, key):
        n, L = key
        E = self.envs_per_level
        env = VecEnvironment(E, L, n, reward_fn=self.reward_fn, device=self.device)
        maps, agents, goals, _ = generate_scenarios(E, L, n, -1.0, seed=self.seed * 7919 + n * 131 + L)
        env.load(maps, agents, goals)
        return VecActor(env, self.model, self.buffer, max_steps=self.max_steps, seed=self.seed + n * 1000 + L, density=-1.0)

    def sync_levels(self):
        """Create actors for new levels, drop actors of levels no longer in GlobalBuffer.level (worker.py:224)."""
        want = [tuple(k) for k in self.buffer.get_level()]
        for key in want:
            if key not in self.actors:
                self.actors[key] = self._make(key)
        for key in list(self.actors):
            if key not in want:
                self.retired_env_steps += self.actors[key].env_steps
                del self.actors[key]
        return want

    def step(self):
        for actor in self.actors.values():
            actor.step()

    @property
    def env_steps(self):
        return self.retired_env_steps + sum(a.env_steps for a in self.actors.values())

    @property
    def episodes(self):
        return sum(a.episodes for a in self.actors.values())
