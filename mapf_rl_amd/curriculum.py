"""Curriculum of (num_agents, map_length) levels: the reference's adaptive training schedule
(reference worker.py:74-82 statistics, :205-226 level promotion, :237-250 stop criterion; config.py:49-52;
Environment.reset(level) environment.py:148-151).

The reference's 16 actors each draw a random level per episode.  Here every active level owns a
`VecEnvironment` + `VecActor` of `envs_per_level` lock-step environments; all of them record into ONE device
replay whose rows are laid out for `config.max_num_agetns` agents (smaller levels are a zero-padded prefix of
the row, see actor.py).  `sync_levels()` follows `GlobalBuffer.level` after every `stats()` call: levels that
appeared get actors, levels that were promoted away are retired."""
import torch

from .actor import VecActor
from .environment import VecEnvironment, generate_scenarios


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
