"""mapf_rl_amd -- MI355X-native vectorised MAPF environment + DQN hot path (drop-in for the
environment.step/observe/search-heuristic and DQN-update path of ZiyuanMa/MAPF_RL).

Importing this package loads the in-tree HIP library (mapf_rl_amd/libmapf_env.so); it raises
ImportError if the library has not been built -- there is no CPU fallback."""
from . import _lib  # noqa: F401  (fails loudly when the HIP extension is missing)
from .environment import Environment, VecEnvironment, generate_scenarios, action_list  # noqa: F401

__all__ = ["Environment", "VecEnvironment", "generate_scenarios", "action_list"]
