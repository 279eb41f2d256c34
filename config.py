"""Configuration module kept attribute-compatible with the reference's `config.py` (reference config.py:1-65):
every name the reference defines exists here with the same value, including the historical misspellings
`max_num_agetns` / `max_map_lenght`, because `train.py` / `test.py` / user code read them as `config.<name>`.

The values live in grouped tables and are published as module attributes below; which of them the
reference actually reads is noted per group (several are dead there).
"""

_REWARDS = {            # reference config.py:8-12; order = MAPF_RC_* classes of include/mapf_env.h
    "move": -0.075,
    "stay_on_goal": 0,
    "stay_off_goal": -0.075,
    "collision": -0.5,
    "finish": 3,
}

_ENVIRONMENT = {
    "env_level": 0,             # dead in the reference
    "map_length": 20,
    "num_agents": 6,
    "obs_radius": 4,            # the model hard-wires a 9x9 field of view (obs_shape below)
    "reward_fn": _REWARDS,
    "obs_shape": (6, 9, 9),
}

_TRAINING = {
    "training_times": 1000000,  # dead in the reference; train.py here uses it as the default --max-updates
    "save_interval": 2500,
    "gamma": 0.99,              # dead in the reference (0.99 is hard-coded in worker.py:306 and buffer.py:12)
    "batch_size": 192,
    "learning_starts": 50000,
    "target_network_update_freq": 2500,
    "save_path": "./models",
    "max_steps": 256,
    "bt_steps": 16,
    "load_model": None,         # dead in the reference
    "actor_update_steps": 400,
    "grad_norm_dqn": 40,        # dead in the reference (40 is hard-coded in worker.py:319)
    "prioritized_replay_alpha": 0.6,
    "prioritized_replay_beta": 0.4,
    "double_q": False,          # dead in the reference (worker.py:300-303 always takes max Q_target); here: opt-in double-DQN target (learner.py)
    "forward_steps": 2,
}

_CURRICULUM = {
    "init_set": (1, 10),
    "max_num_agetns": 6,
    "max_map_lenght": 40,
    "pass_rate": 0.9,
}

_NETWORK = {
    "cnn_channel": 64,          # accepted and ignored by Network (the encoder hard-codes 128 channels)
    "latent_dim": 256,
    "max_comm_agents": 3,       # including the agent itself
    "num_comm_layers": 2,
    "num_comm_heads": 2,
}

for _group in (_ENVIRONMENT, _TRAINING, _CURRICULUM, _NETWORK):
    globals().update(_group)

# derived values (reference config.py:33-34)
local_buffer_size = _TRAINING["max_steps"]
global_buffer_size = 1024 * local_buffer_size   # dead in the reference

del _group
