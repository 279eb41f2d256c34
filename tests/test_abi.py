"""CPU: the C-ABI library loads without a GPU and exports every symbol include/mapf_env.h declares;
host-only entry points (generator, error strings, argument checks) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in ("mapf_env.h", "mapf_replay.h", "mapf_dqn.h", "mapf_search.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(mapf_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported_and_bound():
    from mapf_rl_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libmapf_env.so does not export %s" % n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(names), (bound ^ set(names))


def test_abi_version_and_strerror():
    from mapf_rl_amd._lib import lib

    assert lib.mapf_abi_version() == 2
    assert lib.mapf_strerror(0) == b"ok"
    assert b"action" in lib.mapf_strerror(-2)
    assert b"unique" in lib.mapf_strerror(-3)


def test_create_argument_checks_without_gpu():
    from mapf_rl_amd import _lib

    h = ctypes.c_void_p()
    lib = _lib.lib
    assert lib.mapf_create(0, 8, 2, 4, 0, ctypes.byref(h)) == _lib.ERR_INVALID_ARG
    assert lib.mapf_create(1, 65, 2, 4, 0, ctypes.byref(h)) == _lib.ERR_UNSUPPORTED
    assert lib.mapf_create(1, 8, 256, 4, 0, ctypes.byref(h)) == _lib.ERR_UNSUPPORTED
    assert lib.mapf_create(1, 8, 2, 3, 0, ctypes.byref(h)) == _lib.ERR_UNSUPPORTED
    if lib.mapf_device_count() == 0:
        assert lib.mapf_create(1, 8, 2, 4, 0, ctypes.byref(h)) == _lib.ERR_HIP
        assert not h.value


def test_product_does_not_import_oracle():
    """the product package must never route through oracle/ (it is test infrastructure)"""
    pkg = os.path.join(ROOT, "mapf_rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "mapf_oracle" not in text, f


def test_vec_environment_fails_loudly_without_gpu():
    import torch

    import mapf_rl_amd as M

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        M.VecEnvironment(1, 8, 2)


def test_config_attribute_compatibility():
    """config.py keeps every attribute name and value of the reference's config.py (values listed here so the
    check also runs where the reference is absent)."""
    import config

    expect = dict(env_level=0, map_length=20, num_agents=6, obs_radius=4, obs_shape=(6, 9, 9), training_times=1000000,
                  save_interval=2500, gamma=0.99, batch_size=192, learning_starts=50000, target_network_update_freq=2500,
                  save_path='./models', max_steps=256, bt_steps=16, load_model=None, local_buffer_size=256,
                  global_buffer_size=1024 * 256, actor_update_steps=400, grad_norm_dqn=40, prioritized_replay_alpha=0.6,
                  prioritized_replay_beta=0.4, double_q=False, init_set=(1, 10), max_num_agetns=6, max_map_lenght=40,
                  pass_rate=0.9, cnn_channel=64, latent_dim=256, max_comm_agents=3, num_comm_layers=2, num_comm_heads=2,
                  forward_steps=2)
    for k, v in expect.items():
        assert getattr(config, k) == v, k
    assert config.reward_fn == dict(move=-0.075, stay_on_goal=0, stay_off_goal=-0.075, collision=-0.5, finish=3)
    assert list(config.reward_fn) == ["move", "stay_on_goal", "stay_off_goal", "collision", "finish"]
