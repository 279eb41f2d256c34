"""GPU: mapf_rl_amd.streams -- one HIP stream per role and device (the learner's second stream, the actors' stream, the scenario staging
stream, one capture stream for the learner's graphs and one for the actors'): the same object for the same role whoever asks, different
streams for different roles (two graphs captured on ONE stream take turns when they are replayed into two streams: the learner's and the
actors' capture streams must differ), and every object of the package draws its streams from there."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_stream_per_role():
    from mapf_rl_amd import streams

    dev = torch.device("cuda", 0)
    got = {r: streams.role_stream(dev, r) for r in streams.ROLES}
    assert all(streams.role_stream("cuda", r) is got[r] for r in streams.ROLES)          # same object, whatever spelling of the device
    assert len({s.cuda_stream for s in got.values()}) == len(streams.ROLES)                 # different streams for different roles
    assert all(s.cuda_stream != torch.cuda.default_stream(dev).cuda_stream for s in got.values())
    assert got["capture_learner"].cuda_stream != got["capture_actors"].cuda_stream
    hp = streams.role_stream(dev, "learner_side", priority=-1)                              # (a priority is part of the key)
    assert hp is streams.role_stream(dev, "learner_side", priority=-1) and hp is not got["learner_side"]
    with pytest.raises(AssertionError):
        streams.role_stream(dev, "no_such_role")


def test_the_package_draws_its_streams_from_the_registry():
    import mapf_rl_amd as M
    from mapf_rl_amd import streams
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    dev = torch.device("cuda", 0)
    buf = GlobalBuffer(64, max_agents=6, device=dev, init_set=(2, 10), fixed_level=True)
    a, b = Learner(buf, device=dev, batch_size=8), Learner(buf, device=dev, batch_size=8, model=Network())
    assert a._side is b._side is streams.role_stream(dev, "learner_side")
    env = M.VecEnvironment(8, 10, 2, device=dev)
    maps, agents, goals, _ = M.generate_scenarios(8, 10, 2, 0.2, seed=3)
    env.load(maps, agents, goals)
    act = VecActor(env, a.model, buf, seed=1)
    assert act._stage_stream is streams.role_stream(dev, "actor_stage")
