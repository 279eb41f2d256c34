"""One HIP stream per ROLE and device, shared by every object that plays that role.

Why: the runtime multiplexes HIP streams onto a handful of hardware queues (four by default), assigned as streams are created.  A
process that keeps creating streams -- bench.py builds a second learner + actor set for its reference-shape legs behind the config-2
ones -- ends up with two streams that are meant to run side by side (the learner's and the actors') on the SAME hardware queue, where
they take turns: measured, the train loop at the reference's training shape read 3.96 ms per pair behind the config-2 legs and 3.44 ms in
a fresh process (profiles/r06_train_loop_hw_queues.txt).  With one stream per role the process holds at most five streams besides the
default one, created in a fixed order, whatever it builds and tears down.

Sharing a stream between two objects of the same role only adds ordering between them (they are used one after the other anyway); it
never removes an ordering either of them relies on."""
import torch

import os

_streams = {}

ROLES = ("learner_side", "actors", "actor_stage", "capture_learner", "capture_actors")
# creation order of a device's role streams ("-" = a stream nobody uses: it only takes its turn in the runtime's assignment of streams
# to hardware queues).  The variable: experiments (tools/micro/train_loop_overlap.py).
ORDER = tuple(os.environ.get("MAPF_STREAM_ORDER", "learner_side,capture_actors,capture_learner,actors,actor_stage").split(","))
_spacers = []


def role_stream(device, role, priority=0):
    """The stream of `role` on `device`.  ALL of a device's role streams are created together, in ORDER, the first time one is asked
    for: which hardware queue a stream lands on follows from its creation index, so the arrangement is the same in every process.
    Roles: learner_side (target network / prefetch stage beside the update), actors (the actor iteration beside the update: train.py
    --overlap-actors, bench.py), actor_stage (scenarios drawn ahead), capture_learner / capture_actors (the stream the learner's / the
    actors' graphs are captured on; never carries replayed work -- but a replayed graph keeps an affinity to the hardware queue of the
    stream it was captured on: two graphs captured on ONE stream take turns when they are replayed into two streams, measured: the
    actors' iteration beside the update 3.94 ms per pair with a shared capture stream, 3.43 ms with two,
    profiles/r06_train_loop_capture_streams.txt)."""
    assert role in ROLES, role
    if os.environ.get("MAPF_STREAM_REGISTRY", "1") == "0":  # (experiments: a fresh stream per call, as rounds 1-5 did)
        return torch.cuda.Stream(device=device, priority=int(priority))
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    if not any(k[0] == device.index for k in _streams):
        for r in ORDER:
            s = torch.cuda.Stream(device=device)
            if r in ROLES:
                _streams[(device.index, r, 0)] = s
            else:
                _spacers.append(s)
    key = (device.index, role, int(priority))
    s = _streams.get(key)
    if s is None:
        s = _streams[key] = torch.cuda.Stream(device=device, priority=int(priority))
    return s
