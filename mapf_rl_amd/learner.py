"""DQN batch update of the reference (`Learner.train` body, reference worker.py:282-344) on one MI355X, with
an optional data-parallel gradient all-reduce across the GPUs of a node.

One `update()` = worker.py:287-338: sample a prioritized batch of [B, 18, A] windows from the device replay,
target = r + 0.99^steps * (1 - done) * max_a Q_target(window, bt+steps)  (1-step reward even when steps = 2:
quirk Q2; no double-Q: quirk Q6 -- `Learner(double_q=True)` is an opt-in, see __init__), Huber(kappa=1) loss weighted by the IS weights, Adam(1e-4) with
MultiStepLR(100k, 300k, x0.5), global grad-norm clip 40, new priorities |td| back into the sum tree, target
sync + checkpoint every 2500 updates.

Deviations (declared): bf16 autocast instead of fp16 autocast + GradScaler (bf16 needs no loss scaling;
on CPU both are fp32); the inner counter that restarts every 10,000 iterations in the reference
(worker.py:285,336) is the plain update counter here.

Multi-GPU: every rank owns its environments, replay ring and sum tree; the only exchange
is ONE all-reduce per update of the flat 2,050,582-element fp32 gradient bucket (8.2 MB) over RCCL/xGMI
(`torch.distributed`, backend "nccl"), averaged over ranks.  Parameter .grad tensors are views into the
bucket, so there is no flatten/unflatten copy."""
import os
from copy import deepcopy

import torch
import torch.nn as nn

from .model import Network

GAMMA = 0.99          # hard-coded in the reference (worker.py:306), config.gamma is dead
GRAD_CLIP = 40.0      # worker.py:319
TARGET_SYNC = 2500    # config.target_network_update_freq (config.py:27)
FORWARD_STEPS = 2     # config.forward_steps (config.py:65)
# tests / tools: treat an initialised ONE-rank process group as several ranks, i.e. run the multi-rank code path -- the collective
# calls (RCCL on the one GPU of a test box), the stream orders around them, the graph stages split around the exchange
FORCE_EXCHANGE = os.environ.get("MAPF_FORCE_EXCHANGE", "0") == "1"


def exchanging(group=None):
    """True when an update has to all-reduce its gradients: a process group with more than one rank (or FORCE_EXCHANGE)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_EXCHANGE


def huber_loss(td_error, kappa=1.0):
    """worker.py:341-344"""
    a = td_error.abs()
    flag = (a < kappa).to(td_error.dtype)
    return flag * a.pow(2) * 0.5 + (1 - flag) * (a - 0.5)


class FlatGradBucket:
    """All parameter gradients as views of one contiguous fp32 buffer -> one collective per update."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        for p, g in zip(self.params, self._views()):
            p.grad = g

    @staticmethod
    def _view_like(flat_slice, p):
        """A view of the flat slice with p's shape AND strides (channels_last conv weights keep NHWC strides,
        so autograd accumulates without a layout-converting copy)."""
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            o, i, h, w = p.shape
            return flat_slice.view(o, h, w, i).permute(0, 3, 1, 2)
        return flat_slice.view_as(p)

    def zero(self):
        self.flat.zero_()
        for p, g in zip(self.params, self._views()):  # autograd may have replaced .grad (it does not, but be safe)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def _views(self):
        off = 0
        for p in self.params:
            yield self._view_like(self.flat[off:off + p.numel()], p)
            off += p.numel()

    # ---- the data-parallel exchange: ONE buffer, reduced in (up to) two pieces ----
    # The encoder's gradients (the first `split` elements in both layouts: module order and update.PARAM_ORDER start with
    # obs_encoder.*; 88 % of the bytes) are the LAST thing a backward pass produces; recurrence + head (12 %) are final before the
    # encoder's backward chain starts.  `begin(lo, hi)` issues the asynchronous all-reduce of flat[lo:hi] (on the backend's own
    # stream, behind what the current stream has queued), `finish()` waits for the pieces and divides by the world size -- the
    # fused update calls begin(split, n) in front of the encoder backward (update.FusedUpdate._backward), the autograd path reduces
    # the same two pieces back to back.  Nothing happens on one rank.
    split = None       # element index where the encoder's gradients end (None: one piece)
    pieces = 0         # collectives issued by the last exchange (tests)
    _pending = ()
    # bench.py / tools: a list that begin() / finish() append to -- ("begin", host seconds, None, None) per piece issued,
    # ("finish", host seconds, HIP event in front of the waits, HIP event behind the division) per exchange; the two events sit on the
    # update's stream, so their distance is the EXPOSED part of the exchange (what travelled beside the backward chain does not show)
    timing = None

    def _world(self, group=None):
        import torch.distributed as dist

        return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def _host_staged(self, group=None):
        """True when the group's backend has no device collectives (gloo: the CPU-side tests and the one-GPU rehearsal of the
        multi-rank path) and the buffer lives on a HIP device.  PyTorch's own gloo path for device tensors allocates a fresh pinned
        buffer per call; with two streams busy those allocations (hipHostMalloc beside running kernels) stalled an update for
        0.1-1 s at random (profiles/r05_two_rank_probe.md), so the staging is done here, through ONE persistent pinned buffer."""
        import torch.distributed as dist

        return self.flat.is_cuda and dist.get_backend(group) == "gloo"

    def begin(self, lo, hi, group=None):
        import torch.distributed as dist

        if not (exchanging(group) and hi > lo):
            return
        if self.timing is not None:
            import time

            t0 = time.perf_counter()
            try:
                return self._begin(lo, hi, group)
            finally:
                self.timing.append(("begin", time.perf_counter() - t0, None, None))
        return self._begin(lo, hi, group)

    def _begin(self, lo, hi, group=None):
        import torch.distributed as dist

        if self._host_staged(group):
            if getattr(self, "_stage", None) is None or self._stage.numel() != self.flat.numel():
                self._stage = torch.empty(self.flat.numel(), dtype=self.flat.dtype, pin_memory=True)
            self._stage[lo:hi].copy_(self.flat[lo:hi], non_blocking=True)  # behind the gradients, on the stream this is issued on
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            self._pending = tuple(self._pending) + ((lo, hi, ev),)
            return
        self._pending = tuple(self._pending) + (dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True),)

    def finish(self, group=None):
        import torch.distributed as dist

        w = self._world(group)
        self.pieces = len(self._pending)
        timed = self.timing is not None and (self._pending or w > 1)
        if timed:
            import time

            t0, e0, e1 = time.perf_counter(), None, None
            if self.flat.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.flat.device))
        for work in self._pending:
            if isinstance(work, tuple):  # host-staged piece: wait for its copy, reduce on the host, hand it back on this stream
                lo, hi, ev = work
                ev.synchronize()
                dist.all_reduce(self._stage[lo:hi], op=dist.ReduceOp.SUM, group=group)
                self.flat[lo:hi].copy_(self._stage[lo:hi], non_blocking=True)
            else:
                work.wait()
        self._pending = ()
        if w > 1:
            self.flat.div_(w)
        if timed:
            if e1 is not None:
                e1.record(torch.cuda.current_stream(self.flat.device))
            self.timing.append(("finish", time.perf_counter() - t0, e0, e1))

    def all_reduce_mean(self, group=None):
        n = self.flat.numel()
        if exchanging(group):
            k = self.split if (self.split is not None and 0 < self.split < n) else n
            if k < n:
                self.begin(k, n, group)
            self.begin(0, k, group)
        self.finish(group)


class _FlatView:
    """FlatGradBucket's interface over a gradient buffer that already is flat (update.FlatParams.grads)."""

    def __init__(self, flat, split=None):
        self.flat, self.split = flat, split

    def zero(self):
        self.flat.zero_()

    pieces, _pending, timing = 0, (), None
    _world, begin, _begin, finish, all_reduce_mean, _host_staged = (FlatGradBucket._world, FlatGradBucket.begin, FlatGradBucket._begin, FlatGradBucket.finish,
                                                                     FlatGradBucket.all_reduce_mean, FlatGradBucket._host_staged)


class Learner:
    FUSED_UPDATE = True  # HIP device: the update as an explicit forward / backward over the kernels (update.FusedUpdate)

    def __init__(self, buffer=None, device=None, batch_size=192, lr=1e-4, milestones=(100000, 300000), save_path="./models",
                 model=None, prefetch=True, double_q=False):
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.model = (model if model is not None else Network()).to(self.device)
        self.tar_model = deepcopy(self.model)
        for p in self.tar_model.parameters():
            p.requires_grad_(False)
        self.base_lr, self.milestones = lr, tuple(milestones)
        self.grad_hook = None  # tests: called with the learner after backward (+ all-reduce), before the clip (worker.py:316-319)
        # actors that write the replay from ANOTHER stream (train.py --overlap-actors): `replay_gate`, if set, is an event this update's
        # replay operations (priority write-back, next prioritized sample) wait for; `replay_released` is recorded behind them
        self.replay_gate = self.replay_released = None
        self._fused = None
        if self.device.type == "cuda" and self.FUSED_UPDATE:
            from .update import FusedUpdate

            # parameters, gradients and Adam moments as flat buffers; clip + Adam (worker.py:260,319-322) are one kernel pair there
            self._fused = FusedUpdate(self)
            self.bucket = _FlatView(self._fused.flat.grads, split=self._fused.flat.offsets["recurrent.weight_ih"])
            self.optimizer = self.scheduler = None
        else:
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)                       # worker.py:260
            self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(milestones), gamma=0.5)  # :261
            self.bucket = FlatGradBucket(self.model.parameters())
            self.bucket.split = sum(p.numel() for p in self.model.obs_encoder.parameters() if p.requires_grad)
        self.buffer, self.batch_size, self.save_path = buffer, batch_size, save_path
        # config.double_q (config.py:46) is dead in the reference (quirk Q6: worker.py:300-303 always takes max_a Q_target).
        # Opt-in here for BASELINE config 5 ("prioritized replay + double-DQN"): the ONLINE network picks the action, the
        # TARGET network values it (van Hasselt et al.); off by default = the reference's update.
        self.double_q = bool(double_q)
        self.counter = self.last_counter = 0
        self.loss = 0.0
        self.done = False
        # Pipelining across updates (the reference prepares batches ahead too, in a background thread: worker.py:47-58): as soon
        # as update k has written its priorities, batch k + 1 is sampled and PLANNED: the rows of its two windows that can reach
        # agent 0's Q-value (model.relevance) are marked on the device and their counts copied to pinned host memory, so that
        # update k + 1 knows its encoder batch sizes without waiting for the GPU.  Update k + 1 then runs the TARGET-network
        # forward on a second HIP stream beside the online forward.  Same numbers as the sequential order: priorities are
        # written before the sample.
        # (Round 1 launched the target forward of batch k + 1 already during update k's backward; with ~1/9 of the observations
        # left to encode, the host's wait for the row count cost more than that overlap gave.)
        # (Measured, tools/learner_modes.py: 39.5 ms per update without prefetch, 38.7 with it; sampling one update EARLIER and
        # running the update on a high-priority stream -- so that the target forward would only fill idle CUs -- gave 39.7: the chip
        # is saturated by the update's own kernels, the update is ~39.5 ms of work whichever way it is ordered.)
        self.prefetch = bool(prefetch) and buffer is not None and self.device.type == "cuda"
        # (a plain second stream.  PyTorch maps `priority` to a stream pool by clamp(-priority, 0, max-1): a positive value does NOT make a
        # lower-priority HIP stream, so rounds 3-4's `priority=1` was the default pool all along and the knob is gone.)
        # (MAPF_LEARNER_STREAM_PRIORITY=-1: a high-priority second stream, for callers that run update() on a high-priority stream of
        # their own beside actors on a normal one -- train.py --learner-priority, tools/micro/train_loop_overlap.py)
        # ONE second stream per device, whoever builds learners (streams.py: streams are multiplexed onto four hardware queues)
        from .streams import role_stream

        self._side = role_stream(self.device, "learner_side", int(os.environ.get("MAPF_LEARNER_STREAM_PRIORITY", "0"))) if self.prefetch else None
        self._pre = None

    def current_lr(self):
        """MultiStepLR(milestones, gamma = 0.5) (worker.py:261) for the step that follows `counter` completed ones."""
        return self.base_lr * 0.5 ** sum(1 for m in self.milestones if m <= self.counter)

    # ------------------------------------------------------------------ one update
    def target_q(self, batch, rows=None):
        """(1 - done) max_a Q_target(s_{t+steps}) [B,1] (worker.py:300-303): the part of an update that needs no gradient."""
        b_obs, _, _, b_done, b_steps, b_bt_steps, b_hidden, b_comm_mask = batch[:8]
        b_next_bt_steps = b_bt_steps + b_steps.view(-1).to(b_bt_steps.dtype)
        with torch.no_grad():
            q_tar = self.tar_model.bootstrap(b_obs, b_next_bt_steps, b_hidden, b_comm_mask, rows=rows)
            if self.double_q:
                pick = self.model.bootstrap(b_obs, b_next_bt_steps, b_hidden, b_comm_mask, rows=rows).argmax(1, keepdim=True)
                return (1 - b_done) * q_tar.gather(1, pick)
            return (1 - b_done) * q_tar.max(1, keepdim=True)[0]

    def compute_td(self, batch, q_next=None, rows=None):
        """worker.py:296-306 on an 11-tuple from GlobalBuffer.sample_batch. Returns (td_error [B,1], q [B,1], q_next [B,1]).
        `q_next` may be a callable that yields it (the learner's side-stream hand-over); `rows`: see Network.bootstrap."""
        b_obs, b_action, b_reward, b_done, b_steps, b_bt_steps, b_hidden, b_comm_mask = batch[:8]
        if q_next is None:
            q_next = self.target_q(batch)
        q = self.model.bootstrap(b_obs[:, :-FORWARD_STEPS], b_bt_steps, b_hidden, b_comm_mask[:, :-FORWARD_STEPS], rows=rows).gather(1, b_action)
        if callable(q_next):
            q_next = q_next()
        td = q - (b_reward + (GAMMA ** b_steps) * q_next)
        return td, q, q_next

    # -- which observations an update has to encode, known one update ahead --
    def _plan(self, batch):
        """Marks the reachable rows of the batch's online and target windows (model.relevance) and starts the copy of their counts
        to pinned host memory; nothing here waits for the GPU."""
        from .model import Network, relevance

        if self._fused is not None and self._fused.usable(batch):
            return self._fused.plan(batch)
        if not (Network.PRUNE_UNREACHABLE and batch[7].is_cuda):
            return None
        comm, bt, steps = batch[7], batch[5], batch[4].view(-1)
        rel_o = relevance(comm[:, :-FORWARD_STEPS], bt)
        rel_t = relevance(comm, bt + steps.to(bt.dtype))
        counts = torch.stack([rel_o.sum(), rel_t.sum(), rel_o[0].sum(dim=1).max(), rel_t[0].sum(dim=1).max()])
        host = torch.empty(4, dtype=counts.dtype, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        return rel_o, rel_t, host, done, counts

    @staticmethod
    def _rows(plan):
        """Row indices of a planned batch: (online window, target window).  Waits for the count copy -- long finished when the
        batch was planned during the update before."""
        if plan is None:
            return None, None
        from .model import Reach

        rel_o, rel_t, host, done, _ = plan
        done.synchronize()
        n_o, n_t, k_o, k_t = (int(v) for v in host.tolist())
        return (Reach(torch.nonzero_static(rel_o.view(-1), size=n_o).squeeze(1), rel_o[0], k_o),
                Reach(torch.nonzero_static(rel_t.view(-1), size=n_t).squeeze(1), rel_t[0], k_t))

    def update(self, batch=None):
        """One Learner.train iteration (worker.py:287-338).  `batch` defaults to a fresh prioritized sample."""
        return self._update(batch)

    def _sample(self):
        """A prioritized sample -- into the fused update's static batch buffers when it replays captured graphs (update.FusedUpdate)."""
        slot = self._fused.batch_slot() if self._fused is not None else None
        return self.buffer.sample_batch(self.batch_size, out=slot)

    def _launch_prefetch(self):
        nxt = self._sample()
        self._pre = (nxt, self._plan(nxt))

    def _update(self, batch=None):
        own_batch = batch is None
        plan = None
        if own_batch and self._pre is not None:
            batch, plan = self._pre
            self._pre = None
        elif own_batch:
            if self.replay_gate is not None and self.device.type == "cuda":
                torch.cuda.current_stream(self.device).wait_event(self.replay_gate)
            batch = self._sample()
        if self._fused is not None and self._fused.usable(batch):
            out = self._fused.run(batch, plan if isinstance(plan, dict) else None, own_batch)
            self.last_path = "fused"
            self.counter += 1
            self._last = (out["loss"], out["grad_norm"])
            if self.counter % TARGET_SYNC == 0:                                                  # worker.py:336-338
                self.sync_target()
                self.save()
            return out
        if isinstance(plan, dict):
            plan = None
        fallback = self._fused is not None  # a batch outside the fused kernels' limits on a Learner that was built for them
        self.last_path = "autograd"
        if fallback:
            self._say_fallback(batch)
            # ONE optimizer state whichever path takes a step: torch's Adam runs on the flat buffers' own moment tensors (views), at
            # the step count and learning rate the fused path is at, and the count moves on afterwards
            flat = self._fused.flat
            flat.sync()
            if self.optimizer is None:
                self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.base_lr)
                for name, p in zip(flat.names, flat._plist):
                    self.optimizer.state[p] = dict(step=torch.tensor(float(flat.step)), exp_avg=flat._as_param(flat.exp_avg, name),
                                                   exp_avg_sq=flat._as_param(flat.exp_avg_sq, name))
            for p in flat._plist:
                self.optimizer.state[p]["step"].fill_(float(flat.step))
            for g in self.optimizer.param_groups:
                g["lr"] = self.current_lr()
        q_next = None
        rows_o = None
        if own_batch and self.prefetch:
            if plan is None:
                plan = self._plan(batch)
            rows_o, rows_t = self._rows(plan)
            # the target network's forward on the second stream, beside the online forward
            cur = torch.cuda.current_stream(self.device)
            if self.double_q:
                # the side stream runs the ONLINE network too (the arg-max): its packed weight images are built here, on the
                # main stream, before the side stream starts behind it -- packing is lazy and keyed on the host, so whichever
                # stream came first would pack and the other would read the image before the pack kernel had run
                self.model.prepack()
            self._side.wait_stream(cur)
            for t in list(batch) + ([rows_t.rows, rows_t.agents] if rows_t is not None else []):
                if torch.is_tensor(t) and t.is_cuda:  # allocated on this stream, read on the side stream: keep the allocator from reusing them early
                    t.record_stream(self._side)
            with torch.cuda.stream(self._side):
                qn = self.target_q(batch, rows_t)
                ready = torch.cuda.Event()
                ready.record(self._side)
            qn.record_stream(cur)  # allocated on the side stream, consumed on this one

            def q_next():
                cur.wait_event(ready)
                return qn
        idxes, weights, old_ptr = batch[8], batch[9], batch[10]
        td, q, q_next = self.compute_td(batch, q_next, rows_o)
        priorities = td.detach().view(-1).abs().clamp(1e-6)                                      # worker.py:308
        loss = (weights * huber_loss(td)).mean()                                                 # worker.py:310
        if self.replay_gate is not None and self.device.type == "cuda":
            torch.cuda.current_stream(self.device).wait_event(self.replay_gate)
        if self.buffer is not None and idxes is not None:
            self.buffer.update_priorities(idxes, priorities, old_ptr)                            # worker.py:331 (values known here)
        if self.prefetch and own_batch:
            self._launch_prefetch()
        if self.device.type == "cuda":
            self.replay_released = torch.cuda.Event()
            self.replay_released.record(torch.cuda.current_stream(self.device))
        self.bucket.zero()
        loss.backward()
        self.bucket.all_reduce_mean()                                                            # the only collective
        if self.grad_hook is not None:
            self.grad_hook(self)
        grad_norm = nn.utils.clip_grad_norm_(self.model.parameters(), GRAD_CLIP)                 # worker.py:319
        self.optimizer.step()
        if fallback:
            self._fused.flat.step = self._fused.flat.step + 1
            self._fused.flat.refresh_bf16()
            self.model.weights_epoch += 1
        else:
            self.scheduler.step()
        self.counter += 1
        self._last = (loss.detach(), grad_norm.detach())
        if self.counter % TARGET_SYNC == 0:                                                      # worker.py:336-338
            self.sync_target()
            self.save()
        return dict(loss=loss.detach(), td=td.detach(), priorities=priorities, grad_norm=grad_norm.detach(), q=q.detach(),
                    q_next=q_next)

    last_path = None   # "fused" | "autograd": which path the last update took (bench.py: learner_path)
    _fallback_said = ()

    def _say_fallback(self, batch):
        """A Learner built for the hand-written kernels whose batch leaves their shape limits runs through PyTorch (autograd, library
        GEMMs): correct, several times slower -- said ONCE per (shape, reason) on stderr, never silently."""
        import sys

        obs = batch[0]
        why = self._fused.why_not(batch)
        key = (tuple(obs.shape[:3]), why)
        if key not in self._fallback_said:
            self._fallback_said = tuple(self._fallback_said) + (key,)
            print("mapf_rl_amd.learner: update on a batch of shape B x T x A = %s leaves the fused path (%s): running it through "
                  "PyTorch autograd" % (" x ".join(str(v) for v in obs.shape[:3]), why), file=sys.stderr, flush=True)

    def _drop_prefetch(self):
        """Forget the batch sampled ahead (it carries no network output, so a weight change does not invalidate it; for callers
        that change what a plan means, e.g. Network.PRUNE_UNREACHABLE)."""
        self._pre = None

    def sync_target(self):
        self.tar_model.load_state_dict(self.model.state_dict())

    def load_state_dict(self, state_dict, sync_target=True):
        """Loads online-network weights (reference key names) and, by default, copies them to the target network."""
        self.model.load_state_dict(state_dict)  # (in place: the parameters stay views of the flat buffers)
        if sync_target:
            self.tar_model.load_state_dict(self.model.state_dict())

    def save(self, path=None):
        """Checkpoint with the reference's key names (worker.py:338); the directory is created if missing."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return None
        os.makedirs(self.save_path, exist_ok=True)
        path = path or os.path.join(self.save_path, "{}.pth".format(self.counter))
        torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, path)
        return path

    def stats(self, interval):
        """worker.py:347-352"""
        if hasattr(self, "_last"):
            self.loss = float(self._last[0])
        print("number of updates: {}".format(self.counter))
        print("update speed: {}/s".format((self.counter - self.last_counter) / interval))
        print("loss: {}".format(self.loss))
        self.last_counter = self.counter
        return self.done
