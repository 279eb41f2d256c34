"""One DQN batch update (reference worker.py:287-338, the body of `Learner.train`) on a HIP device as an explicit forward /
backward over the hand-written kernels -- no autograd graph, ~70 launches instead of ~300.

Same mathematics as `Learner._update` over `Network.bootstrap` (which stays the path on the CPU, in fp32, and for callers that want
autograd); what changes is who issues the work.  Through PyTorch operations an update was host-bound: 12.2 ms of enqueue time for
13 ms of wall time (tools/update_times.py), ~200 of its launches tiny element-wise / reduction / copy kernels around the dueling
head, the loss, gradient accumulation, the clip, Adam, weight re-packing and the bookkeeping of which observations to encode.  Here:

  plan (at sample time)   `mapf_plan_mark` x 2: per window the (step, agent) entries that can reach agent 0's Q-value (model.relevance)
                          and an order of the window's agents in which the entries needed at a step are a prefix;
  rows                    `mapf_plan_rows` x 2: observation rows to encode, row indices, communication masks and initial hidden
                          states in that compact numbering;
  forward                 encoder (forward-save kernel / inference kernel on the side stream for the target network), input
                          projection GEMM, row scatter, recurrence kernels;
  head + loss             `mapf_dqn_head_loss`: dueling head of both networks, TD error, priorities, Huber loss and their gradients;
  backward                BPTT kernel, the tall weight-gradient GEMMs written straight into the flat gradient buffer, encoder backward-
                          data chain + weight-gradient kernels;
  step                    `mapf_adam_step`: gradient norm, clip and Adam over the flat buffers, refreshing the bf16 parameter copy.

Parameters, gradients and Adam moments live in flat fp32 buffers (`FlatParams`): the module's parameters are views of them, so
`state_dict()` / `load_state_dict()` / checkpoints are unaffected, and the data-parallel all-reduce is one collective over one buffer."""
import contextlib
import ctypes
import os

import numpy as np
import torch

from ._lib import check, lib
from .fused import (capture_mode, no_gc_during_capture, warm_up_gemm_library, ENC_OBS_PER_BLOCK, ENC_WGRAD0_PARTS, ENC_WGRAD_PARTS, RECUR_NARROW_AGENTS, RECUR_WEIGHT_ELEMS, PackedEncoder,
                    PackedRecurrence, pack_encoder_backward, recurrence_params, rows_buffer, ENC_ELEMENT, INPROJ_PACKED_ELEMS, LATGRAD_PACKED_ELEMS,
                    input_proj_rows, latent_grad_rows, sum_parts_into, tall_tn_into)

GAMMA = 0.99
GRAD_CLIP = 40.0
# The target network's forward runs on the second stream beside the online forward only while neither fills the chip: measured (round 4,
# tools/update_timeline.py) at config 2 -- ~10-18 k rows per network -- the two encoders just share the MFMA time (2.17 ms together, 1.1
# each alone), and the two 192-workgroup recurrence kernels (one workgroup per CU each) make the head wait for the second of them:
# 7.0 ms serial against 7.3 side by side; at 6 agents (~6 k rows) side by side wins (3.08 against 3.16 ms).
SIDE_STREAM_MAX_ROWS = int(os.environ.get("MAPF_SIDE_MAX_ROWS", "8192"))
WGRAD_SPLIT = 2048  # rows per batch of the recurrence's split-K weight-gradient GEMMs (compact rows are padded to a multiple)
FORWARD_STEPS = 2
BETAS, EPS = (0.9, 0.999), 1e-8  # torch.optim.Adam defaults (worker.py:260)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


# the order of the flat buffers: W_Q / W_K / W_V adjacent (the fused q|k|v weight gradient is one view), the seven 128-channel
# convolution biases adjacent (their gradient is one [7, 128] reduction)
PARAM_ORDER = (
    "obs_encoder.0.weight", "obs_encoder.2.block1.weight", "obs_encoder.2.block2.weight", "obs_encoder.3.block1.weight",
    "obs_encoder.3.block2.weight", "obs_encoder.4.block1.weight", "obs_encoder.4.block2.weight", "obs_encoder.5.weight",
    "obs_encoder.0.bias", "obs_encoder.2.block1.bias", "obs_encoder.2.block2.bias", "obs_encoder.3.block1.bias",
    "obs_encoder.3.block2.bias", "obs_encoder.4.block1.bias", "obs_encoder.4.block2.bias", "obs_encoder.5.bias",
    "recurrent.weight_ih", "recurrent.weight_hh", "recurrent.bias_ih", "recurrent.bias_hh",
    "comm.self_attn.W_Q.weight", "comm.self_attn.W_K.weight", "comm.self_attn.W_V.weight",
    "comm.self_attn.W_Q.bias", "comm.self_attn.W_K.bias", "comm.self_attn.W_V.bias", "comm.self_attn.W_O.weight",
    "comm.update_cell.weight_ih", "comm.update_cell.weight_hh", "comm.update_cell.bias_ih", "comm.update_cell.bias_hh",
    "adv.weight", "adv.bias", "state.weight", "state.bias",
)


class FlatParams:
    """The parameters of a `Network` on a HIP device as views of ONE fp32 buffer, with a gradient buffer (the parameters' .grad are
    views of it), Adam's two moment buffers and a bf16 copy of the parameters, all in the same layout.  4-D convolution weights keep
    PyTorch's channels_last memory ([co][kh][kw][ci]): what the weight-gradient kernels write and the pack kernels read."""

    ALIGN = 8  # elements: 32 bytes in fp32, 16 in the bf16 copy

    def __init__(self, model):
        named = dict(model.named_parameters())
        assert set(named) == set(PARAM_ORDER), "FlatParams is laid out for mapf_rl_amd.model.Network"
        self.model = model
        self.names = list(PARAM_ORDER)
        self.offsets, off = {}, 0
        for k in self.names:
            self.offsets[k] = off
            off += -(-named[k].numel() // self.ALIGN) * self.ALIGN
        self.numel = off
        dev = named[self.names[0]].device
        assert dev.type == "cuda" and all(p.dtype == torch.float32 for p in named.values())
        self.device = dev
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.bf16 = torch.empty(off, dtype=torch.bfloat16, device=dev)
        self.scratch = torch.empty(256, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        # Adam's step count lives on the DEVICE (mapf_adam_step_dev increments it and derives the bias corrections from it: no host
        # scalar changes from update to update, so the launches can be replayed from a captured graph); step_host mirrors it
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.step_host = 0
        self.shapes = {k: tuple(p.shape) for k, p in named.items()}
        self._views = {}
        self._plist = [named[k] for k in self.names]  # (the parameter objects, in buffer order: no module-tree walk per update)
        for k in self.names:
            p = named[k]
            v = self._as_param(self.params, k)
            v.copy_(p.detach())
            p.data = v
            p.grad = self._as_param(self.grads, k)
        self.refresh_bf16()

    @property
    def step(self):
        return self.step_host

    @step.setter
    def step(self, value):
        self.step_host = int(value)
        self.step_dev.fill_(int(value))

    def _mem_shape(self, k):
        s = self.shapes[k]
        return (s[0], s[2], s[3], s[1]) if len(s) == 4 else s

    def mem(self, buf, k):
        """The slice of `buf` that belongs to parameter k, shaped like its MEMORY ([co, kh, kw, ci] for a convolution weight).
        (Views of the four persistent buffers: made once -- an update asks for ~30 of them and is host-bound at few agents.)"""
        key = (id(buf), k)
        v = self._views.get(key)
        if v is None or v.data_ptr() != buf.data_ptr() + self.offsets[k] * buf.element_size():
            n = int(np.prod(self.shapes[k]))
            v = buf[self.offsets[k]:self.offsets[k] + n].view(self._mem_shape(k))
            if buf is self.params or buf is self.grads or buf is self.exp_avg or buf is self.exp_avg_sq or buf is self.bf16:
                self._views[key] = v
        return v

    def _as_param(self, buf, k):
        v = self.mem(buf, k)
        return v.permute(0, 3, 1, 2) if len(self.shapes[k]) == 4 else v

    def span(self, buf, first, last):
        """One view over the adjacent parameters first..last (equal trailing shape, no padding in between: checked)."""
        n0 = self.offsets[first]
        n1 = self.offsets[last] + int(np.prod(self.shapes[last]))
        rows = sum(self.shapes[k][0] for k in self.names[self.names.index(first):self.names.index(last) + 1])
        tail = self.shapes[first][1:]
        assert (n1 - n0) == rows * int(np.prod(tail)) if tail else (n1 - n0) == rows
        return buf[n0:n1].view((rows,) + tuple(tail))

    def attached(self):
        """True while the module's parameters (and their .grad) still are the views made here (a `.to()` / `.float()` re-allocates)."""
        p0 = self._plist[0]
        return p0.data_ptr() == self.params.data_ptr() and p0.grad is not None and p0.grad.data_ptr() == self.grads.data_ptr()

    def refresh_bf16(self):
        check(lib.mapf_to_bf16(_ptr(self.params), _ptr(self.bf16), self.numel, _stream(self.device)), "mapf_to_bf16")
        self._versions = self._version_key()

    def _version_key(self):
        return tuple(p._version for p in self._plist)

    def sync(self):
        """Before an update: the module's parameters must still be the views made here, and the bf16 copy must follow whatever
        wrote them through PyTorch since the last step (load_state_dict, a test poking a weight)."""
        if not self.attached():
            raise RuntimeError("the Network's parameters were re-allocated (.to() / .float() / a second Learner on the same module) after "
                               "this Learner flattened them; build the Learner last")
        if self._version_key() != self._versions:
            self.refresh_bf16()

    def adam_step(self, lr):
        """clip_grad_norm_(40) + Adam over the whole buffer in two launches; returns the pre-clip gradient norm (device scalar)."""
        self.step_host += 1
        check(lib.mapf_adam_step_dev(self.numel, _ptr(self.params), _ptr(self.grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), _ptr(self.bf16),
                                     _ptr(self.scratch), _ptr(self.norm), float(lr), BETAS[0], BETAS[1], EPS, _ptr(self.step_dev), GRAD_CLIP,
                                     _stream(self.device)), "mapf_adam_step_dev")
        self.model.weights_epoch += 1  # the packed weight images of the fused kernels are stale now (fused.PackedEncoder)
        return self.norm[0].clone()


class WindowPlan:
    """What `mapf_plan_mark` knows about one window set of a batch (online: T - 2 steps ending at bt_steps; target: T steps ending
    `steps` later), plus -- once the host has the counts -- the compact tensors of `mapf_plan_rows`."""

    def __init__(self, T, B, N, dev):
        self.T, self.B, self.N = T, B, N
        self.slot = torch.empty((B, N), dtype=torch.int16, device=dev)
        self.order = torch.empty((B, N), dtype=torch.int16, device=dev)
        self.nact = torch.empty((T, B), dtype=torch.int32, device=dev)
        self.rows = self.nc = None


def _inner_contiguous(t, inner_dims):
    """t [B, T, ...]: the trailing `inner_dims` dimensions are laid out contiguously (any strides for B and T)."""
    exp = 1
    for d in range(t.dim() - 1, t.dim() - 1 - inner_dims, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            return False
        exp *= t.shape[d]
    return True


class _Ctx:
    """The tensors one update hands from stage to stage (online forward -> head -> backward)."""


def _round_up(n, step):
    return max(step, -(-int(n) // step) * step)


class FusedUpdate:
    # ---- replaying the update from captured HIP graphs (few-agent batches the learner samples itself) ----
    # At <= 16 agents per window -- the reference's own training shape (6) and every curriculum level -- an update is ~85 launches of
    # small kernels and was host-bound: 2.7-2.9 of its 3.55 ms were Python, ctypes and the caching allocator (round 3).  The sequence
    # is fixed except for its row counts (how many observations can reach agent 0's Q-value, how many of those are distinct), which
    # differ from batch to batch.  Graph mode rounds them UP to buckets and pads: the kernels run on bucket-sized buffers whose
    # padding rows are harmless by construction (their gradient is exactly zero, see _plan_rows / _online_forward / _backward), so a
    # handful of captured graphs covers every batch.  An update is eight graph launches over static interface buffers (round 6; five
    # in rounds 4-5):
    #   pack     (side stream)  the online network's weight images, behind the last optimizer step                 [one graph]
    #   rows_t, rows_o (main)   mapf_plan_rows of the target / online window set (beside `pack`)                   [key: target / online buckets]
    #   target   (side)         the target network's forward on the target window                -> a0_tg        [kept with rows_t]
    #   online   (main)         forward-save of the online network                               -> a0, saved    [kept with rows_o]
    #   head     (main)         dueling heads, TD error, loss, priorities                                         [one graph]
    #   prefetch (side)         priority write-back, next prioritized sample + its plan (closure marks, duplicate flags, counts -> pinned)
    #   backward (main)         BPTT, weight gradients, encoder backward, clip + Adam                              [kept with rows_o; key: lr]
    # with ordinary events between them (the actors' replay gate in front of `prefetch`, `replay_released` behind it), so the
    # overlap of actors and learner (train.py) is what it was -- except that the main stream no longer waits for the actors at all.
    # Memory.  Captures allocate from private pools, and a capture may reuse whatever an EARLIER capture of its pool has freed -- its
    # temporaries.  That is only sound if no graph that still writes such a temporary is replayed between the producer and the
    # consumer of a tensor a later capture placed there.  Hence three pools: "main" (pack, online, backward: the saved tensors of an
    # online graph live until its backward graph has run, and only `head` runs in between), "head", and "side" (target, prefetch:
    # nothing of theirs outlives its graph; their results land in buffers allocated outside any capture).
    TILE_WINDOWS = os.environ.get("MAPF_TILE_WINDOWS", "1") != "0"  # several few-agent windows per recurrence tile (_compact_width; the variable: A/B runs)
    GRAPH = os.environ.get("MAPF_UPDATE_GRAPH", "1") != "0"   # (the variable: A/B runs of train.py)
    GRAPH_MAX_AGENTS = 16       # replay rows wider than this are GPU-bound (no gain) and would need many more buckets
    GRAPH_ROW_STEP = 2048       # bucket of the entry count (== WGRAD_SPLIT: the weight-gradient GEMMs' K is padded to it anyway)
    GRAPH_UROW_STEP = int(os.environ.get("MAPF_GRAPH_UROW_STEP", "1024"))  # bucket of the distinct-observation count (the encoder kernels' batch; the variable: A/B runs)
    GRAPH_CACHE = 24            # captured graphs kept per stage; LRU beyond it
    GRAPH_CACHE_BYTES = int(float(os.environ.get("MAPF_UPDATE_GRAPH_GB", "48")) * (1 << 30))  # ... and what their private pools may hold
    # in all (an online entry pins its saved tensors -- ~0.1 MB per distinct observation row -- and its backward graphs: a moving
    # curriculum that touches many (rows, distinct rows) buckets at 16 agents would otherwise pin tens of GB; advisor, round 4).
    # `graph_captures` counts captures: a thrashing cache shows there.

    def __init__(self, learner):
        self.lr = learner
        self.dev = learner.device
        self.flat = FlatParams(learner.model)
        self.packed_tar_recur = PackedRecurrence()
        self.packed_tar_enc = PackedEncoder()
        self.packed_on_recur = PackedRecurrence()
        self.packed_on_enc = PackedEncoder()
        self._tar_head = None
        self._slot = None       # static batch buffers (GlobalBuffer.sample_batch(out=...)), graph mode
        self._splan = None      # static plan buffers for that batch
        self._iface = {}        # static interface tensors between the stage graphs
        self._graphs = {}       # (stage, key) -> [graph, ctx, last use]
        self._pools = {}
        self._capturing = False
        self._cap_stream = None
        self._tar_w_ih = None
        # (allocated here, outside any capture: the pack launches of a captured stage write them in place)
        self._on_w_ihp = torch.empty(INPROJ_PACKED_ELEMS, dtype=torch.bfloat16, device=self.dev)
        self._on_w_iht = torch.empty(LATGRAD_PACKED_ELEMS, dtype=torch.bfloat16, device=self.dev)
        self._tick = 0
        self.graph_replays = self.graph_captures = 0

    # ------------------------------------------------------------------ batch views
    def _views(self, batch):
        """Typed, flat views of a sample_batch 11-tuple (no copies for batches of the device replay)."""
        obs, action, reward, done, steps, bt, hidden, comm = batch[:8]
        B, T, N = obs.shape[:3]
        if obs.dtype != torch.bfloat16 or not _inner_contiguous(obs, 4):
            obs = obs.to(torch.bfloat16).contiguous()
        cm = comm.view(torch.uint8) if comm.dtype == torch.bool else comm.to(torch.uint8)
        if not _inner_contiguous(cm, 2):
            cm = cm.contiguous()
        hid = hidden
        if hid.dtype not in (torch.float16, torch.bfloat16):
            hid = hid.to(torch.bfloat16)
        hid = hid.contiguous()
        f32 = lambda t: t.reshape(-1).to(torch.float32).contiguous()
        return dict(B=B, T=T, N=N, obs=obs, comm=cm, hidden=hid, action=action.reshape(-1).to(torch.int64).contiguous(), reward=f32(reward),
                    done=f32(done), steps=f32(steps), bt=bt.reshape(-1).to(torch.int64).contiguous(), weights=f32(batch[9]))

    # ------------------------------------------------------------------ graph mode: static buffers
    def graph_mode(self):
        """Graph replay applies to the batches the learner samples itself from a device replay of narrow rows.  With several ranks too
        (round 5): the backward graph then ends where the gradients are final, the exchange and the optimizer step follow it as
        ordinary stream work (_run_graphed)."""
        lr = self.lr
        return bool(self.GRAPH and lr.buffer is not None and lr.prefetch and lr.buffer.max_agents <= self.GRAPH_MAX_AGENTS and lr.grad_hook is None)

    def batch_slot(self):
        """The static buffers every prioritized sample of this learner is written into (graph mode), or None."""
        if not self.graph_mode():
            return None
        lr = self.lr
        key = (lr.batch_size, lr.buffer.max_agents)
        if self._slot is None or self._slot["key"] != key:
            self._slot = lr.buffer.make_batch_slot(lr.batch_size)
            self._slot["key"] = key
            self._splan = None
            self._drop_graphs()
        return self._slot

    def _is_slot(self, batch):
        return self._slot is not None and torch.is_tensor(batch[0]) and batch[0].data_ptr() == self._slot["obs"].data_ptr()

    def _static(self, name, shape, dtype, pinned=False):
        key = (name, tuple(shape), dtype)
        t = self._iface.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, pin_memory=True) if pinned else torch.empty(shape, dtype=dtype, device=self.dev)
            self._iface[key] = t
        return t

    def _out(self, c, name, shape, dtype):
        """An interface tensor of the update: static in graph mode (the stage graphs of different buckets meet in it), fresh otherwise."""
        return self._static(name, shape, dtype) if c.static else torch.empty(shape, dtype=dtype, device=self.dev)

    def _drop_graphs(self):
        self._graphs.clear()

    def _prealloc(self, B, To, Tt):
        """The interface tensors between the stage graphs, allocated outside any capture."""
        bf = torch.bfloat16
        for name, shape, dt in (("a0", (To, B, 256), bf), ("a0_tg", (Tt, B, 256), bf), ("a0_on2", (Tt, B, 256), bf), ("d_a0", (To, B, 256), bf),
                                ("outs", (3, B), torch.float32), ("prio", (B,), torch.float64), ("loss", (1,), torch.float32)):
            self._static(name, shape, dt)

    # ------------------------------------------------------------------ plan: at sample time, one update ahead
    def plan(self, batch):
        """Launches the two closure kernels and the asynchronous copy of the per-window counts to pinned host memory."""
        v = self._views(batch)
        B, T, N, dev = v["B"], v["T"], v["N"], self.dev
        st = _stream(dev)
        static = self._is_slot(batch)
        from .model import Network

        flags = (bool(Network.PRUNE_UNREACHABLE), bool(self.DEDUP))
        if static:
            if self._splan is None or self._splan["shape"] != (T, B, N):
                self._splan = dict(shape=(T, B, N), po=WindowPlan(T - FORWARD_STEPS, B, N, dev), pt=WindowPlan(T, B, N, dev),
                                   counts=torch.empty((6, B), dtype=torch.int32, device=dev), dup=torch.empty((T, B, N), dtype=torch.uint8, device=dev),
                                   host=torch.empty((6, B), dtype=torch.int32, pin_memory=True), totals=torch.zeros(8, dtype=torch.int32, device=dev))
            sp = self._splan
            po, pt, counts, host, totals = sp["po"], sp["pt"], sp["counts"], sp["host"], sp["totals"]
        else:
            po, pt = WindowPlan(T - FORWARD_STEPS, B, N, dev), WindowPlan(T, B, N, dev)
            counts = torch.empty((6, B), dtype=torch.int32, device=dev)  # cnt online, nag online, cnt target, nag target, distinct online, distinct target
            host = torch.empty((6, B), dtype=torch.int32, pin_memory=True)
            totals = None
        cm = v["comm"]
        mark_all = 0 if flags[0] else 1  # 1: encode every observation up to the window's last step, like the reference
        for k, (p, extra) in enumerate(((po, None), (pt, v["steps"]))):
            check(lib.mapf_plan_mark(_ptr(cm), cm.stride(0), cm.stride(1), _ptr(v["bt"]), _ptr(extra), p.T, B, N, mark_all, None, _ptr(p.slot), _ptr(p.order),
                                     _ptr(p.nact), _ptr(counts[2 * k]), _ptr(counts[2 * k + 1]), _ptr(counts[4 + k]), st), "mapf_plan_mark")
            p.cnt, p.nag, p.ucnt = counts[2 * k], counts[2 * k + 1], counts[4 + k]
        dup = None
        if flags[1]:
            # which entries repeat the observation of the same agent one step earlier (exact reuse: one encoder pass per run)
            obs = v["obs"]
            dup = self._splan["dup"] if static else torch.empty((T, B, N), dtype=torch.uint8, device=dev)
            check(lib.mapf_obs_dup(T, po.T, B, N, _ptr(obs), obs.stride(0), obs.stride(1), _ptr(po.slot), _ptr(pt.slot), _ptr(po.nact), _ptr(pt.nact),
                                   _ptr(dup), _ptr(po.ucnt), _ptr(pt.ucnt), st), "mapf_obs_dup")
        po.dup = pt.dup = dup
        if totals is not None:
            # the batch totals stay on the device too: the bucket-sized launches of a graph-replayed update read their true row counts
            # from here (include/mapf_dqn.h: the `_bounded` encoder entry points) -- [2 k] entries, [4 + k] distinct observations of set k
            check(lib.mapf_plan_totals(_ptr(counts), 6, B, _ptr(totals), st), "mapf_plan_totals")
        host.copy_(counts, non_blocking=True)
        ev = None
        if not self._capturing:  # (inside a capture the event is recorded behind the graph's replay, _run_graphed)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        return dict(views=v, online=po, target=pt, counts=counts, host=host, event=ev, static=static, flags=flags, totals=totals)

    def _plan_sizes(self, pl, padded=False):
        """Host side of the plan: row totals and the compact width of both window sets (waits for the count copy -- long finished
        when the batch was planned during the update before).  padded (graph mode): the totals rounded up to their buckets."""
        pl["event"].synchronize()
        h = pl["host"].numpy()
        for k, p in enumerate((pl["online"], pl["target"])):
            p.rows = int(h[2 * k].sum())
            p.nc = self._compact_width(int(h[2 * k + 1].max()), p.B)
            p.urows = int(h[4 + k].sum()) if p.dup is not None else p.rows
            p.true_rows, p.true_urows = p.rows, p.urows
            p.valid_rows = p.valid_entries = None  # device-side counts: rows the encoder kernels compute / entries (bucket-sized launches only)
            if padded:
                p.nc = self._compact_width(p.N, p.B)
                p.rows = _round_up(p.rows, self.GRAPH_ROW_STEP)
                p.urows = _round_up(p.urows, self.GRAPH_UROW_STEP) if p.dup is not None else p.rows
                if self.BOUNDED_ROWS and pl.get("totals") is not None:
                    i = (4 + k) if p.dup is not None else 2 * k
                    p.valid_rows = pl["totals"][i:i + 1]
                    p.valid_entries = pl["totals"][2 * k:2 * k + 1]  # (entries of the window set: the rows of the recurrence's tensors)
        return pl

    def _compact_width(self, agents, B):
        """Agent positions per window in the compact numbering: a multiple of 16 (the recurrence kernels' agent tile) -- or, for windows
        of <= 4 / <= 8 agents that matter, 4 / 8: four / two consecutive windows then share a tile (`TILE_WINDOWS`; include/mapf_dqn.h:
        mapf_plan_rows, mapf_recurrent_*_packed), and a batch of 192 six-agent windows is 96 workgroups per recurrence launch instead
        of 192 -- the online and the target network's launches, which met on 256 CUs one behind the other, run side by side."""
        if self.TILE_WINDOWS:
            for w in (4, 8):
                if agents <= w and B % (16 // w) == 0:
                    return w
        return 16 * max(1, -(-agents // 16))

    @staticmethod
    def _tile(p):
        """(environments, agent rows per environment, agent-0 stride) the recurrence kernels are launched with for window set p."""
        if p.nc < 16:
            return p.B // (16 // p.nc), 16, p.nc
        return p.B, p.nc, 0

    def _plan_rows(self, p, v, padded=False):
        """`mapf_plan_rows` for one window set.  padded: the launch sizes are buckets >= the real counts; the index tables are
        initialised so that every padding row is harmless -- its observation is a copy of observation 0 of the batch (row_src = 0: the
        encoder sees finite values), its entry uses distinct row 0 (umap = 0) and is skipped by the gradient sum (row_tbp = -1)."""
        dev = self.dev
        st = _stream(dev)
        T, B, N, Nc = p.T, p.B, p.N, p.nc
        p.gidx = torch.empty((T, B, Nc), dtype=torch.int32, device=dev)
        p.comm_c = torch.empty((T, B, Nc, Nc) if Nc >= 16 else (T, B // (16 // Nc), 16, 16), dtype=torch.uint8, device=dev)
        p.h0_c = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device=dev)
        # the observations to encode: every row, or -- with the duplicate flags -- the distinct ones (umap: entry -> distinct row)
        p.obs_rows = rows_buffer((), max(p.urows, 1), (6, 9, 9), torch.bfloat16, dev)
        row_src = rows_buffer((), max(p.urows, 1), (), torch.int64, dev)
        p.umap = rows_buffer((), max(p.rows, 1), (), torch.int32, dev) if p.dup is not None else None
        p.row_tbp = rows_buffer((), max(p.rows, 1), (), torch.int32, dev) if p.dup is not None else None
        cm, obs, hid = v["comm"], v["obs"], v["hidden"]
        if padded and not self.FOLD_FILLS:  # (rounds 4-5, kept for A/B runs: three element-wise fill launches in front of the call)
            row_src.fill_(0)
            if p.umap is not None:
                p.umap.fill_(0)
                p.row_tbp.fill_(-1)
        elif padded:
            # the index tables' padding is initialised INSIDE the call (three element-wise fill launches per window set until round 6; never
            # zero_() / torch.zeros: those are hipMemsetAsync calls, i.e. memset NODES of the captured graph -- and on this runtime a graph
            # holding a (small) memset node faulted at a later replay once other memsets had been issued in between; see
            # mapf_obs_changed in csrc/mapf_actor.hip.  Nothing that is captured calls hipMemsetAsync.)
            check(lib.mapf_plan_rows_padded(T, B, N, Nc, _ptr(p.order), _ptr(p.nact), _ptr(p.cnt), _ptr(p.nag), _ptr(cm), cm.stride(0), cm.stride(1),
                                            _ptr(hid), int(hid.dtype == torch.bfloat16), _ptr(obs), obs.stride(0), obs.stride(1), _ptr(p.gidx),
                                            _ptr(p.comm_c), _ptr(p.h0_c), p.urows, _ptr(row_src), _ptr(p.obs_rows), _ptr(p.dup), _ptr(p.ucnt),
                                            _ptr(p.umap), _ptr(p.row_tbp), max(p.rows, 1) if p.umap is not None else 0, max(p.urows, 1), st),
                  "mapf_plan_rows_padded")
            return
        check(lib.mapf_plan_rows(T, B, N, Nc, _ptr(p.order), _ptr(p.nact), _ptr(p.cnt), _ptr(p.nag), _ptr(cm), cm.stride(0), cm.stride(1),
                                 _ptr(hid), int(hid.dtype == torch.bfloat16), _ptr(obs), obs.stride(0), obs.stride(1), _ptr(p.gidx),
                                 _ptr(p.comm_c), _ptr(p.h0_c), p.urows, _ptr(row_src), _ptr(p.obs_rows), _ptr(p.dup), _ptr(p.ucnt), _ptr(p.umap),
                                 _ptr(p.row_tbp), st), "mapf_plan_rows")

    def _finish_plan(self, pl):
        """Exact sizes + `mapf_plan_rows` for both window sets (the eager path; tools and bench.py read `urows` / `rows` off it)."""
        self._plan_sizes(pl)
        for p in (pl["online"], pl["target"]):
            self._plan_rows(p, pl["views"])
        return pl

    # ------------------------------------------------------------------ pieces
    def _w_ihp(self, net, own):
        """recurrent.weight_ih as the bf16 fragment image of mapf_input_proj_rows (csrc/mapf_inproj.hip).  Online network: packed by
        _pack_online with every update (its weights just changed); target network: kept in ONE buffer, re-packed in place when the
        target's weights changed (captured graphs hold its address)."""
        if own:
            return self._on_w_ihp
        w = net.recurrent.weight_ih
        key = (id(net), net.weights_epoch, w.data_ptr(), w._version)
        if self._tar_w_ih is None or self._tar_w_ih[0] != key:
            buf = self._tar_w_ih[1] if self._tar_w_ih is not None else torch.empty(INPROJ_PACKED_ELEMS, dtype=torch.bfloat16, device=w.device)
            src = w.detach()
            src = src if (src.dtype == torch.float32 and src.is_contiguous()) else src.float().contiguous()
            check(lib.mapf_input_proj_pack(_ptr(src), _ptr(buf), _stream(w.device)), "mapf_input_proj_pack")
            self._tar_w_ih = (key, buf)
        return self._tar_w_ih[1]

    def _infer_a0(self, net, images, p, own, a0):
        """agent-0 states bf16 [T, B, 256] of `net` on the window set `p` into `a0`, no gradients (target network; double-DQN's arg-max).
        images: the network's packed (encoder weights, encoder bias, recurrence weights, recurrence bias)."""
        dev, T, B, Nc = self.dev, p.T, p.B, p.nc
        st = _stream(dev)
        wp, bp, w, b = images
        lat = rows_buffer((), p.urows, (784,), torch.bfloat16, dev)
        if getattr(p, "valid_rows", None) is not None:
            check(lib.mapf_encoder_forward_bounded(_ptr(p.obs_rows), 1, p.urows, _ptr(p.valid_rows), _ptr(wp), _ptr(bp), _ptr(lat), st), "mapf_encoder_forward_bounded")
        else:
            check(lib.mapf_encoder_forward(_ptr(p.obs_rows), 1, p.urows, _ptr(wp), _ptr(bp), _ptr(lat), st), "mapf_encoder_forward")
        gi = self._expand(input_proj_rows(lat, self._w_ihp(net, own), out=rows_buffer((), p.urows, (768,), torch.bfloat16, dev)), p)  # [rows, 768]
        compact = Nc <= RECUR_NARROW_AGENTS  # the <= 48-agent kernels read / write the rows that exist (gidx); the wide ones are dense
        if not compact:
            gi_rows, gi = gi, torch.empty((T, B, Nc, 768), dtype=torch.bfloat16, device=dev)
            check(lib.mapf_rows_scatter(_ptr(gi_rows), _ptr(p.gidx), _ptr(gi), T * B * Nc, 1536, 1, st), "mapf_rows_scatter")
        h_out = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device=dev)
        E_, N_, stride = self._tile(p)
        check(lib.mapf_recurrent_infer_packed(_ptr(gi), _ptr(p.h0_c), _ptr(p.comm_c), _ptr(w), _ptr(b), T, E_, N_, _ptr(h_out), _ptr(a0),
                                              _ptr(p.gidx) if compact else None, p.rows if compact else 0, stride, st), "mapf_recurrent_infer_packed")
        return a0

    def _expand(self, x_u, p):
        """Rows of the distinct observations [urows, w] -> one row per entry [rows, w] (umap); the identity without duplicate flags."""
        if p.umap is None:
            return x_u
        out = rows_buffer((), p.rows, (x_u.shape[1],), x_u.dtype, x_u.device)
        check(lib.mapf_rows_scatter(_ptr(x_u), _ptr(p.umap), _ptr(out), p.rows, x_u.shape[1] * x_u.element_size(), 1, _stream(self.dev)),
              "mapf_rows_scatter")
        return out

    # graph mode: the encoder kernels of a bucket-sized launch compute the TRUE number of distinct observations only (read from device
    # memory), not the bucket (the variable: A/B runs)
    BOUNDED_ROWS = os.environ.get("MAPF_BOUNDED_ROWS", "1") != "0"
    # graph mode: the padding of the index tables is initialised inside mapf_plan_rows_padded and only the padding rows of the GEMM operands
    # are cleared (mapf_zero_rows_from), instead of fill launches over whole buffers (the variable: A/B runs)
    FOLD_FILLS = os.environ.get("MAPF_FOLD_FILLS", "1") != "0"
    DEDUP = True  # encode the distinct observations of a batch only (mapf_obs_dup: same agent, consecutive steps, same 486 values)

    # ------------------------------------------------------------------ the update
    def usable(self, batch):
        """The kernels' shape limits (include/mapf_dqn.h); anything else takes Learner's autograd path."""
        from .model import Network

        obs = batch[0]
        return (obs.is_cuda and obs.dim() == 6 and obs.shape[2] <= 128 and FORWARD_STEPS < obs.shape[1] <= 20 and Network.FUSED_TRAINING and
                Network.FUSED_INFERENCE and Network.FUSED_BPTT and Network.FUSED_RECURRENCE and Network.FAST_RECURRENCE)

    def why_not(self, batch):
        """Why `usable(batch)` is False, in words (Learner._say_fallback)."""
        from .model import Network

        obs = batch[0]
        if not obs.is_cuda:
            return "the batch is not on a HIP device"
        if obs.dim() != 6:
            return "observations are not [B, T, A, 6, 9, 9]"
        if obs.shape[2] > 128:
            return "%d agents per window: the recurrence kernels hold at most 128" % obs.shape[2]
        if not FORWARD_STEPS < obs.shape[1] <= 20:
            return "windows of %d steps: the kernels take %d..20" % (obs.shape[1], FORWARD_STEPS + 1)
        off = [k for k in ("FUSED_TRAINING", "FUSED_INFERENCE", "FUSED_BPTT", "FUSED_RECURRENCE", "FAST_RECURRENCE") if not getattr(Network, k)]
        return "Network.%s switched off" % ", ".join(off) if off else "usable"

    # ---- stages (each is a fixed launch sequence for given sizes: issued one after the other, or captured once per bucket) ----
    def _target_images(self):
        """Packed weight images of the target network (re-packed when its parameters changed; in place: captured graphs hold them)."""
        tar = self.lr.tar_model
        return self.packed_tar_enc.get(tar.obs_encoder, tar.weights_epoch) + self.packed_tar_recur.get(tar, inplace=True)

    def _target_forward(self, c, pt, online_images):
        """Target network (and double-DQN's online arg-max) on the target window -> c.a0_tg (c.a0_on2)."""
        lr = self.lr
        c.a0_tg = self._infer_a0(lr.tar_model, self._target_images(), pt, False, self._out(c, "a0_tg", (pt.T, pt.B, 256), torch.bfloat16))
        c.a0_on2 = None
        if lr.double_q:
            c.a0_on2 = self._infer_a0(lr.model, online_images, pt, True, self._out(c, "a0_on2", (pt.T, pt.B, 256), torch.bfloat16))

    def _pack_online(self, c):
        """Weight images of the online network (its parameters changed with the last optimizer step)."""
        model, dev = self.lr.model, self.dev
        c.wp, c.bp = self.packed_on_enc.get(model.obs_encoder, model.weights_epoch, force=self._capturing)
        c.w_rec, c.b_rec = self.packed_on_recur.get(model, force=self._capturing)
        c.wt = torch.empty(RECUR_WEIGHT_ELEMS, dtype=torch.bfloat16, device=dev)  # the backward kernel's transposed image
        check(lib.mapf_recurrent_pack(_ptr_array([p.detach() for p in recurrence_params(model)]), None, None, _ptr(c.wt), _stream(dev)), "mapf_recurrent_pack")
        c.wpt = pack_encoder_backward(model.obs_encoder)
        # the input projection's weight as fragments of W_ih (forward) and of W_ih^T (gradient w.r.t. the latents), from the flat fp32 copy
        w_ih32 = self.flat.mem(self.flat.params, "recurrent.weight_ih")
        check(lib.mapf_input_proj_pack(_ptr(w_ih32), _ptr(self._on_w_ihp), _stream(dev)), "mapf_input_proj_pack")
        check(lib.mapf_latent_grad_pack(_ptr(w_ih32), _ptr(self._on_w_iht), _stream(dev)), "mapf_latent_grad_pack")

    def _online_forward(self, c, po):
        """Online network forward on the online window, saving what the backward needs -> c.a0 and the saved tensors.
        c.padded: every tensor that enters a weight-gradient GEMM is zeroed first -- the kernels only write the rows that exist."""
        dev = self.dev
        st = _stream(dev)
        B, To, Nc = po.B, po.T, po.nc
        M, Mu = po.rows, po.urows  # entries of the window set / distinct observations among them
        bf = torch.bfloat16
        c.acts = rows_buffer((7,), Mu, (7, 7, 128), ENC_ELEMENT, dev)
        c.lat = rows_buffer((), Mu, (784,), bf, dev)
        c.bits = rows_buffer((7,), Mu, (49, 4), torch.int32, dev)
        c.valid_rows = getattr(po, "valid_rows", None)
        if c.valid_rows is not None:
            check(lib.mapf_encoder_forward_save_bounded(_ptr(po.obs_rows), 1, Mu, _ptr(c.valid_rows), _ptr(c.wp), _ptr(c.bp), _ptr(c.lat), _ptr(c.acts),
                                                        _ptr(c.bits), st), "mapf_encoder_forward_save_bounded")
        else:
            check(lib.mapf_encoder_forward_save(_ptr(po.obs_rows), 1, Mu, _ptr(c.wp), _ptr(c.bp), _ptr(c.lat), _ptr(c.acts), _ptr(c.bits), st),
                  "mapf_encoder_forward_save")
        gi_rows = self._expand(input_proj_rows(c.lat, self._on_w_ihp, out=rows_buffer((), Mu, (768,), bf, dev)), po)
        c.compact = compact = Nc <= RECUR_NARROW_AGENTS
        # rows of the recurrence's saved tensors / gradient outputs: the M rows that exist (compact: the <= 48-agent kernels address
        # them through gidx) or all To x B x Nc (step, window, position) entries (the wide kernels)
        # (compact: padded to a multiple of the weight-gradient GEMMs' split size, the padding rows zeroed below)
        c.R = R = -(-M // WGRAD_SPLIT) * WGRAD_SPLIT if compact else To * B * Nc
        c.ridx, c.nrows = (_ptr(po.gidx), R) if compact else (None, 0)
        if compact:
            gi = gi_rows
        else:
            gi = torch.empty((To, B, Nc, 768), dtype=bf, device=dev)
            check(lib.mapf_rows_scatter(_ptr(gi_rows), _ptr(po.gidx), _ptr(gi), To * B * Nc, 1536, 1, st), "mapf_rows_scatter")
        if c.padded:
            # one allocation for everything the tall GEMMs read, one memset: [saves 0, 2, 4, 5 | outputs of the backward 0..5]
            widths = [256, 2 * 256, 2 * 128, 2 * 64, 768, 768, 2 * 768, 2 * 768, 2 * 64, 2 * 384]
            ve = getattr(po, "valid_entries", None) if self.FOLD_FILLS else None
            zero = torch.empty(R * sum(widths), dtype=bf, device=dev)
            if ve is None:
                zero.fill_(0)  # (a fill kernel, not a memset node: see _plan_rows)
            parts, off = [], 0
            for w in widths:
                parts.append(zero[off:off + R * w])
                off += R * w
            s0, s2, s4, s5 = parts[0].view(R, 256), parts[1].view(2, R, 256), parts[2].view(2, R, 128), parts[3].view(2, R, 64)
            c.outs_b = [parts[4].view(R, 768), parts[5].view(R, 768), parts[6].view(2, R, 768), parts[7].view(2, R, 768), parts[8].view(2, R, 64),
                        parts[9].view(2, R, 384), torch.empty((B, 2432), dtype=torch.float32, device=dev)]
            if ve is not None:
                # only the PADDING rows -- behind this update's true entry count, read from device memory -- are cleared: the kernels write
                # every row that exists (round 5 filled all 82 MB of these operands per update, 16 us on the main chain)
                ob = c.outs_b
                ops = [s0, s2[0], s2[1], s4[0], s4[1], s5[0], s5[1], ob[0], ob[1], ob[2][0], ob[2][1], ob[3][0], ob[3][1], ob[4][0], ob[4][1],
                       ob[5][0], ob[5][1]]
                rb = (ctypes.c_int * len(ops))(*[t.shape[-1] * 2 for t in ops])
                check(lib.mapf_zero_rows_from(_ptr_array(ops), rb, len(ops), _ptr(ve), R, st), "mapf_zero_rows_from")
        else:
            s0, s2, s4, s5 = (rows_buffer((), R, (256,), bf, dev), rows_buffer((2,), R, (256,), bf, dev), rows_buffer((2,), R, (128,), bf, dev),
                              rows_buffer((2,), R, (64,), bf, dev))
            c.outs_b = None
        c.saves = [s0, rows_buffer((), R, (1024,), bf, dev), s2, rows_buffer((2,), R, (384,), bf, dev), s4, s5, rows_buffer((2,), R, (1024,), bf, dev),
                   torch.empty((2, To * B, 2, 48, 64) if Nc <= RECUR_NARROW_AGENTS else (8,), dtype=bf, device=dev)]
        h_out = torch.empty((B, Nc, 256), dtype=bf, device=dev)
        c.a0 = self._out(c, "a0", (To, B, 256), bf)
        c.sp = _ptr_array(c.saves)
        E_, N_, stride = self._tile(po)
        check(lib.mapf_recurrent_forward_save_packed(_ptr(gi), _ptr(po.h0_c), _ptr(po.comm_c), _ptr(c.w_rec), _ptr(c.b_rec), To, E_, N_, _ptr(h_out),
                                                     _ptr(c.a0), c.sp, c.ridx, c.nrows, stride, st), "mapf_recurrent_forward_save_packed")

    def _head(self, c, v, To, Tt, batch):
        """Dueling heads, TD error, priorities, loss and their gradients; then the priority write-back."""
        lr, dev, flat = self.lr, self.dev, self.flat
        tar, G, B = lr.tar_model, flat.grads, v["B"]
        flat.grads.fill_(0)  # (a fill kernel, not a memset node: see _plan_rows)
        c.outs = self._out(c, "outs", (3, B), torch.float32)  # q, q_next, td
        c.prio = self._out(c, "prio", (B,), torch.float64)
        c.loss = self._out(c, "loss", (1,), torch.float32)
        scratch = torch.empty(9 * B, dtype=torch.float32, device=dev)
        c.d_a0 = self._out(c, "d_a0", (To, B, 256), torch.bfloat16)
        head_names = ("adv.weight", "adv.bias", "state.weight", "state.bias")
        head_on = _ptr_array([flat.mem(flat.params, k) for k in head_names])
        if self._tar_head is None or self._tar_head[0] is not tar:
            tnamed = dict(tar.named_parameters())
            self._tar_head = (tar, [tnamed[k] for k in head_names])
        head_tg_t = [t.detach().to(torch.float32).contiguous() for t in self._tar_head[1]]
        head_g = _ptr_array([flat.mem(G, k) for k in head_names])
        check(lib.mapf_dqn_head_loss(B, To, Tt, _ptr(c.a0), _ptr(c.a0_tg), _ptr(c.a0_on2), _ptr(v["bt"]), _ptr(v["steps"]), _ptr(v["action"]),
                                     _ptr(v["reward"]), _ptr(v["done"]), _ptr(v["weights"]), head_on, _ptr_array(head_tg_t), GAMMA, _ptr(c.outs[0]),
                                     _ptr(c.outs[1]), _ptr(c.outs[2]), _ptr(c.prio), _ptr(c.loss), _ptr(scratch), _ptr(c.d_a0), head_g, _stream(dev)),
              "mapf_dqn_head_loss")

    def _write_priorities(self, c, batch):
        lr = self.lr
        idxes, old_ptr = batch[8], batch[10]
        if lr.buffer is not None and idxes is not None:
            lr.buffer.update_priorities(idxes, c.prio, old_ptr)                                   # worker.py:331 (values known here)

    def _backward(self, c, po, lr_value):
        """Backward through time, weight gradients, encoder backward, the only collective, clip + Adam.  Returns the gradient norm."""
        lr, dev, flat = self.lr, self.dev, self.flat
        st, bf, G = _stream(dev), torch.bfloat16, flat.grads
        B, To, Nc, R, M, Mu, compact = po.B, po.T, po.nc, c.R, po.rows, po.urows, c.compact
        saves = c.saves
        outs_b = c.outs_b
        if outs_b is None:
            outs_b = [rows_buffer((), R, (768,), bf, dev), rows_buffer((), R, (768,), bf, dev),
                      rows_buffer((2,), R, (768,), bf, dev), rows_buffer((2,), R, (768,), bf, dev),
                      rows_buffer((2,), R, (64,), bf, dev), rows_buffer((2,), R, (384,), bf, dev),
                      torch.empty((B, 2432), dtype=torch.float32, device=dev)]
        E_, N_, stride = self._tile(po)  # (bsum: one row per tile, E_ of its B rows)
        check(lib.mapf_recurrent_backward_packed(c.sp, _ptr(po.comm_c), _ptr(c.d_a0), _ptr(c.wt), To, E_, N_, _ptr_array(outs_b), c.ridx, c.nrows, stride, st),
              "mapf_recurrent_backward_packed")
        if compact and R > M and not c.padded:  # rows M..R of every GEMM operand (both rounds of the [2, R, w] tensors)
            ops = [saves[0], saves[2][0], saves[2][1], saves[4][0], saves[4][1], saves[5][0], saves[5][1], outs_b[1], outs_b[2][0], outs_b[2][1],
                   outs_b[3][0], outs_b[3][1], outs_b[4][0], outs_b[4][1], outs_b[5][0], outs_b[5][1]]
            rb = (ctypes.c_int * len(ops))(*[t.shape[-1] * 2 for t in ops])
            check(lib.mapf_zero_rows(_ptr_array(ops), rb, len(ops), M, R, st), "mapf_zero_rows")
        d_gi1, d_gh1, d_gi2, d_gh2, d_info, d_qkv, bsum = outs_b
        hin0, _, hr, _, ctxs, info, _, _ = saves
        hrf = hr.view(2 * R, 256)
        rows_k = WGRAD_SPLIT if compact else 8192
        # the recurrence's weight / bias gradients (five tall GEMMs + their sums: ~0.3 ms of small launches at few agents) depend only
        # on what the BPTT kernel wrote, the encoder's backward chain below only on d_gi1: inside a capture they become two branches
        # of the graph (a second stream that forks here and joins before the optimizer step).  Everything the branch reads is held by
        # `c` / `outs_b` until the join; what it allocates, it allocates on its own stream.
        # Eagerly too (config 2: nine ~50 us launches that would otherwise sit between the BPTT kernel and the encoder's backward).
        # Several ranks: the recurrence's and the head's piece of the gradient exchange is issued FROM the branch, behind the last
        # gradient it contains (the collective is ordered behind the stream it is issued on), so it travels while the encoder's
        # backward chain runs on the main stream.  (Rounds 3-4 kept one stream with several ranks and lost the overlap.)
        from .learner import exchanging

        ex = exchanging()
        aux = self.lr._side if (self._capturing or self.lr.grad_hook is None) else None
        cur_s = torch.cuda.current_stream(dev)
        n_all, split = flat.grads.numel(), lr.bucket.split
        # (inside a capture no collective is issued: the captured stage ends where the gradients are final, see _run_graphed)
        two_pieces = ex and not self._capturing and aux is not None and split is not None and 0 < split < n_all
        if aux is not None:
            aux.wait_stream(cur_s)
        with (torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext()):
            st_x = _stream(dev)
            _tall_tn_into(flat.mem(G, "recurrent.weight_hh"), d_gh1, hin0, rows_k)
            _tall_tn_into(flat.span(G, "comm.self_attn.W_Q.weight", "comm.self_attn.W_V.weight"), d_qkv.view(2 * R, 384), hrf, rows_k)
            _tall_tn_into(flat.mem(G, "comm.self_attn.W_O.weight"), d_info.view(2 * R, 64), ctxs.view(2 * R, 128), rows_k)
            _tall_tn_into(flat.mem(G, "comm.update_cell.weight_ih"), d_gi2.view(2 * R, 768), info.view(2 * R, 64), rows_k)
            _tall_tn_into(flat.mem(G, "comm.update_cell.weight_hh"), d_gh2.view(2 * R, 768), hrf, rows_k)
            bias_names = ("recurrent.bias_ih", "recurrent.bias_hh", "comm.self_attn.W_Q.bias", "comm.self_attn.W_K.bias", "comm.self_attn.W_V.bias",
                          "comm.update_cell.bias_ih", "comm.update_cell.bias_hh")
            check(lib.mapf_recurrent_bias_grads(_ptr(bsum), E_, _ptr_array([flat.mem(G, k) for k in bias_names]), st_x), "mapf_recurrent_bias_grads")
        # ---- input projection ----
        if compact:
            d_gi_rows = d_gi1[:M]
        else:
            d_gi_rows = rows_buffer((), M, (768,), bf, dev)
            check(lib.mapf_rows_scatter(_ptr(d_gi_rows), _ptr(po.gidx), _ptr(d_gi1), R, 1536, 0, st), "mapf_rows_scatter")
        if po.umap is not None:  # gradient of a shared row = the sum over the entries that use it
            d_gi_u = rows_buffer((), Mu, (768,), bf, dev)
            if c.padded:
                d_gi_u.fill_(0)  # (distinct rows behind the real ones have no entry: their gradient must read zero)
            check(lib.mapf_dedup_sum(To, B, Nc, M, 1536, _ptr(po.gidx), _ptr(po.umap), _ptr(po.row_tbp), _ptr(d_gi_rows), _ptr(d_gi_u), st),
                  "mapf_dedup_sum")
            d_gi_rows = d_gi_u
        if aux is not None:  # (the input projection's weight gradient joins the side branch: only the encoder's chain needs g_lat)
            aux.wait_stream(cur_s)
            with torch.cuda.stream(aux):
                _tall_tn_into(flat.mem(G, "recurrent.weight_ih"), d_gi_rows, c.lat, rows=4096)
                # the recurrence's and the head's gradients are final on this branch: their piece of the exchange (12 % of the bytes)
                if two_pieces:
                    lr.bucket.begin(split, n_all)
        g_lat = latent_grad_rows(d_gi_rows, self._on_w_iht)
        if aux is None:
            _tall_tn_into(flat.mem(G, "recurrent.weight_ih"), d_gi_rows, c.lat, rows=4096)
        # ---- encoder: backward-data chain in one kernel, then the weight-gradient kernels ----
        self._encoder_backward(po.obs_rows, Mu, c.acts, c.lat, c.bits, g_lat, c.wpt, aux, getattr(c, "valid_rows", None))
        if aux is not None:
            cur_s.wait_stream(aux)
        if ex and self._capturing:
            return None  # several ranks, captured: the exchange and the optimizer step follow the graph (_run_graphed)
        # ---- the exchange (its second piece: the encoder's gradients), clip, Adam ----
        if two_pieces:
            lr.bucket.begin(0, split)
            lr.bucket.finish()
        else:
            lr.bucket.all_reduce_mean()
        if lr.grad_hook is not None:
            lr.grad_hook(lr)
        return flat.adam_step(lr_value)

    def run(self, batch, pl=None, own_batch=False):
        lr, dev, flat = self.lr, self.dev, self.flat
        flat.sync()
        if pl is None:
            pl = self.plan(batch)
        if own_batch and pl.get("static") and self.graph_mode() and lr.prefetch and lr._side is not None:
            return self._run_graphed(batch, pl)
        pl = self._finish_plan(pl)
        v, po, pt = pl["views"], pl["online"], pl["target"]
        B, To, Tt = v["B"], po.T, pt.T
        cur = torch.cuda.current_stream(dev)
        c = _Ctx()
        c.static = c.padded = False
        # ---- weight images of the online network, on THIS stream before the side stream may read them (double-DQN) ----
        self._pack_online(c)
        # ---- target network (and double-DQN's online arg-max) on the second stream ----
        # (only while the batches are small: see SIDE_STREAM_MAX_ROWS)
        side = lr._side if max(po.rows, pt.rows) <= SIDE_STREAM_MAX_ROWS else None
        if side is not None:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._target_forward(c, pt, (c.wp, c.bp, c.w_rec, c.b_rec))
                ready = torch.cuda.Event()
                ready.record(side)
        else:
            self._target_forward(c, pt, (c.wp, c.bp, c.w_rec, c.b_rec))
        # ---- online network forward, saving what the backward needs ----
        self._online_forward(c, po)
        # ---- dueling heads, TD error, priorities, loss and their gradients ----
        if side is not None:
            cur.wait_event(ready)
            for t in (c.a0_tg, c.a0_on2):  # allocated on the side stream, consumed on this one
                if t is not None:
                    t.record_stream(cur)
            for t in (pt.obs_rows, pt.gidx, pt.comm_c, pt.h0_c, pt.umap):  # allocated on this stream, read on the side stream
                if t is None:
                    continue
                t.record_stream(side)
        self._head(c, v, To, Tt, batch)
        pre_ready = None
        if lr.prefetch and own_batch and lr._side is not None:
            # the priority write-back (a 40 us sum-tree walk) and, with its priorities in, the next batch's sample and plan go to the second
            # stream: ~0.5 ms of small kernels that fit beside the backward-through-time kernel (192 workgroups on 256 CUs) instead of
            # in front of it
            lr._side.wait_stream(cur)
            with torch.cuda.stream(lr._side):
                if lr.replay_gate is not None:  # (actors on their own stream: their last episode flush precedes this update's replay operations)
                    lr._side.wait_event(lr.replay_gate)
                self._write_priorities(c, batch)
                lr._launch_prefetch()
                pre_ready = torch.cuda.Event()
                pre_ready.record(lr._side)
        else:
            if lr.replay_gate is not None:
                cur.wait_event(lr.replay_gate)
            self._write_priorities(c, batch)
            if lr.prefetch and own_batch:
                lr._launch_prefetch()
        # this update's replay operations (priority write-back, next sample) are enqueued: whoever else writes the replay waits for this
        lr.replay_released = pre_ready
        if pre_ready is None:
            lr.replay_released = torch.cuda.Event()
            lr.replay_released.record(cur)
        grad_norm = self._backward(c, po, lr.current_lr())
        if pre_ready is not None:
            # everything the caller enqueues from here on (the next update; an actor step that appends to the replay) is ordered behind
            # the sample.  Memory: the prefetched tensors come from the second stream's pool and are consumed on this one -- safe without
            # record_stream because every use of the second stream starts with wait_stream(this one)
            cur.wait_event(pre_ready)
        return dict(loss=c.loss[0], td=c.outs[2].view(B, 1), priorities=c.prio, grad_norm=grad_norm, q=c.outs[0].view(B, 1), q_next=c.outs[1].view(B, 1))

    # ------------------------------------------------------------------ graph mode
    def _capture(self, pool, fn):
        """Captures the launches of fn() into a new graph whose allocations come from the private pool `pool`.  (Not
        `torch.cuda.graph(...)`: that context manager runs gc.collect() and empties the allocator's cache around every capture --
        tens of milliseconds each, and new buckets keep appearing while the curriculum moves.)  The learner's two streams are idle
        when the capture starts; other streams (the actors') may be busy."""
        dev = self.dev
        cur = torch.cuda.current_stream(dev)
        cur.synchronize()
        if self.lr._side is not None:
            self.lr._side.synchronize()
        if self._cap_stream is None:
            from .streams import role_stream

            self._cap_stream = role_stream(dev, "capture_learner")
            if not OWN_TALL_GEMM:  # (the A/B formulation with library products: hipBLASLt must have made its handle before a capture)
                warm_up_gemm_library(self._cap_stream)
                if self.lr._side is not None:
                    warm_up_gemm_library(self.lr._side)  # (the capture forks onto it)
        from .fused import prepare_tall_ws

        prepare_tall_ws(dev, (self._cap_stream, self.lr._side))
        g = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            with no_gc_during_capture(), torch.cuda.stream(self._cap_stream):
                g.capture_begin(pool=self._pools.get(pool), capture_error_mode=capture_mode())
                try:
                    out = fn()
                finally:
                    g.capture_end()
        finally:
            self._capturing = False
        self._pools.setdefault(pool, g.pool())
        self.graph_captures += 1
        return g, out

    _update_tick = 0   # `_tick` at the start of the update in flight: entries used since are not evicted

    def _pool_bytes(self):
        """Bytes the allocator holds in blocks that are in use (captures allocate from private pools: their blocks stay 'active' for the
        graph's lifetime, whether the pool grew or reused a block an earlier capture's temporary had freed -- reserved-memory growth,
        round 5's measure, reads 0 for the latter)."""
        return int(torch.cuda.memory_stats(self.dev).get("active_bytes.all.current", 0))

    def _graph(self, stage, key, pool, fn):
        """The captured graph of `stage` for `key` (captured now if new).  Returns (graph, what fn returned at capture time)."""
        self._tick += 1
        ent = self._graphs.get((stage, key))
        if ent is None:
            same = [k for k in self._graphs if k[0] == stage and self._graphs[k][2] < self._update_tick]
            if len(same) >= self.GRAPH_CACHE:
                del self._graphs[min(same, key=lambda k: self._graphs[k][2])]
            # the byte bound: least recently used entries of any stage go until the new one fits (sizes: what the allocator's
            # reserved memory grew by during an entry's captures -- its own and, for an online entry, its backward graphs')
            # (never an entry THIS update has already fetched -- `_update_tick`: its pack / target / online graphs are in flight and its
            # backward capture is still to be accounted to the online entry; advisor, round 5)
            while sum(e[3] for e in self._graphs.values()) > self.GRAPH_CACHE_BYTES:
                old = [k for k, e in self._graphs.items() if e[2] < self._update_tick]
                if not old:
                    break
                del self._graphs[min(old, key=lambda k: self._graphs[k][2])]
            before = self._pool_bytes()
            g, out = self._capture(pool, fn)
            ent = [g, out, self._tick, max(0, self._pool_bytes() - before)]
            self._graphs[(stage, key)] = ent
        ent[2] = self._tick
        return ent[0], ent[1]

    def _run_graphed(self, batch, pl):
        """One update as five graph replays (see the class comment).  `batch` / `pl` are the static slot and its static plan."""
        lr, dev, flat = self.lr, self.dev, self.flat
        self._update_tick = self._tick + 1
        self._plan_sizes(pl, padded=True)
        v, po, pt = pl["views"], pl["online"], pl["target"]
        B, To, Tt = v["B"], po.T, pt.T
        cur, side = torch.cuda.current_stream(dev), lr._side
        flags = pl["flags"] + (bool(lr.double_q),)
        shape = (B, To, Tt, po.N)
        key_t = shape + flags + (pt.nc, pt.rows, pt.urows)
        key_o = shape + flags + (po.nc, po.rows, po.urows)
        lr_value = lr.current_lr()
        tar = lr.tar_model
        # the target network's weight images follow its parameters OUTSIDE the graphs (they change every 2,500 updates), in place, on
        # the stream that reads them
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._target_images()
            self._w_ihp(tar, False)
        cur.wait_stream(side)  # (captures below synchronise the device anyway; replays of the target graph run on `side` behind the packs)

        ROWS = ("gidx", "comm_c", "h0_c", "obs_rows", "umap", "row_tbp")

        def cap_rows(p):
            """`mapf_plan_rows` of one window set as a graph of its own: both sets' row tables are built on the main stream BEFORE
            the two networks' forward passes fork (round 5 had each at the head of its forward graph, where the online set's
            plan_rows_kernel ran beside the target network's encoder and took 110 us instead of 20:
            profiles/r05_update6_graph_timeline_tiles.md).  The forward graph of the same key is kept WITH this entry (`c.fwd`): it
            reads the tensors this capture allocated, so the two are evicted together."""
            def fn():
                c = _Ctx()
                self._plan_rows(p, v, padded=True)
                c.rows = tuple(getattr(p, k) for k in ROWS)
                # THIS update's true row count, copied out of the plan's totals: the prefetch stage overwrites those with the next
                # batch's while this update's backward stage is still to read them
                c.valid = c.valid_e = None
                if p.valid_rows is not None:
                    both = torch.cat([p.valid_rows, p.valid_entries])  # (one small launch for the two words)
                    c.valid, c.valid_e = both[0:1], both[1:2]
                c.fwd = None
                return c
            return fn

        def use_rows(p, c):
            for k, t in zip(ROWS, c.rows):
                setattr(p, k, t)
            p.valid_rows, p.valid_entries = c.valid, c.valid_e

        def cap_target():
            c = _Ctx()
            c.static, c.padded = True, True
            use_rows(pt, c_rt)
            self._target_forward(c, pt, (c_p.wp, c_p.bp, c_p.w_rec, c_p.b_rec))
            return c

        def cap_pack():
            c = _Ctx()
            self._pack_online(c)
            return c

        def cap_online():
            c = _Ctx()
            c.static, c.padded = True, True
            c.wp, c.bp, c.w_rec, c.b_rec, c.wt, c.wpt = c_p.wp, c_p.bp, c_p.w_rec, c_p.b_rec, c_p.wt, c_p.wpt
            c.backward = {}
            use_rows(po, c_ro)
            self._online_forward(c, po)
            return c

        self._prealloc(B, To, Tt)
        # the online network's weight images (six pack launches behind the last optimizer step) on the SECOND stream, beside the two
        # window sets' row tables on this one -- neither needs the other; both forward passes wait for both
        # (captures happen with THIS stream current -- _capture drains it and the second one first -- replays go where they belong)
        g_k, c_p = self._graph("pack", (), "main", cap_pack)
        g_rt, c_rt = self._graph("rows_t", key_t, "side", cap_rows(pt))
        g_ro, c_ro = self._graph("rows_o", key_o, "main", cap_rows(po))
        if c_rt.fwd is None or c_ro.fwd is None:  # (new buckets: the row tables must exist on the device before a forward is captured on them)
            g_k.replay(), g_rt.replay(), g_ro.replay()
            if c_rt.fwd is None:
                c_rt.fwd = self._capture_into(("rows_t", key_t), "side", cap_target)
            if c_ro.fwd is None:
                c_ro.fwd = self._capture_into(("rows_o", key_o), "main", cap_online)
        (g_t, c_t), (g_o, c_o) = c_rt.fwd, c_ro.fwd
        # second stream: packs, the target set's row table, the target network's forward; this stream: the online set's row table,
        # then -- behind the packs -- the online network's forward (the critical path: it starts ~60 us into the update)
        # (the host's launches reach the GPU in the order they are issued, a graph launch costs it 10-20 us, and this stream's chain is
        # the critical one: its row table, the packs it waits for, its forward -- then the target network's two graphs)
        g_ro.replay()
        with torch.cuda.stream(side):
            g_k.replay()
            packed = torch.cuda.Event()
            packed.record(side)
        cur.wait_event(packed)
        g_o.replay()
        with torch.cuda.stream(side):
            g_rt.replay()
            g_t.replay()
        cur.wait_stream(side)

        def cap_head():
            c = _Ctx()
            c.static, c.padded = True, True
            c.a0, c.a0_tg, c.a0_on2 = c_o.a0, c_t.a0_tg, c_t.a0_on2
            self._head(c, v, To, Tt, batch)
            return c

        # (its own memory pool: this graph runs BETWEEN the online graph and its backward graph, whose saved tensors live in pool "main" --
        # a temporary of this capture may not alias what a later capture keeps there across graphs)
        g_h, c_h = self._graph("head", shape + flags, "head", cap_head)  # (a0 / a0_tg / a0_on2 are static interface buffers: any bucket's)
        g_h.replay()

        def cap_prefetch():
            # the priority write-back (a 40 us sum-tree walk) leads the side stream's stage: round 5 had it in the head graph, i.e. on
            # the main stream between the loss and the backward-through-time kernel
            self._write_priorities(c_h, batch)
            lr._launch_prefetch()
            return lr._pre

        from .model import Network

        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if lr.replay_gate is not None:  # (actors on their own stream: their last episode flush precedes this update's replay operations)
                side.wait_event(lr.replay_gate)
            # (keyed by what the NEXT batch's plan bakes in: the switches as they are now)
            g_p, pre = self._graph("prefetch", shape + (bool(Network.PRUNE_UNREACHABLE), bool(self.DEDUP)), "side", cap_prefetch)
            g_p.replay()
            pre_ready = torch.cuda.Event()
            pre_ready.record(side)
            # this update's results, copied out of the static interface buffers (the caller may keep them across updates): on this stream,
            # beside the backward stage -- they were four serial copy launches at the end of the chain
            outs, prio, loss = c_h.outs.clone(), c_h.prio.clone(), c_h.loss.clone()
            out_ready = torch.cuda.Event()
            out_ready.record(side)
        pre[1]["event"] = pre_ready
        lr._pre = pre
        lr.replay_released = pre_ready

        def cap_backward():
            use_rows(po, c_ro)
            c_o.d_a0 = c_h.d_a0
            norm = self._backward(c_o, po, lr_value)
            return norm

        ent = c_o.backward.get(lr_value)  # (kept with the online graph whose saved tensors it reads: evicted together)
        if ent is None:
            saved_step, saved_epoch = flat.step_host, lr.model.weights_epoch
            g_b, norm = self._capture_into(("rows_o", key_o), "main", cap_backward)
            flat.step_host, lr.model.weights_epoch = saved_step, saved_epoch  # (a capture runs adam_step's Python without executing it)
            ent = c_o.backward[lr_value] = (g_b, norm)
        g_b, norm = ent
        g_b.replay()
        if norm is None:
            # several ranks: the captured stage ended where this rank's gradients are final -- the one collective of an update and
            # the optimizer step (two launches, device-side step count) are ordinary stream work behind it.  RCCL's collective is
            # stream-ordered (no host wait); the second piece is not overlapped here as in the eager path: at the shapes that replay
            # graphs an update is launch-latency-bound and the 8.2 MB exchange a fraction of it.
            lr.bucket.all_reduce_mean()
            norm = flat.adam_step(lr_value)
        else:
            flat.step_host += 1
            lr.model.weights_epoch += 1
            norm = norm.clone()
        cur.wait_event(out_ready)  # (behind pre_ready on the same stream: the next batch is sampled and planned, the results are copied)
        self.graph_replays += 1
        return dict(loss=loss[0], td=outs[2].view(B, 1), priorities=prio, grad_norm=norm, q=outs[0].view(B, 1), q_next=outs[1].view(B, 1))

    def _capture_into(self, entry_key, pool, fn):
        """A capture whose graph is kept INSIDE the cache entry `entry_key` (a forward graph with its row tables, a backward graph with
        its forward): its bytes are accounted to that entry, it is evicted with it."""
        before = self._pool_bytes()
        g, out = self._capture(pool, fn)
        ent = self._graphs.get(entry_key)
        if ent is not None:
            ent[3] += max(0, self._pool_bytes() - before)
        return g, out

    WGRAD_MERGED = os.environ.get("MAPF_WGRAD_MERGED", "1") != "0"  # (the variable: A/B runs)

    def _encoder_backward(self, obs_rows, M, acts, lat, bits, g_lat, wpt, aux=None, valid=None):
        """aux: a second stream for what only needs the backward-data kernel's outputs besides the six 3x3 weight-gradient launches
        (bias sums, conv0's and the 1x1 head's weight gradients: ~0.15 ms of small launches at few agents); the caller joins it."""
        dev, flat = self.dev, self.flat
        G, st, bf = flat.grads, _stream(dev), torch.bfloat16
        nblk = -(-M // ENC_OBS_PER_BLOCK)
        gz = rows_buffer((7,), M, (7, 7, 128), ENC_ELEMENT, dev)
        gz7 = rows_buffer((), M, (49, 16), ENC_ELEMENT, dev).view(M * 49, 16)
        gb_part = rows_buffer((7,), nblk, (128,), torch.float32, dev)
        gb7_part = rows_buffer((), 4 * nblk, (16,), torch.float32, dev)
        # the chain's gradients are f16 times a power-of-two loss scale S picked from max |g_lat| (include/mapf_dqn.h);
        # scale[1] = the bits of 1 / S, which the weight-gradient kernels multiply their partial sums by
        scale = torch.empty(2, dtype=torch.int32, device=dev)
        # valid: the true observation count on the device (bucket-sized launches of a graph-replayed update; rows behind it are skipped)
        if valid is not None:
            check(lib.mapf_encoder_backward_bounded(_ptr(g_lat), _ptr(lat), M, _ptr(valid), _ptr(bits), _ptr(wpt), _ptr(gz), _ptr(gb_part), _ptr(gz7),
                                                    _ptr(gb7_part), _ptr(scale), st), "mapf_encoder_backward_bounded")
        else:
            check(lib.mapf_encoder_backward(_ptr(g_lat), _ptr(lat), M, _ptr(bits), _ptr(wpt), _ptr(gz), _ptr(gb_part), _ptr(gz7), _ptr(gb7_part),
                                            _ptr(scale), st), "mapf_encoder_backward")
        names = ["obs_encoder.0", "obs_encoder.2.block1", "obs_encoder.2.block2", "obs_encoder.3.block1", "obs_encoder.3.block2",
                 "obs_encoder.4.block1", "obs_encoder.4.block2", "obs_encoder.5"]
        if aux is not None:
            aux.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext()):
            st_x = _stream(dev)
            ws0 = torch.empty((ENC_WGRAD0_PARTS, 128, 64), dtype=torch.float32, device=dev)
            if valid is not None:
                check(lib.mapf_encoder_wgrad0_bounded(_ptr(gz[0]), _ptr(obs_rows), 1, M, _ptr(valid), _ptr(scale), _ptr(ws0), st_x), "mapf_encoder_wgrad0_bounded")
            else:
                check(lib.mapf_encoder_wgrad0(_ptr(gz[0]), _ptr(obs_rows), 1, M, _ptr(scale), _ptr(ws0), st_x), "mapf_encoder_wgrad0")
            # bias gradients from the backward kernel's per-workgroup partials, conv0's weight gradient from its slabs (columns
            # j = ci*9 + ky*3 + kx -> the weight's memory [co][ky][kx][ci]): two small launches of this library
            check(lib.mapf_encoder_small_grads(_ptr(gb_part), nblk, _ptr(flat.span(G, names[0] + ".bias", names[6] + ".bias")), _ptr(gb7_part), 4 * nblk,
                                               _ptr(flat.mem(G, names[7] + ".bias")), _ptr(ws0), ENC_WGRAD0_PARTS, _ptr(flat.mem(G, names[0] + ".weight")),
                                               _ptr(torch.empty(65536, dtype=torch.float32, device=dev)), st_x), "mapf_encoder_small_grads")
            # the 1x1 head: [16, 128] = gz7^T acts6 over M * 49 positions (f16, the chain's loss scale taken out in fp32)
            tall_tn_into(flat.mem(G, names[7] + ".weight").view(16, 128), gz7, acts[6].reshape(M * 49, 128), scale=scale)
        if (self.WGRAD_MERGED or valid is not None) and gz.stride(0) == acts.stride(0) == M * 6272:
            # the six 3x3 layers' weight gradients in ONE launch (one workgroup per CU: 6 x 21 partitions x 2 slabs) and one sum -- six
            # serial (wgrad 54-75 us, sum 12-17 us) pairs were the last 400 us of a 6-agent update's chain
            # (profiles/r05_update6_graph_timeline_tiles.md), most of each launch the write of 128 partial slabs
            # (21 partitions per layer = 252 workgroups, one round.  A resident workgroup takes its CU's whole register file, so the
            # side branch's small launches -- conv0's weight gradient, bias sums, the 1x1 head -- run once the round ends; leaving 16 / 40 /
            # 64 CUs free for them -- 20 / 18 / 16 partitions -- measured the same update time: profiles/r06_update6_wgrad_parts.txt)
            P = int(os.environ.get("MAPF_WGRAD_PARTS", str(ENC_WGRAD_PARTS // 6)))
            ws = torch.empty((6, P, 128, 3, 3, 128), dtype=torch.float32, device=dev)
            check(lib.mapf_encoder_wgrad_multi(_ptr(gz[1]), gz.stride(0), _ptr(acts[0]), acts.stride(0), 6, P, M, _ptr(valid), _ptr(scale), _ptr(ws), st),
                  "mapf_encoder_wgrad_multi")
            sum_parts_into([flat.mem(G, names[k] + ".weight") for k in range(1, 7)], [ws[k - 1] for k in range(1, 7)])
            return
        ws = torch.empty((ENC_WGRAD_PARTS, 128, 3, 3, 128), dtype=torch.float32, device=dev)
        for k in range(1, 7):
            check(lib.mapf_encoder_wgrad(_ptr(gz[k]), _ptr(acts[k - 1]), M, _ptr(scale), _ptr(ws), st), "mapf_encoder_wgrad")
            sum_parts_into([flat.mem(G, names[k] + ".weight")], [ws])  # [co][ky][kx][ci] == the weight's channels_last memory


OWN_TALL_GEMM = os.environ.get("MAPF_OWN_TALL_GEMM", "1") != "0"  # (A/B runs)


def _tall_tn_into(out, a, b, rows=8192):
    """out[m, n] = a^T b for a [K, m], b [K, n] with K in the 10^4 .. 10^6 range, written in place: this library's split-K kernel
    (fused.tall_tn_into; fp32 partial slabs summed in partition order).  MAPF_OWN_TALL_GEMM=0: rounds 1-4's formulation -- K split into
    batches of `rows` for a library bmm (16-bit in / fp32 out) whose fp32 partial products are summed into `out`."""
    if OWN_TALL_GEMM:
        return tall_tn_into(out, a, b)
    K, m = a.shape
    S = K // rows
    if S > 1:
        part = torch.bmm(a[:S * rows].view(S, rows, m).transpose(1, 2), b[:S * rows].view(S, rows, -1), out_dtype=torch.float32)
        torch.sum(part, dim=0, out=out)
        if K > S * rows:
            out += torch.mm(a[S * rows:].t(), b[S * rows:], out_dtype=torch.float32)
    else:
        out.copy_(torch.mm(a.t(), b, out_dtype=torch.float32))
    return out
