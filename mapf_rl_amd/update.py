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
import ctypes

import numpy as np
import torch

from ._lib import check, lib
from .fused import (ENC_OBS_PER_BLOCK, ENC_WGRAD0_PARTS, ENC_WGRAD_PARTS, RECUR_NARROW_AGENTS, RECUR_WEIGHT_ELEMS, PackedEncoder,
                    PackedRecurrence, mm_rows, pack_encoder_backward, recurrence_params, rows_buffer, ENC_ELEMENT)

GAMMA = 0.99
GRAD_CLIP = 40.0
SIDE_STREAM_MAX_ROWS = 262144
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
        self.step = 0
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
        self.step += 1
        check(lib.mapf_adam_step(self.numel, _ptr(self.params), _ptr(self.grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), _ptr(self.bf16),
                                 _ptr(self.scratch), _ptr(self.norm), float(lr), BETAS[0], BETAS[1], EPS, self.step, GRAD_CLIP,
                                 _stream(self.device)), "mapf_adam_step")
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


class FusedUpdate:
    def __init__(self, learner):
        self.lr = learner
        self.dev = learner.device
        self.flat = FlatParams(learner.model)
        self.packed_tar_recur = PackedRecurrence()
        self.packed_tar_enc = PackedEncoder()
        self.packed_on_recur = PackedRecurrence()
        self.packed_on_enc = PackedEncoder()
        self._tar_head = None

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

    # ------------------------------------------------------------------ plan: at sample time, one update ahead
    def plan(self, batch):
        """Launches the two closure kernels and the asynchronous copy of the per-window counts to pinned host memory."""
        v = self._views(batch)
        B, T, N, dev = v["B"], v["T"], v["N"], self.dev
        st = _stream(dev)
        po, pt = WindowPlan(T - FORWARD_STEPS, B, N, dev), WindowPlan(T, B, N, dev)
        counts = torch.empty((6, B), dtype=torch.int32, device=dev)  # cnt online, nag online, cnt target, nag target, distinct online, distinct target
        cm = v["comm"]
        from .model import Network

        mark_all = 0 if Network.PRUNE_UNREACHABLE else 1  # 1: encode every observation up to the window's last step, like the reference
        for k, (p, extra) in enumerate(((po, None), (pt, v["steps"]))):
            check(lib.mapf_plan_mark(_ptr(cm), cm.stride(0), cm.stride(1), _ptr(v["bt"]), _ptr(extra), p.T, B, N, mark_all, None, _ptr(p.slot), _ptr(p.order),
                                     _ptr(p.nact), _ptr(counts[2 * k]), _ptr(counts[2 * k + 1]), _ptr(counts[4 + k]), st), "mapf_plan_mark")
            p.cnt, p.nag, p.ucnt = counts[2 * k], counts[2 * k + 1], counts[4 + k]
        dup = None
        if self.DEDUP:
            # which entries repeat the observation of the same agent one step earlier (exact reuse: one encoder pass per run)
            obs = v["obs"]
            dup = torch.empty((T, B, N), dtype=torch.uint8, device=dev)
            check(lib.mapf_obs_dup(T, po.T, B, N, _ptr(obs), obs.stride(0), obs.stride(1), _ptr(po.slot), _ptr(pt.slot), _ptr(po.nact), _ptr(pt.nact),
                                   _ptr(dup), _ptr(po.ucnt), _ptr(pt.ucnt), st), "mapf_obs_dup")
        po.dup = pt.dup = dup
        host = torch.empty((6, B), dtype=torch.int32, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return dict(views=v, online=po, target=pt, counts=counts, host=host, event=ev)

    def _finish_plan(self, pl):
        """Host side of the plan: row totals and the compact width of both window sets (waits for the count copy -- long finished
        when the batch was planned during the update before), then `mapf_plan_rows`."""
        pl["event"].synchronize()
        h = pl["host"].numpy()
        v, dev = pl["views"], self.dev
        st = _stream(dev)
        for k, p in enumerate((pl["online"], pl["target"])):
            p.rows = int(h[2 * k].sum())
            p.nc = 16 * max(1, -(-int(h[2 * k + 1].max()) // 16))
            T, B, N, Nc = p.T, p.B, p.N, p.nc
            p.gidx = torch.empty((T, B, Nc), dtype=torch.int32, device=dev)
            p.comm_c = torch.empty((T, B, Nc, Nc), dtype=torch.uint8, device=dev)
            p.h0_c = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device=dev)
            # the observations to encode: every row, or -- with the duplicate flags -- the distinct ones (umap: entry -> distinct row)
            p.urows = int(h[4 + k].sum()) if p.dup is not None else p.rows
            p.obs_rows = rows_buffer((), max(p.urows, 1), (6, 9, 9), torch.bfloat16, dev)
            row_src = rows_buffer((), max(p.urows, 1), (), torch.int64, dev)
            p.umap = rows_buffer((), max(p.rows, 1), (), torch.int32, dev) if p.dup is not None else None
            p.row_tbp = rows_buffer((), max(p.rows, 1), (), torch.int32, dev) if p.dup is not None else None
            cm, obs, hid = v["comm"], v["obs"], v["hidden"]
            check(lib.mapf_plan_rows(T, B, N, Nc, _ptr(p.order), _ptr(p.nact), _ptr(p.cnt), _ptr(p.nag), _ptr(cm), cm.stride(0), cm.stride(1),
                                     _ptr(hid), int(hid.dtype == torch.bfloat16), _ptr(obs), obs.stride(0), obs.stride(1), _ptr(p.gidx),
                                     _ptr(p.comm_c), _ptr(p.h0_c), p.urows, _ptr(row_src), _ptr(p.obs_rows), _ptr(p.dup), _ptr(p.ucnt), _ptr(p.umap),
                                     _ptr(p.row_tbp), st), "mapf_plan_rows")
        return pl

    # ------------------------------------------------------------------ pieces
    def _w_ih(self, net, own):
        """bf16 [768, 784] input-projection weight: a view of the flat bf16 copy for the online network, a cast for the target."""
        if own:
            return self.flat.mem(self.flat.bf16, "recurrent.weight_ih")
        c = getattr(net, "_w_ih_bf16", None)
        w = net.recurrent.weight_ih
        key = (net.weights_epoch, w.data_ptr(), w._version)
        if c is None or c[0] != key:
            c = (key, w.detach().to(torch.bfloat16))
            net._w_ih_bf16 = c
        return c[1]

    def _infer_a0(self, net, penc, prec, p, own):
        """agent-0 states bf16 [T, B, 256] of `net` on the window set `p`, no gradients (target network; double-DQN's arg-max)."""
        dev, T, B, Nc = self.dev, p.T, p.B, p.nc
        st = _stream(dev)
        wp, bp = penc.get(net.obs_encoder, net.weights_epoch)
        w, b = prec.get(net)
        lat = rows_buffer((), p.urows, (784,), torch.bfloat16, dev)
        check(lib.mapf_encoder_forward(_ptr(p.obs_rows), 1, p.urows, _ptr(wp), _ptr(bp), _ptr(lat), st), "mapf_encoder_forward")
        gi = self._expand(mm_rows(lat, self._w_ih(net, own)), p)  # [rows, 768]
        compact = Nc <= RECUR_NARROW_AGENTS  # the <= 48-agent kernels read / write the rows that exist (gidx); the wide ones are dense
        if not compact:
            gi_rows, gi = gi, torch.empty((T, B, Nc, 768), dtype=torch.bfloat16, device=dev)
            check(lib.mapf_rows_scatter(_ptr(gi_rows), _ptr(p.gidx), _ptr(gi), T * B * Nc, 1536, 1, st), "mapf_rows_scatter")
        h_out = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device=dev)
        a0 = torch.empty((T, B, 256), dtype=torch.bfloat16, device=dev)
        check(lib.mapf_recurrent_infer(_ptr(gi), _ptr(p.h0_c), _ptr(p.comm_c), _ptr(w), _ptr(b), T, B, Nc, _ptr(h_out), _ptr(a0),
                                       _ptr(p.gidx) if compact else None, p.rows if compact else 0, st), "mapf_recurrent_infer")
        return a0

    def _expand(self, x_u, p):
        """Rows of the distinct observations [urows, w] -> one row per entry [rows, w] (umap); the identity without duplicate flags."""
        if p.umap is None:
            return x_u
        out = rows_buffer((), p.rows, (x_u.shape[1],), x_u.dtype, x_u.device)
        check(lib.mapf_rows_scatter(_ptr(x_u), _ptr(p.umap), _ptr(out), p.rows, x_u.shape[1] * x_u.element_size(), 1, _stream(self.dev)),
              "mapf_rows_scatter")
        return out

    DEDUP = True  # encode the distinct observations of a batch only (mapf_obs_dup: same agent, consecutive steps, same 486 values)

    # ------------------------------------------------------------------ the update
    def usable(self, batch):
        """The kernels' shape limits (include/mapf_dqn.h); anything else takes Learner's autograd path."""
        from .model import Network

        obs = batch[0]
        return (obs.is_cuda and obs.dim() == 6 and obs.shape[2] <= 128 and FORWARD_STEPS < obs.shape[1] <= 20 and Network.FUSED_TRAINING and
                Network.FUSED_INFERENCE and Network.FUSED_BPTT and Network.FUSED_RECURRENCE and Network.FAST_RECURRENCE)

    def run(self, batch, pl=None, own_batch=False):
        lr, dev, flat = self.lr, self.dev, self.flat
        model, tar = lr.model, lr.tar_model
        flat.sync()
        if pl is None:
            pl = self.plan(batch)
        pl = self._finish_plan(pl)
        v, po, pt = pl["views"], pl["online"], pl["target"]
        B, To, Tt, Nc = v["B"], po.T, pt.T, po.nc
        cur = torch.cuda.current_stream(dev)
        st = _stream(dev)
        G = flat.grads
        # ---- weight images of the online network, on THIS stream before the side stream may read them (double-DQN) ----
        wp, bp = self.packed_on_enc.get(model.obs_encoder, model.weights_epoch)
        w_rec, b_rec = self.packed_on_recur.get(model)
        wt = torch.empty(RECUR_WEIGHT_ELEMS, dtype=torch.bfloat16, device=dev)  # the backward kernel's transposed image
        check(lib.mapf_recurrent_pack(_ptr_array([p.detach() for p in recurrence_params(model)]), None, None, _ptr(wt), st), "mapf_recurrent_pack")
        wpt = pack_encoder_backward(model.obs_encoder)
        # ---- target network (and double-DQN's online arg-max) on the second stream ----
        # (beyond ~260 k rows -- every observation of 128-agent windows -- either network's kernels fill the chip for tens of
        # milliseconds: a second stream buys nothing there and doubles the transient allocations)
        side = lr._side if max(po.rows, pt.rows) <= SIDE_STREAM_MAX_ROWS else None
        a0_on2 = None
        if side is not None:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                a0_tg = self._infer_a0(tar, self.packed_tar_enc, self.packed_tar_recur, pt, False)
                if lr.double_q:
                    a0_on2 = self._infer_a0(model, self.packed_on_enc, self.packed_on_recur, pt, True)
                ready = torch.cuda.Event()
                ready.record(side)
        else:
            a0_tg = self._infer_a0(tar, self.packed_tar_enc, self.packed_tar_recur, pt, False)
            if lr.double_q:
                a0_on2 = self._infer_a0(model, self.packed_on_enc, self.packed_on_recur, pt, True)
        # ---- online network forward, saving what the backward needs ----
        M, Mu = po.rows, po.urows  # entries of the window set / distinct observations among them
        bf = torch.bfloat16
        acts = rows_buffer((7,), Mu, (7, 7, 128), ENC_ELEMENT, dev)
        lat = rows_buffer((), Mu, (784,), bf, dev)
        bits = rows_buffer((7,), Mu, (49, 4), torch.int32, dev)
        check(lib.mapf_encoder_forward_save(_ptr(po.obs_rows), 1, Mu, _ptr(wp), _ptr(bp), _ptr(lat), _ptr(acts), _ptr(bits), st),
              "mapf_encoder_forward_save")
        w_ih = self._w_ih(model, True)
        gi_rows = self._expand(mm_rows(lat, w_ih), po)
        compact = Nc <= RECUR_NARROW_AGENTS
        # rows of the recurrence's saved tensors / gradient outputs: the M rows that exist (compact: the <= 48-agent kernels address
        # them through gidx) or all To x B x Nc (step, window, position) entries (the wide kernels)
        # (compact: padded to a multiple of the weight-gradient GEMMs' split size, the padding rows zeroed below)
        R = -(-M // WGRAD_SPLIT) * WGRAD_SPLIT if compact else To * B * Nc
        ridx, nrows = (_ptr(po.gidx), R) if compact else (None, 0)
        if compact:
            gi = gi_rows
        else:
            gi = torch.empty((To, B, Nc, 768), dtype=bf, device=dev)
            check(lib.mapf_rows_scatter(_ptr(gi_rows), _ptr(po.gidx), _ptr(gi), To * B * Nc, 1536, 1, st), "mapf_rows_scatter")
        saves = [rows_buffer((), R, (256,), bf, dev), rows_buffer((), R, (1024,), bf, dev),
                 rows_buffer((2,), R, (256,), bf, dev), rows_buffer((2,), R, (384,), bf, dev),
                 rows_buffer((2,), R, (128,), bf, dev), rows_buffer((2,), R, (64,), bf, dev),
                 rows_buffer((2,), R, (1024,), bf, dev),
                 torch.empty((2, To * B, 2, 48, 64) if Nc <= RECUR_NARROW_AGENTS else (8,), dtype=bf, device=dev)]
        h_out = torch.empty((B, Nc, 256), dtype=bf, device=dev)
        a0 = torch.empty((To, B, 256), dtype=bf, device=dev)
        sp = _ptr_array(saves)
        check(lib.mapf_recurrent_forward_save(_ptr(gi), _ptr(po.h0_c), _ptr(po.comm_c), _ptr(w_rec), _ptr(b_rec), To, B, Nc, _ptr(h_out), _ptr(a0), sp,
                                              ridx, nrows, st), "mapf_recurrent_forward_save")
        # ---- dueling heads, TD error, priorities, loss and their gradients ----
        if side is not None:
            cur.wait_event(ready)
            for t in (a0_tg, a0_on2):  # allocated on the side stream, consumed on this one
                if t is not None:
                    t.record_stream(cur)
            for t in (pt.obs_rows, pt.gidx, pt.comm_c, pt.h0_c, pt.umap):  # allocated on this stream, read on the side stream
                if t is None:
                    continue
                t.record_stream(side)
        flat.grads.zero_()
        outs = torch.empty((3, B), dtype=torch.float32, device=dev)  # q, q_next, td
        prio = torch.empty(B, dtype=torch.float64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty(9 * B, dtype=torch.float32, device=dev)
        d_a0 = torch.empty((To, B, 256), dtype=bf, device=dev)
        head_names = ("adv.weight", "adv.bias", "state.weight", "state.bias")
        head_on = _ptr_array([flat.mem(flat.params, k) for k in head_names])
        if self._tar_head is None or self._tar_head[0] is not tar:
            tnamed = dict(tar.named_parameters())
            self._tar_head = (tar, [tnamed[k] for k in head_names])
        head_tg_t = [t.detach().to(torch.float32).contiguous() for t in self._tar_head[1]]
        head_g = _ptr_array([flat.mem(G, k) for k in head_names])
        check(lib.mapf_dqn_head_loss(B, To, Tt, _ptr(a0), _ptr(a0_tg), _ptr(a0_on2), _ptr(v["bt"]), _ptr(v["steps"]), _ptr(v["action"]),
                                     _ptr(v["reward"]), _ptr(v["done"]), _ptr(v["weights"]), head_on, _ptr_array(head_tg_t), GAMMA, _ptr(outs[0]),
                                     _ptr(outs[1]), _ptr(outs[2]), _ptr(prio), _ptr(loss), _ptr(scratch), _ptr(d_a0), head_g, st),
              "mapf_dqn_head_loss")
        idxes, old_ptr = batch[8], batch[10]
        if lr.replay_gate is not None:  # (actors on their own stream: their last episode flush precedes this update's replay operations)
            cur.wait_event(lr.replay_gate)
        if lr.buffer is not None and idxes is not None:
            lr.buffer.update_priorities(idxes, prio, old_ptr)                                     # worker.py:331 (values known here)
        pre_ready = None
        if lr.prefetch and own_batch:
            # the next batch is sampled and planned NOW (its priorities are in), on the second stream: ~0.5 ms of small kernels that
            # fit beside the backward-through-time kernel (192 workgroups on 256 CUs) instead of in front of it
            if lr._side is not None:
                lr._side.wait_stream(cur)
                with torch.cuda.stream(lr._side):
                    lr._launch_prefetch()
                    pre_ready = torch.cuda.Event()
                    pre_ready.record(lr._side)
            else:
                lr._launch_prefetch()
        # this update's replay operations (priority write-back, next sample) are enqueued: whoever else writes the replay waits for this
        lr.replay_released = pre_ready
        if pre_ready is None:
            lr.replay_released = torch.cuda.Event()
            lr.replay_released.record(cur)
        # ---- backward through time ----
        outs_b = [rows_buffer((), R, (768,), bf, dev), rows_buffer((), R, (768,), bf, dev),
                  rows_buffer((2,), R, (768,), bf, dev), rows_buffer((2,), R, (768,), bf, dev),
                  rows_buffer((2,), R, (64,), bf, dev), rows_buffer((2,), R, (384,), bf, dev),
                  torch.empty((B, 2432), dtype=torch.float32, device=dev)]
        check(lib.mapf_recurrent_backward(sp, _ptr(po.comm_c), _ptr(d_a0), _ptr(wt), To, B, Nc, _ptr_array(outs_b), ridx, nrows, st),
              "mapf_recurrent_backward")
        if compact and R > M:  # rows M..R of every GEMM operand (both rounds of the [2, R, w] tensors)
            ops = [saves[0], saves[2][0], saves[2][1], saves[4][0], saves[4][1], saves[5][0], saves[5][1], outs_b[1], outs_b[2][0], outs_b[2][1],
                   outs_b[3][0], outs_b[3][1], outs_b[4][0], outs_b[4][1], outs_b[5][0], outs_b[5][1]]
            rb = (ctypes.c_int * len(ops))(*[t.shape[-1] * 2 for t in ops])
            check(lib.mapf_zero_rows(_ptr_array(ops), rb, len(ops), M, R, st), "mapf_zero_rows")
        d_gi1, d_gh1, d_gi2, d_gh2, d_info, d_qkv, bsum = outs_b
        hin0, _, hr, _, ctxs, info, _, _ = saves
        hrf = hr.view(2 * R, 256)
        rows_k = WGRAD_SPLIT if compact else 8192
        _tall_tn_into(flat.mem(G, "recurrent.weight_hh"), d_gh1, hin0, rows_k)
        _tall_tn_into(flat.span(G, "comm.self_attn.W_Q.weight", "comm.self_attn.W_V.weight"), d_qkv.view(2 * R, 384), hrf, rows_k)
        _tall_tn_into(flat.mem(G, "comm.self_attn.W_O.weight"), d_info.view(2 * R, 64), ctxs.view(2 * R, 128), rows_k)
        _tall_tn_into(flat.mem(G, "comm.update_cell.weight_ih"), d_gi2.view(2 * R, 768), info.view(2 * R, 64), rows_k)
        _tall_tn_into(flat.mem(G, "comm.update_cell.weight_hh"), d_gh2.view(2 * R, 768), hrf, rows_k)
        bias_names = ("recurrent.bias_ih", "recurrent.bias_hh", "comm.self_attn.W_Q.bias", "comm.self_attn.W_K.bias", "comm.self_attn.W_V.bias",
                      "comm.update_cell.bias_ih", "comm.update_cell.bias_hh")
        check(lib.mapf_recurrent_bias_grads(_ptr(bsum), B, _ptr_array([flat.mem(G, k) for k in bias_names]), st), "mapf_recurrent_bias_grads")
        # ---- input projection ----
        if compact:
            d_gi_rows = d_gi1[:M]
        else:
            d_gi_rows = rows_buffer((), M, (768,), bf, dev)
            check(lib.mapf_rows_scatter(_ptr(d_gi_rows), _ptr(po.gidx), _ptr(d_gi1), R, 1536, 0, st), "mapf_rows_scatter")
        if po.umap is not None:  # gradient of a shared row = the sum over the entries that use it
            d_gi_u = rows_buffer((), Mu, (768,), bf, dev)
            check(lib.mapf_dedup_sum(To, B, Nc, M, 1536, _ptr(po.gidx), _ptr(po.umap), _ptr(po.row_tbp), _ptr(d_gi_rows), _ptr(d_gi_u), st),
                  "mapf_dedup_sum")
            d_gi_rows = d_gi_u
        g_lat = mm_rows(d_gi_rows, w_ih, transpose_w=False)
        _tall_tn_into(flat.mem(G, "recurrent.weight_ih"), d_gi_rows, lat, rows=4096)
        # ---- encoder: backward-data chain in one kernel, then the weight-gradient kernels ----
        self._encoder_backward(po.obs_rows, Mu, acts, lat, bits, g_lat, wpt)
        # ---- the only collective, clip, Adam ----
        lr.bucket.all_reduce_mean()
        if lr.grad_hook is not None:
            lr.grad_hook(lr)
        grad_norm = flat.adam_step(lr.current_lr())
        if pre_ready is not None:
            # everything the caller enqueues from here on (the next update; an actor step that appends to the replay) is ordered behind
            # the sample.  Memory: the prefetched tensors come from the second stream's pool and are consumed on this one -- safe without
            # record_stream because every use of the second stream starts with wait_stream(this one)
            cur.wait_event(pre_ready)
        return dict(loss=loss[0], td=outs[2].view(B, 1), priorities=prio, grad_norm=grad_norm, q=outs[0].view(B, 1), q_next=outs[1].view(B, 1))

    def _encoder_backward(self, obs_rows, M, acts, lat, bits, g_lat, wpt):
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
        check(lib.mapf_encoder_backward(_ptr(g_lat), _ptr(lat), M, _ptr(bits), _ptr(wpt), _ptr(gz), _ptr(gb_part), _ptr(gz7), _ptr(gb7_part),
                                        _ptr(scale), st), "mapf_encoder_backward")
        names = ["obs_encoder.0", "obs_encoder.2.block1", "obs_encoder.2.block2", "obs_encoder.3.block1", "obs_encoder.3.block2",
                 "obs_encoder.4.block1", "obs_encoder.4.block2", "obs_encoder.5"]
        # bias gradients: the kernel's per-workgroup partials, summed in two stages (see fused._EncoderTrain.backward)
        pad = (-nblk) % 256
        gp = gb_part if pad == 0 else torch.cat([gb_part, gb_part.new_zeros((7, pad, 128))], dim=1)
        torch.sum(gp.view(7, -1, 256, 128).sum(dim=2), dim=1, out=flat.span(G, names[0] + ".bias", names[6] + ".bias").view(7, 128))
        torch.sum(gb7_part, dim=0, out=flat.mem(G, names[7] + ".bias"))
        ws = torch.empty((ENC_WGRAD_PARTS, 128, 3, 3, 128), dtype=torch.float32, device=dev)
        for k in range(1, 7):
            check(lib.mapf_encoder_wgrad(_ptr(gz[k]), _ptr(acts[k - 1]), M, _ptr(scale), _ptr(ws), st), "mapf_encoder_wgrad")
            torch.sum(ws, dim=0, out=flat.mem(G, names[k] + ".weight"))  # [co][ky][kx][ci] == the weight's channels_last memory
        ws0 = torch.empty((ENC_WGRAD0_PARTS, 128, 64), dtype=torch.float32, device=dev)
        check(lib.mapf_encoder_wgrad0(_ptr(gz[0]), _ptr(obs_rows), 1, M, _ptr(scale), _ptr(ws0), st), "mapf_encoder_wgrad0")
        # conv0: columns j = ci*9 + ky*3 + kx -> the weight's memory [co][ky][kx][ci]
        flat.mem(G, names[0] + ".weight").copy_(ws0.sum(dim=0)[:, :54].view(128, 6, 3, 3).permute(0, 2, 3, 1))
        g7 = flat.mem(G, names[7] + ".weight").view(16, 128)
        _tall_tn_into(g7, gz7, acts[6].reshape(M * 49, 128))
        g7.mul_(scale.view(torch.float32)[1])


def _tall_tn_into(out, a, b, rows=8192):
    """out[m, n] = a^T b for a [K, m], b [K, n] with K in the 10^5 .. 10^6 range (model._tall_tn), written in place: K is split into
    batches of `rows` (bmm, 16-bit in / fp32 out) that are summed in fp32 straight into `out`.  (fp32 partial products: the
    encoder's f16 operands carry a loss scale, and a 16-bit partial of 8192 rows also costs the gradient 2-3 digits.)"""
    K, m = a.shape
    S = K // rows
    if S > 1:
        part = torch.bmm(a[:S * rows].view(S, rows, m).transpose(1, 2), b[:S * rows].view(S, rows, -1), out_dtype=torch.float32)
        torch.sum(part, dim=0, out=out)
        if K > S * rows:
            out += torch.mm(a[S * rows:].t(), b[S * rows:], out_dtype=torch.float32)
    else:
        out.copy_(torch.mm(a.t(), b, out_dtype=torch.float32))
