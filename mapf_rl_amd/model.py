"""Recurrent, communicating dueling DQN of the reference (reference model.py:139-263), re-expressed for
batched execution on one MI355X: E environments x N agents per actor step, [B, T, N] windows per update.

Compatibility surface kept: `Network()` has the reference's module tree, so `state_dict()` key names and
shapes are identical (`obs_encoder.{0,2,3,4,5}.*`, `recurrent.*`, `comm.self_attn.W_{Q,K,V,O}.*`,
`comm.update_cell.*`, `adv.*`, `state.*`; 2,050,582 parameters) and reference checkpoints load unchanged;
`Network.step(obs, pos)`, `.reset()` and `.bootstrap(obs, steps, hidden, comm_mask)` keep the reference's
signatures and return types (model.py:180-263).

Declared deviations from the reference:
  * `bootstrap` / `CommBlock` take the batch size from their inputs (the reference hard-codes
    config.batch_size, model.py:128,239,245,255 -- quirk Q5).
  * The 3-nearest-neighbour selection (model.py:203, `topk`) breaks distance ties by LOWEST agent index
    (deterministic); the reference's CPU topk order on ties is unspecified.  Parity tests inject the
    reference's comm_mask and check the mask itself on tie-free rows.
  * On a HIP device the forward runs under bf16 autocast (the reference uses fp16 autocast + GradScaler on
    CUDA and fp32 on CPU); attention scores / softmax stay fp32 as in model.py:75-78.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

OBS_SHAPE = (6, 9, 9)   # reference config.py:14
LATENT_DIM = 256        # config.py:56
OBS_RADIUS = 4          # config.py:7
MAX_COMM_AGENTS = 3     # config.py:59
NUM_COMM_LAYERS = 2     # config.py:62
NUM_COMM_HEADS = 2      # config.py:63
ENC_FEATURES = 16 * 7 * 7
RECUR_MAX_AGENTS = 128  # widest environment of the fused recurrence kernels (include/mapf_dqn.h); beyond it: the PyTorch-level path
RECUR_NARROW_AGENTS = 48  # the one-workgroup-per-environment kernels (csrc/mapf_recur.hip, mapf_recur_bwd.hip); above: mapf_recur_wide*
BPTT_MAX_AGENTS = 128


Q_HEAD_ROW_CHUNK = 32768  # rows per Q-head GEMM on a HIP device (see Network.q_head)


class ResBlock(nn.Module):
    """Two 3x3 pad-1 convolutions with an identity skip, no normalisation (reference model.py:7-42, type='cnn')."""

    def __init__(self, channel):
        super().__init__()
        self.block1 = nn.Conv2d(channel, channel, 3, 1, 1)
        self.block2 = nn.Conv2d(channel, channel, 3, 1, 1)

    def forward(self, x):
        return F.relu(self.block2(F.relu(self.block1(x))) + x)


class MultiHeadAttention(nn.Module):
    """Masked multi-head attention over the agents of one environment (reference model.py:45-87)."""

    def __init__(self, input_dim, output_dim, num_heads):
        super().__init__()
        self.num_heads, self.input_dim, self.output_dim = num_heads, input_dim, output_dim
        self.W_Q = nn.Linear(input_dim, output_dim * num_heads)
        self.W_K = nn.Linear(input_dim, output_dim * num_heads)
        self.W_V = nn.Linear(input_dim, output_dim * num_heads)
        self.W_O = nn.Linear(output_dim * num_heads, output_dim, bias=False)

    def forward(self, x, blocked):
        """x [B, N, input_dim]; blocked bool [B, N, N] (True = may NOT attend)."""
        B, N, _ = x.shape
        H, D = self.num_heads, self.output_dim
        q = self.W_Q(x).view(B, N, H, D).transpose(1, 2)
        k = self.W_K(x).view(B, N, H, D).transpose(1, 2)
        v = self.W_V(x).view(B, N, H, D).transpose(1, 2)
        if Network.SDPA and x.is_cuda and q.dtype == torch.bfloat16:
            # fused kernel, fp32 accumulation of scores / softmax from the bf16 q, k (see Network._recur_fast)
            ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=_sdpa_mask(~blocked).unsqueeze(1))
            return self.W_O(ctx.transpose(1, 2).reshape(B, N, H * D))
        # scores and softmax in fp32 (model.py:75-78)
        scores = torch.matmul(q.float(), k.float().transpose(-1, -2)) / (D ** 0.5)
        scores = scores.masked_fill(blocked.unsqueeze(1), -1e9)
        attn = F.softmax(scores, dim=-1)
        ctx = torch.matmul(attn.to(v.dtype), v)
        ctx = ctx.transpose(1, 2).reshape(B, N, H * D)
        return self.W_O(ctx)


class CommBlock(nn.Module):
    """`NUM_COMM_LAYERS` rounds (shared weights) of attention + GRU update, applied only to agents that have
    at least one communication partner (reference model.py:89-135)."""

    def __init__(self, input_dim, output_dim=64, num_heads=NUM_COMM_HEADS, num_layers=NUM_COMM_LAYERS):
        super().__init__()
        self.input_dim, self.output_dim, self.num_layers = input_dim, output_dim, num_layers
        self.self_attn = MultiHeadAttention(input_dim, output_dim, num_heads)
        self.update_cell = nn.GRUCell(output_dim, input_dim)

    def forward(self, latent, comm_mask):
        """latent [B, N, input_dim]; comm_mask bool [B, N, N] (True = communicates)."""
        B, N, _ = latent.shape
        update = (comm_mask.sum(dim=-1) > 1).unsqueeze(-1)  # model.py:103
        blocked = ~comm_mask
        for _ in range(self.num_layers):
            info = self.self_attn(latent, blocked)
            new = self.update_cell(info.reshape(B * N, self.output_dim), latent.reshape(B * N, self.input_dim))
            latent = torch.where(update, new.view(B, N, self.input_dim).to(latent.dtype), latent)
        return latent


class _WGradSink:
    """Weight gradients of the linears inside the T-step recurrence, deferred to the end of the backward pass.

    Autograd would run one small GEMM per (time step, weight) -- [out, in] = dy_t^T x_t with K = B*N = 7,680 rows,
    224 launches of ~55 us per update that each fill a fraction of the chip -- and then add each result into
    .grad.  The sink keeps (x_t, dy_t) instead and, from an end-of-backward callback (the hook DDP finalises in),
    runs ONE GEMM per weight over the concatenated K = T*B*N rows and accumulates straight into .grad."""

    def __init__(self):
        self.items = {}
        self.queued = False

    def add(self, key, x, dy):
        ent = self.items.get(id(key))
        if ent is None:
            ent = self.items[id(key)] = (key, [], [])
        ent[1].append(x)
        ent[2].append(dy)
        if not self.queued:
            self.queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        items, self.items, self.queued = self.items, {}, False
        for key, xs, dys in items.values():
            X = torch.cat(xs, dim=0) if len(xs) > 1 else xs[0]
            DY = torch.cat(dys, dim=0) if len(dys) > 1 else dys[0]
            gw = _tall_tn(DY, X)  # [out, in] fp32
            gb = DY.sum(dim=0, dtype=torch.float32)
            for w, b, lo, hi in key:
                _accumulate(w, gw[lo:hi])
                if b is not None:
                    _accumulate(b, gb[lo:hi])


def _tall_tn(a, b, rows=8192):
    """a^T b for a [K, m], b [K, n] with K in the 10^5..10^6 range and m, n a few hundred at most: one GEMM call gives the
    BLAS library an output of a handful of tiles (a fraction of the chip walking all of K), so K is split into
    batches of `rows` (bmm, bf16 in / fp32 accumulate inside each batch) that are then summed in fp32."""
    K, m = a.shape
    S = K // rows
    out = None
    if S > 1:
        out = torch.bmm(a[:S * rows].view(S, rows, m).transpose(1, 2), b[:S * rows].view(S, rows, -1)).sum(dim=0, dtype=torch.float32)
        if K > S * rows:
            out += torch.mm(a[S * rows:].t(), b[S * rows:]).float()
        return out
    return torch.mm(a.t(), b).float()


def _accumulate(p, g):
    if p.grad is None:
        p.grad = g.to(p.dtype).reshape(p.shape).clone()
    else:
        p.grad.add_(g.reshape(p.shape))


class _InputProj(torch.autograd.Function):
    """y = x W^T (+ b) for the GRU input projection over all T*B*N rows, with the backward GEMMs in the forms that run well at
    this shape (tools/micro/proj_gemm.py, R = 122,880 rows x 784 -> 768): d_x through a transposed copy of W (TN, 0.21 ms; the
    NN GEMM autograd would issue takes 0.33 ms) and d_W split along K (0.25 ms; as one GEMM its 6 x 7 output tiles occupy a
    sixth of the chip for 0.64 ms)."""

    @staticmethod
    def forward(ctx, x, w, b):
        from .fused import mm_rows

        w_lp = w.detach().to(x.dtype)
        ctx.save_for_backward(x, w_lp)
        ctx.has_bias = b is not None
        if x.is_cuda and x.dtype == torch.bfloat16:  # row-chunked (fused.mm_rows): no stream-K GEMM next to the second stream
            y = mm_rows(x, w_lp)
            return y if b is None else y + b.detach().to(x.dtype)
        return F.linear(x, w_lp, None if b is None else b.detach().to(x.dtype))

    @staticmethod
    def backward(ctx, dy):
        x, w_lp = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.is_cuda and dy.dtype == torch.bfloat16:
            from .fused import mm_rows

            dx = mm_rows(dy, w_lp, transpose_w=False) if ctx.needs_input_grad[0] else None
        else:
            dx = F.linear(dy, w_lp.t().contiguous()) if ctx.needs_input_grad[0] else None
        dw = _tall_tn(dy, x, rows=4096)
        db = dy.sum(dim=0, dtype=torch.float32) if ctx.has_bias else None
        return dx, dw, db


class _TimeLinear(torch.autograd.Function):
    """y = x W^T (+ b) for one time step of the recurrence; dx in backward, dW/db deferred to `sink`."""

    @staticmethod
    def forward(ctx, x, w_lp, b_lp, sink, key, anchor):
        # `anchor` (the fp32 weight Parameter) only makes autograd record this node when x itself carries no
        # gradient (the first time step); it receives no gradient here -- the sink accumulates into .grad directly
        ctx.save_for_backward(x, w_lp)
        ctx.sink, ctx.key = sink, key
        return F.linear(x, w_lp, b_lp)

    @staticmethod
    def backward(ctx, dy):
        x, w_lp = ctx.saved_tensors
        dy = dy.contiguous()
        ctx.sink.add(ctx.key, x, dy)
        return (torch.mm(dy, w_lp) if ctx.needs_input_grad[0] else None), None, None, None, None, None


class _Select(torch.autograd.Function):
    """torch.where(cond, a, b) whose backward is two masked multiplies (autograd's own makes two zero tensors and two
    more where-kernels per call; this runs 32 times per update)."""

    @staticmethod
    def forward(ctx, cond, a, b):
        ctx.save_for_backward(cond)
        return torch.where(cond, a, b)

    @staticmethod
    def backward(ctx, g):
        (cond,) = ctx.saved_tensors
        return None, g * cond, g * ~cond


def _sdpa_mask(allowed):
    """Boolean SDPA mask for `allowed` [..., N, N].  SDPA fills masked scores with -inf, the reference with -1e9
    (model.py:77): the same softmax unless a row has NO allowed entry -- zero-padded window rows, padded agents of
    smaller curriculum levels, the last comm row of finished episodes (buffer.py:124) --, where -inf gives NaN (and
    0 * NaN in the GRU backward would poison the weight gradients) while -1e9 gives a uniform row.  Such an agent has
    no partner, so CommBlock discards its output (`update`, model.py:103); letting it attend to itself keeps the
    row finite without changing any result."""
    empty = ~allowed.any(dim=-1, keepdim=True)
    eye = torch.eye(allowed.shape[-1], dtype=torch.bool, device=allowed.device)
    return allowed | (empty & eye)


def relevance(comm_mask: torch.Tensor, steps: torch.Tensor) -> torch.Tensor:
    """Which (step, sample, agent) entries of a training window can influence its output at all.

    `bootstrap` learns from agent 0's hidden state at step `steps - 1` only (model.py:248,255), and within a step an agent's state
    depends on another's only through the communication mask: CommBlock runs two attention rounds in which agent i reads the
    agents j with comm_mask[b, t, i, j] (model.py:116-130).  Everything else -- encoder, GRU cells, Q head -- is per agent.
    So the entries that matter are the backward closure of {agent 0 at step steps - 1}: per step two hops along the mask,
    carried to the step before through the agent's own recurrent state.  Entries outside it (other agents' observations that never
    reach agent 0, steps behind the window's end) contribute exactly nothing to q and receive an exactly zero gradient.

    comm_mask bool [B, T, N, N]; steps int64 [B] (1-based) -> bool [T, B, N] (time-major, as `bootstrap` lays out its latents)."""
    B, T, N, _ = comm_mask.shape
    if comm_mask.is_cuda and N <= 128:  # one launch (csrc/mapf_dqn.hip) instead of ~20 per step
        from .fused import window_relevance

        return window_relevance(comm_mask, steps.view(B))
    m = comm_mask.to(torch.bool)
    last = steps.to(comm_mask.device).view(B) - 1
    agent0 = torch.zeros((B, N), dtype=torch.bool, device=comm_mask.device)
    agent0[:, 0] = True
    r = torch.zeros((B, N), dtype=torch.bool, device=comm_mask.device)
    rel = torch.zeros((T, B, N), dtype=torch.bool, device=comm_mask.device)
    for t in range(T - 1, -1, -1):
        r = r | (agent0 & (last == t).view(B, 1))
        for _ in range(2):  # i needed and i reads j  =>  j needed
            r = r | (r.unsqueeze(2) & m[:, t]).any(dim=1)
        rel[t] = r
    return rel


def comm_mask_from_pos(pos: torch.Tensor, obs_radius: int = OBS_RADIUS, max_comm: int = MAX_COMM_AGENTS) -> torch.Tensor:
    """pos [E, N, 2] (any integer/float dtype) -> bool [E, N, N]: j is within i's FOV square AND among i's
    `max_comm` nearest agents by Euclidean distance, itself included (reference model.py:195-208).
    Distance ties are broken by lowest agent index."""
    if pos.is_cuda and pos.dtype == torch.int16 and pos.shape[1] <= 128 and max_comm <= 8:
        from .fused import comm_mask  # one small HIP kernel instead of an [E, N, N] int64 topk + scatter

        return comm_mask(pos, obs_radius, max_comm)[0]
    p = pos.to(torch.int64)
    E, N, _ = p.shape
    d = (p.unsqueeze(2) - p.unsqueeze(1)).abs()              # [E, N, N, 2]
    in_fov = (d <= obs_radius).all(-1)
    d2 = d[..., 0] ** 2 + d[..., 1] ** 2                      # same ordering as the reference's sqrt
    key = d2 * N + torch.arange(N, device=p.device)           # unique keys: ties -> lowest index first
    k = min(max_comm, N)
    nearest = key.topk(k, dim=-1, largest=False).indices
    near = torch.zeros((E, N, N), dtype=torch.bool, device=p.device)
    near.scatter_(2, nearest, True)
    return in_fov & near


class Reach:
    """What `relevance` marks, in the form `bootstrap` uses: rows = indices of the marked entries in the time-major [T*B*N] order;
    agents bool [B, N] = the agents marked at step 0 (the set only grows going back in time, so these are all agents that matter
    anywhere in the window); max_agents = the largest such set of the batch, as a host integer."""

    def __init__(self, rows, agents, max_agents):
        self.rows, self.agents, self.max_agents = rows, agents, int(max_agents)

    def compact(self, lat, hidden, comm_mask):
        """The window restricted to its marked agents, padded to a multiple of 16 per window with agents that read only themselves:
        (latents as _SparseRows over [T, B, Nc], hidden [B*Nc, 256], comm_mask [B, T, Nc, Nc]).  Agent 0 stays agent 0."""
        B, T, N, _ = comm_mask.shape
        Nc = 16 * max(1, -(-self.max_agents // 16))
        dev = comm_mask.device
        order = torch.argsort((~self.agents).to(torch.uint8), dim=1, stable=True)[:, :Nc]      # marked agents first, ascending
        valid = torch.arange(Nc, device=dev).view(1, Nc) < self.agents.sum(dim=1, keepdim=True)  # [B, Nc]
        pos = torch.zeros((B, N), dtype=torch.int64, device=dev).scatter_(1, order, torch.arange(Nc, device=dev).expand(B, Nc))
        cm = comm_mask.gather(2, order.view(B, 1, Nc, 1).expand(B, T, Nc, N)).gather(3, order.view(B, 1, 1, Nc).expand(B, T, Nc, Nc))
        cm = (cm & valid.view(B, 1, Nc, 1) & valid.view(B, 1, 1, Nc)) | torch.eye(Nc, dtype=torch.bool, device=dev)
        h = hidden.view(B, N, -1).gather(1, order.view(B, Nc, 1).expand(B, Nc, hidden.shape[-1])) * valid.view(B, Nc, 1).to(hidden.dtype)
        n = self.rows % N
        tb = self.rows // N
        rows_c = tb * Nc + pos[tb % B, n]
        return _SparseRows(lat, rows_c, (T, B, Nc)), h.reshape(B * Nc, -1), cm


class _SparseRows:
    """Latents of the reachable observations only: `rows` index the time-major [T*B*N] rows, every other row is zero."""

    def __init__(self, values, rows, tbn):
        self.values, self.rows, self.shape = values, rows, tuple(tbn) + (values.shape[1],)

    def dense(self):
        T, B, N, F_ = self.shape
        return torch.zeros((T * B * N, F_), dtype=self.values.dtype, device=self.values.device).index_copy(0, self.rows, self.values).view(T, B, N, F_)

    def project(self, fn, width):
        """fn applied to the stored rows only ([M', F] -> [M', width]; a bias-free linear map keeps the other rows zero), scattered
        into the dense [T, B, N, width] the recurrence kernels read."""
        T, B, N, _ = self.shape
        out = fn(self.values)
        return torch.zeros((T * B * N, width), dtype=out.dtype, device=out.device).index_copy(0, self.rows, out).view(T, B, N, width)


class Network(nn.Module):
    def __init__(self, cnn_channel: int = 64):  # `cnn_channel` is accepted and ignored like the reference (model.py:140)
        super().__init__()
        self.latent_dim = LATENT_DIM
        self.obs_encoder = nn.Sequential(
            nn.Conv2d(OBS_SHAPE[0], 128, 3, 1),
            nn.ReLU(True),
            ResBlock(128),
            ResBlock(128),
            ResBlock(128),
            nn.Conv2d(128, 16, 1, 1),
            nn.ReLU(True),
            nn.Flatten(),
        )
        self.recurrent = nn.GRUCell(ENC_FEATURES, self.latent_dim)
        self.comm = CommBlock(self.latent_dim)
        self.adv = nn.Linear(self.latent_dim, 5)
        self.state = nn.Linear(self.latent_dim, 1)
        self.hidden = None
        self._packed = None  # PackedEncoder of the fused inference kernel, built on first use
        self._packed_recur = None  # PackedRecurrence of the fused recurrence kernel
        self.weights_epoch = 0  # bumped by whoever writes the parameters behind PyTorch's back (the learner's fused Adam kernel)
        # model.py:174-178: Xavier-uniform weights / zero bias on Linear and Conv2d only
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        # NHWC weights: MIOpen's bf16 NHWC convolutions run 1.5-1.6x faster than NCHW at this shape on MI355X
        # (418-447 vs 277-279 TFLOP/s for the 128->128 3x3 layers, tools/conv_bench.py).  Shapes and
        # state_dict values are unaffected (only strides change).
        self.obs_encoder.to(memory_format=torch.channels_last)

    # ------------------------------------------------------------------ building blocks
    def _autocast(self, device):
        on_gpu = device.type == "cuda"
        return torch.autocast(device_type="cuda" if on_gpu else "cpu", dtype=torch.bfloat16, enabled=on_gpu)

    ENCODE_CHUNK = 32768  # observations per convolution call (see encode)
    FUSED_EPILOGUE = True  # hand-written bias/residual/ReLU epilogue kernels behind every convolution (HIP, bf16)
    FUSED_TRAINING = True   # with autograd: fused forward that saves the layer outputs + layer-wise backward on them
    FUSED_INFERENCE = True  # without autograd: the whole encoder as one hand-written MFMA kernel (csrc/mapf_encoder.hip)

    def encode(self, obs):
        """obs [M, 6, 9, 9] (uint8 / bool / float) -> [M, 784].
        On a HIP device under bf16 autocast the hand-written kernels run (csrc/mapf_encoder.hip, mapf_wgrad.hip): one
        launch without autograd, forward-with-saved-activations + backward-data + weight-gradient launches with it.
        Otherwise (CPU, fp32, the FUSED_* switches off) the layer-by-layer module path runs; on the GPU it is processed
        in chunks of ENCODE_CHUNK observations: MIOpen's convolution for one 138,240-observation call (192x18x40, the
        learner's window batch) returned non-finite rows for finite inputs and weights on ROCm 7.2 / gfx950, and
        chunking bounds the activation footprint (1.7 GB per layer at 138k observations)."""
        w = self.obs_encoder[0].weight
        nhwc = w.device.type == "cuda"  # weights are stored channels_last (see __init__)
        bf16_autocast = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if nhwc and self.FUSED_INFERENCE and bf16_autocast and not torch.is_grad_enabled():
            # actor steps and the target network's bootstrap: the whole encoder in one LDS-resident MFMA kernel
            from .fused import PackedEncoder, encoder_forward

            if self._packed is None:
                self._packed = PackedEncoder()
            wp, bp = self._packed.get(self.obs_encoder, self.weights_epoch)
            return encoder_forward(obs, wp, bp)
        if nhwc and self.FUSED_TRAINING and bf16_autocast and torch.is_grad_enabled():
            # learner: the same kernel also stores the layer outputs the backward chain needs
            from .fused import PackedEncoder, encoder_forward_train

            if self._packed is None:
                self._packed = PackedEncoder()
            return encoder_forward_train(obs, self.obs_encoder, self._packed, self.weights_epoch)

        fused = nhwc and self.FUSED_EPILOGUE and bf16_autocast

        def run(x):
            if fused:
                return self._encode_fused(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
            x = x.to(w.dtype) if x.dtype != w.dtype else x
            if nhwc:
                x = x.contiguous(memory_format=torch.channels_last)
            return self.obs_encoder(x)

        M = obs.shape[0]
        if M <= self.ENCODE_CHUNK:
            return run(obs)
        parts = -(-M // self.ENCODE_CHUNK)          # equal-sized chunks: one convolution shape per call site
        return torch.cat([run(c) for c in obs.split(-(-M // parts))], dim=0)

    def _encode_fused(self, x):
        """The encoder (model.py:147-162) with every bias / residual / ReLU fused behind its convolution by the
        hand-written epilogue kernel (mapf_rl_amd/fused.py): bf16, NHWC, under autocast."""
        from .fused import bias_res_relu

        enc = self.obs_encoder
        h = bias_res_relu(F.conv2d(x, enc[0].weight, None), enc[0].bias)
        for blk in (enc[2], enc[3], enc[4]):
            t = bias_res_relu(F.conv2d(h, blk.block1.weight, None, 1, 1), blk.block1.bias)
            h = bias_res_relu(F.conv2d(t, blk.block2.weight, None, 1, 1), blk.block2.bias, h)
        h = bias_res_relu(F.conv2d(h, enc[5].weight, None), enc[5].bias)
        return h.flatten(1)

    def prepack(self):
        """(Re)builds the fused kernels' packed weight images on the CURRENT stream if a parameter changed since the last pack.
        For callers that run this network on two streams (the learner's double-DQN arg-max on the side stream beside the online
        forward): packing is lazy and keyed on the host, so the first user would otherwise pack on ITS stream while the other
        stream, seeing the key already updated, reads a half-written image."""
        from .fused import PackedEncoder, PackedRecurrence

        if self._packed is None:
            self._packed = PackedEncoder()
        if self._packed_recur is None:
            self._packed_recur = PackedRecurrence()
        self._packed.get(self.obs_encoder, self.weights_epoch)
        self._packed_recur.get(self)

    def q_head(self, hidden):
        if hidden.is_cuda and hidden.dim() == 2 and hidden.shape[0] > Q_HEAD_ROW_CHUNK:
            # tall GEMMs (beyond ~10^5 rows: 4096 x 40 actor rows) make hipBLASLt pick stream-K kernels, whose workgroups spin on peers
            # that may not be resident next to another stream's work (fused.mm_rows; the actors run beside the update): row chunks
            return torch.cat([self.q_head(h) for h in hidden.split(Q_HEAD_ROW_CHUNK)])
        adv = self.adv(hidden)
        return self.state(hidden) + adv - adv.mean(-1, keepdim=True)  # model.py:218,262

    def _policy_head(self, hidden, q_out=None, act_out=None):
        """(Q-values f32 [rows, 5], greedy actions int64 [rows]) of hidden [rows, 256] (model.py:216-220): one kernel on a HIP device
        (fp32 accumulation over the bf16 states and the fp32 head parameters), the module path otherwise."""
        if self.FUSED_INFERENCE and hidden.is_cuda and hidden.dtype == torch.bfloat16 and hidden.is_contiguous() and self.adv.weight.dtype == torch.float32:
            from .fused import q_head_infer

            return q_head_infer(hidden, self.adv, self.state, q_out, act_out)
        q = self.q_head(hidden).float()
        if q_out is not None:
            q = q_out.copy_(q)
        return q, (q.argmax(-1) if act_out is None else torch.argmax(q, dim=-1, out=act_out))

    # ------------------------------------------------------------------ actor side
    @torch.no_grad()
    def step_batch(self, obs, pos, hidden: Optional[torch.Tensor], comm_mask: Optional[torch.Tensor] = None, cache=None):
        """One actor step for E environments at once.
        obs [E, N, 6, 9, 9], pos [E, N, 2], hidden [E*N, 256] or None (episode start, model.py:186-189).
        Returns (actions int64 [E, N], q float32 [E, N, 5], hidden [E*N, 256] (compute dtype), comm_mask bool [E, N, N]).
        `cache` (fused.LatentCache, HIP device, uint8 observations in a persistent buffer): only the agents whose observation changed
        since the previous call are encoded again -- same latents, bit for bit."""
        E, N = obs.shape[:2]
        with self._autocast(obs.device):
            if cache is not None and obs.is_cuda and obs.dtype == torch.uint8 and obs.is_contiguous() and self.FUSED_INFERENCE:
                from .fused import PackedEncoder

                if self._packed is None:
                    self._packed = PackedEncoder()
                fused_rec = self.FUSED_RECURRENCE and N <= RECUR_MAX_AGENTS
                latent = cache.encode(obs.view(E * N, *OBS_SHAPE), self._packed, self.obs_encoder, self.weights_epoch,
                                      proj=self._input_proj_weight() if fused_rec else None)
                gi_cached = cache.gi if fused_rec else None
            else:
                latent = self.encode(obs.reshape(E * N, *OBS_SHAPE))
                gi_cached = None
            if comm_mask is None:
                comm_mask = comm_mask_from_pos(pos)
            if self.FUSED_RECURRENCE and latent.is_cuda and latent.dtype == torch.bfloat16 and N <= RECUR_MAX_AGENTS:
                # GRU cell + both communication rounds in one kernel, an environment's states in one workgroup's LDS (csrc/mapf_recur.hip)
                hidden = self._recur_kernel(latent.view(1, E, N, ENC_FEATURES), hidden, comm_mask.unsqueeze(0), False, gi=gi_cached)[0]
                hidden = hidden.view(E * N, self.latent_dim)
            else:
                hidden = self.recurrent(latent) if hidden is None else self.recurrent(latent, hidden.to(latent.dtype))
                hidden = self.comm(hidden.view(E, N, self.latent_dim), comm_mask).reshape(E * N, self.latent_dim)
            q, act = self._policy_head(hidden)
        return act.view(E, N), q.view(E, N, 5), hidden, comm_mask

    @torch.no_grad()
    def step_levels(self, levels, obs_all, cache=None, hidden_out=None, packed_inplace=False, merged=None, q_out=None, act_out=None):
        """One actor step for SEVERAL groups of environments of different shapes at once -- the reference draws a (num_agents, map)
        level per episode inside one actor (environment.py:148-151, worker.py:422-428); here every active level is a group of E_l
        lock-step environments of N_l agents.  The encoder and the GRU input projection are per observation and do not care about
        the shape: ONE encoder launch and ONE GEMM over the concatenated observations; the recurrence (attention mixes the agents of
        one environment) runs per level; ONE Q head over the concatenated hidden states.
        levels: list of (E, N, pos [E, N, 2], hidden [E*N, 256] or None, comm bool [E, N, N]); obs_all uint8 [sum E_l N_l, 6, 9, 9],
        the levels' observations back to back (every level's `obs` is a view of it).  Returns per level what step_batch returns.
        hidden_out: optional bf16 [sum E_l N_l, 256] buffer the levels' new hidden states are written into (the returned `hidden`s are
        views of it, and the Q head reads it as one tensor); packed_inplace: re-pack changed weights into the buffers of the last pack
        (callers that replay this launch sequence from a captured graph: the graph holds the buffers' addresses); merged: optional
        (envtab int32 [sum E_l, 4], comm_all uint8, hidden_all bf16 [rows, 256]) -- the levels' masks and incoming states in ONE buffer
        each plus the per-environment table of fused.recurrent_infer_multi: the recurrence of all levels is then one launch (every
        level <= 16 agents; needs hidden_out); q_out f32 [rows, 5] / act_out int64 [rows]: buffers for the Q-values and greedy actions."""
        dev = obs_all.device
        if not (self.FUSED_RECURRENCE and self.FUSED_INFERENCE and dev.type == "cuda" and all(lv[1] <= RECUR_MAX_AGENTS for lv in levels)):
            outs, off = [], 0
            for E, N, pos, hidden, comm in levels:
                outs.append(self.step_batch(obs_all[off:off + E * N].view(E, N, *OBS_SHAPE), pos, hidden, comm))
                off += E * N
            return outs
        from .fused import PackedEncoder, PackedRecurrence, input_proj_rows, recurrent_infer

        with self._autocast(dev):
            if self._packed is None:
                self._packed = PackedEncoder()
            if self._packed_recur is None:
                self._packed_recur = PackedRecurrence()
            proj = self._packed_recur.input_weight_packed(self, inplace=packed_inplace)
            if cache is not None and obs_all.dtype == torch.uint8 and obs_all.is_contiguous():
                latent = cache.encode(obs_all, self._packed, self.obs_encoder, self.weights_epoch, proj=proj)
                gi_all = cache.gi  # (kept beside the latents: recomputed for the rows whose observation changed)
            else:
                latent = self.encode(obs_all)
                gi_all = input_proj_rows(latent.contiguous(), proj[0])
            w, b = self._packed_recur.get(self, inplace=packed_inplace)
            hs, off = [], 0
            if merged is not None:
                from .fused import recurrent_infer_multi

                envtab, comm_all, hidden_all = merged
                recurrent_infer_multi(gi_all, hidden_all, comm_all, w, b, envtab, hidden_out)
                for E, N, pos, hidden, comm in levels:
                    hs.append(hidden_out[off:off + E * N])
                    off += E * N
            else:
                for E, N, pos, hidden, comm in levels:
                    h0 = None if hidden is None else hidden.reshape(E, N, self.latent_dim)
                    out = None if hidden_out is None else hidden_out[off:off + E * N]
                    hs.append(recurrent_infer(gi_all[off:off + E * N].view(1, E, N, 768), h0, comm.unsqueeze(0), w, b, False, out=out)[0].view(E * N, self.latent_dim))
                    off += E * N
            h_all = hidden_out if hidden_out is not None else (torch.cat(hs) if len(hs) > 1 else hs[0])
            q_all, act_all = self._policy_head(h_all, q_out, act_out)  # (one launch for all levels; the per-level results are views)
        outs, off = [], 0
        for (E, N, pos, hidden, comm), h in zip(levels, hs):
            outs.append((act_all[off:off + E * N].view(E, N), q_all[off:off + E * N].view(E, N, 5), h, comm))
            off += E * N
        return outs

    @torch.no_grad()
    def step(self, obs, pos, comm_mask=None):
        """Reference-compatible single-environment step (model.py:180-222): keeps the recurrent state in
        `self.hidden`; returns (actions list[int], q ndarray [N,5], hidden ndarray [N,256], comm_mask ndarray [N,N])."""
        dev = self.adv.weight.device
        obs = torch.as_tensor(obs).to(dev)
        pos = torch.as_tensor(pos).to(dev)
        cm = None if comm_mask is None else torch.as_tensor(comm_mask).to(dev).unsqueeze(0)
        actions, q, self.hidden, cm = self.step_batch(obs.unsqueeze(0), pos.unsqueeze(0), self.hidden, cm)
        return (actions[0].tolist(), q[0].cpu().numpy(), self.hidden.float().cpu().numpy(), cm[0].cpu().numpy())

    def reset(self):
        self.hidden = None

    # ------------------------------------------------------------------ learner side
    FUSED_BPTT = True  # with autograd: the T-step recurrence as a forward-save + a backward-through-time kernel
    FUSED_RECURRENCE = True  # without autograd: GRU + CommBlock of all steps in one kernel (csrc/mapf_recur.hip, mapf_recur_wide.hip)
    SDPA = True  # fused scaled-dot-product attention inside _recur_fast
    PRUNE_UNREACHABLE = True  # HIP device, `bootstrap`: encode only the observations that can reach agent 0's Q-value (see `relevance`)
    FAST_RECURRENCE = True  # HIP device: hoisted input projection, fused QKV, deferred weight gradients (see _recur_fast)

    def bootstrap(self, obs, steps, hidden, comm_mask, rows=None):
        """Training forward over a [B, T] window of N agents (model.py:227-263).
        obs [B, T, N, 6, 9, 9]; steps int64 [B] (1-based index of the step whose agent-0 hidden feeds the Q head);
        hidden [B*N, 256]; comm_mask bool [B, T, N, N].  Returns q [B, 5] (float32).
        `rows` (optional, HIP path): a `Reach` -- the indices into the time-major [T*B*N] observation rows that `relevance` marks,
        which agents they belong to and how many of those a window has at most -- when the caller has it already (the learner plans
        it one update ahead, so that the two counts are on the host without a wait)."""
        B, T, N = obs.shape[:3]
        with self._autocast(obs.device):
            if self.FAST_RECURRENCE and obs.is_cuda and torch.get_autocast_dtype("cuda") == torch.bfloat16:
                # time-major from the start: transposing the raw observation bytes is 3x cheaper than transposing the latent
                # (and its gradient, on the way back), and the encoder does not care about row order
                obs_t = obs.transpose(0, 1).contiguous()
                if self.PRUNE_UNREACHABLE:
                    # only the observations agent 0's Q-value can depend on go through the encoder (see `relevance`): at 40
                    # agents that is ~1/9 of the window -- the encoder is 80 % of an update -- and the result is the same, not an
                    # approximation.  The other rows of the latent stay zero: their agents run through the recurrence on
                    # meaningless states that, by construction, nobody who matters reads.
                    if rows is None:
                        rel = relevance(comm_mask, steps)
                        rows = Reach(rel.view(-1).nonzero().squeeze(1), rel[0], int(rel[0].sum(dim=1).max()))  # (host syncs: two counts)
                    lat = self.encode(obs_t.view(T * B * N, *OBS_SHAPE).index_select(0, rows.rows))
                    if N > RECUR_NARROW_AGENTS and 16 * -(-rows.max_agents // 16) < N:
                        # more agents than the one-workgroup-per-window kernels take (48): run the recurrence on the ones that
                        # matter (the closure is closed under "reads": none of them reads an agent left out) -- through the 48-agent
                        # kernels when at most 48 do in every window of the batch, else through the wide ones at 64 instead of 128
                        sparse, hidden_c, comm_c = rows.compact(lat, hidden, comm_mask)
                        agent0 = self._recur_fast(sparse, hidden_c.to(lat.dtype), comm_c)
                    else:
                        agent0 = self._recur_fast(_SparseRows(lat, rows.rows, (T, B, N)), hidden.to(lat.dtype), comm_mask)
                else:
                    latent_t = self.encode(obs_t.view(T * B * N, *OBS_SHAPE)).view(T, B, N, ENC_FEATURES)
                    agent0 = self._recur_fast(latent_t, hidden.to(latent_t.dtype), comm_mask)
            else:
                latent = self.encode(obs.reshape(B * T * N, *OBS_SHAPE)).view(B, T, N, ENC_FEATURES)
                hidden = hidden.to(latent.dtype)
                agent0 = []
                for t in range(T):
                    hidden = self.recurrent(latent[:, t].reshape(B * N, ENC_FEATURES), hidden)
                    hidden = self.comm(hidden.view(B, N, self.latent_dim), comm_mask[:, t])
                    agent0.append(hidden[:, 0])                      # only agent 0's state is learned from (:248)
                    hidden = hidden.reshape(B * N, self.latent_dim)
                agent0 = torch.stack(agent0, dim=1)                   # [B, T, 256]
            sel = agent0[torch.arange(B, device=obs.device), steps.to(obs.device) - 1]
            q = self.q_head(sel)
        return q.float()

    def _input_proj_weight(self, inplace=False):
        from .fused import PackedRecurrence

        if self._packed_recur is None:
            self._packed_recur = PackedRecurrence()
        return self._packed_recur.input_weight_packed(self, inplace=inplace)

    def _recur_kernel(self, latent_t, hidden, comm_t, want_agent0, gi=None):
        """latent_t bf16 [T, E, N, 784] (time-major); hidden [E*N, 256] or None; comm_t bool [T, E, N, N]
        -> (hidden bf16 [E, N, 256], agent-0 states [T, E, 256] or None) through mapf_recurrent_infer.
        gi: the input projection of every row if the caller has it (fused.LatentCache keeps it beside the latents)."""
        from .fused import PackedRecurrence, input_proj_rows, recurrent_infer

        if self._packed_recur is None:
            self._packed_recur = PackedRecurrence()
        w, b = self._packed_recur.get(self)
        T, E, N, _ = latent_t.shape
        from .fused import mm_rows  # (row-chunked: see there why one big GEMM call is avoided)

        if gi is not None:
            gi = gi.view(T, E, N, 768)
        elif isinstance(latent_t, _SparseRows):  # the input projection of the reachable rows only (its bias is added in the kernel)
            w_ih = self._packed_recur.input_weight(self)
            gi = latent_t.project(lambda x: mm_rows(x, w_ih), 768)
        elif T == 1 and latent_t.is_contiguous():
            # the actor's step: the kernel the latent cache uses for the rows that changed (csrc/mapf_inproj.hip), here on every row --
            # the cached and the plain path stay bit-identical
            gi = input_proj_rows(latent_t.view(E * N, ENC_FEATURES), self._input_proj_weight()[0]).view(T, E, N, 768)
        else:
            w_ih = self._packed_recur.input_weight(self)
            gi = mm_rows(latent_t.reshape(T * E * N, ENC_FEATURES), w_ih).view(T, E, N, 768)
        h0 = None if hidden is None else hidden.reshape(E, N, self.latent_dim)
        return recurrent_infer(gi, h0, comm_t, w, b, want_agent0)

    def _recur_fast(self, latent_t, hidden, comm_mask):
        """The T-step GRU + CommBlock recurrence of `bootstrap` (model.py:242-249), same math as the module path,
        arranged for the GPU:
          * the GRU's input projection W_ih x_t does not depend on the recurrence -> ONE GEMM over all T steps;
          * W_Q, W_K, W_V share their input -> one [384, 256] GEMM per communication round;
          * every weight is cast to bf16 once per call (not once per step);
          * with autograd, the per-step weight gradients are deferred to one GEMM per weight (_WGradSink).
        latent_t bf16 [T, B, N, 784] (time-major); hidden bf16 [B*N, 256]; comm_mask bool [B, T, N, N] -> agent-0 states [B, T, 256]."""
        T, B, N, _ = latent_t.shape
        D, H, A = self.latent_dim, NUM_COMM_HEADS, self.comm.output_dim
        lp = torch.bfloat16
        grad = torch.is_grad_enabled()
        if not grad and self.FUSED_RECURRENCE and N <= RECUR_MAX_AGENTS:  # target network: all T steps in one kernel launch
            a0 = self._recur_kernel(latent_t, hidden, comm_mask.transpose(0, 1), True)[1]
            return a0.transpose(0, 1)
        if grad and self.FUSED_BPTT and N <= BPTT_MAX_AGENTS:
            # online network: forward-with-saved-state and backward-through-time kernels, one workgroup per window
            from .fused import PackedRecurrence, recurrence_params, recurrent_train

            if self._packed_recur is None:
                self._packed_recur = PackedRecurrence()
            w, b = self._packed_recur.get(self)
            if isinstance(latent_t, _SparseRows):
                gi = latent_t.project(lambda x: _InputProj.apply(x, self.recurrent.weight_ih, None), 3 * D)
            else:
                gi = _InputProj.apply(latent_t.view(T * B * N, ENC_FEATURES), self.recurrent.weight_ih, None).view(T, B, N, 3 * D)  # bias: in the kernel
            a0 = recurrent_train(gi, hidden.reshape(B, N, D), comm_mask.transpose(0, 1), w, b, recurrence_params(self))
            return a0.transpose(0, 1)
        if isinstance(latent_t, _SparseRows):  # the PyTorch-level recurrence (more than 128 agents) takes dense latents
            latent_t = latent_t.dense()
        sink = _WGradSink() if grad else None
        rc, at, uc = self.recurrent, self.comm.self_attn, self.comm.update_cell

        def cast(*ps):
            return torch.cat([p.detach() for p in ps], dim=0).to(lp) if len(ps) > 1 else ps[0].detach().to(lp)

        w_hh, w_qkv, b_qkv, w_o = cast(rc.weight_hh), cast(at.W_Q.weight, at.W_K.weight, at.W_V.weight), \
            cast(at.W_Q.bias, at.W_K.bias, at.W_V.bias), cast(at.W_O.weight)
        u_ih, u_hh = cast(uc.weight_ih), cast(uc.weight_hh)
        HA = H * A
        keys = {
            "hh": [(rc.weight_hh, rc.bias_hh, 0, 3 * D)],
            "qkv": [(at.W_Q.weight, at.W_Q.bias, 0, HA), (at.W_K.weight, at.W_K.bias, HA, 2 * HA), (at.W_V.weight, at.W_V.bias, 2 * HA, 3 * HA)],
            "o": [(at.W_O.weight, None, 0, A)],
            "uih": [(uc.weight_ih, uc.bias_ih, 0, 3 * D)],
            "uhh": [(uc.weight_hh, uc.bias_hh, 0, 3 * D)],
        }

        def lin(x, w, b, name):
            if grad:
                return _TimeLinear.apply(x, w, b, sink, keys[name], keys[name][0][0])
            return F.linear(x, w, b)

        # all T input projections at once, time-major so that step t is a contiguous [B*N, 768] slab
        lat_t = latent_t.view(T, B * N, ENC_FEATURES)
        # biases are added by the linears (the GRU cell gets none): with autograd their gradients are then ONE deferred
        # column sum per bias instead of a reduction per (step, cell) inside the fused cell's backward (96 launches)
        b_hh, ub_ih, ub_hh = cast(rc.bias_hh), cast(uc.bias_ih), cast(uc.bias_hh)
        gi_all = _InputProj.apply(lat_t.view(T * B * N, ENC_FEATURES), rc.weight_ih, rc.bias_ih).view(T, B * N, 3 * D) if grad else \
            F.linear(lat_t, cast(rc.weight_ih), cast(rc.bias_ih))
        blocked = (~comm_mask).unsqueeze(2)                              # [B, T, 1, N, N]
        allowed = _sdpa_mask(comm_mask).unsqueeze(2)                     # rows without any partner attend to themselves
        update = (comm_mask.sum(dim=-1) > 1).unsqueeze(-1)               # [B, T, N, 1]  (model.py:103)
        scale = 1.0 / (A ** 0.5)
        agent0 = []
        for t in range(T):
            gh = lin(hidden, w_hh, b_hh, "hh")
            hidden = torch.ops.aten._thnn_fused_gru_cell(gi_all[t], gh, hidden)[0]
            for _ in range(self.comm.num_layers):
                qkv = lin(hidden, w_qkv, b_qkv, "qkv").view(B, N, 3, H, A)
                q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
                if self.SDPA:
                    # one fused kernel: scores and softmax accumulate in fp32 from the bf16 q/k (model.py:75-78 keeps them
                    # in fp32 too); no row of `allowed` is fully masked (_sdpa_mask), so -inf == the reference's -1e9
                    ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=allowed[:, t]).transpose(1, 2).reshape(B * N, HA)
                else:
                    scores = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale   # fp32 (model.py:75-78)
                    attn = F.softmax(scores.masked_fill(blocked[:, t], -1e9), dim=-1)
                    ctx = torch.matmul(attn.to(lp), v).transpose(1, 2).reshape(B * N, HA)
                info = lin(ctx, w_o, None, "o")
                gi = lin(info, u_ih, ub_ih, "uih")
                gh = lin(hidden, u_hh, ub_hh, "uhh")
                new = torch.ops.aten._thnn_fused_gru_cell(gi, gh, hidden)[0]
                hidden = _Select.apply(update[:, t], new.view(B, N, D), hidden.view(B, N, D)).reshape(B * N, D)
            agent0.append(hidden.view(B, N, D)[:, 0])                    # only agent 0's state is learned from (:248)
        return torch.stack(agent0, dim=1)


def load_reference_checkpoint(net: Network, path: str, map_location="cpu"):
    """Loads a `.pth` produced by the reference (worker.py:338) or by this package: same key names."""
    sd = torch.load(path, map_location=map_location)
    net.load_state_dict(sd)
    return net


def num_parameters(net: nn.Module) -> int:
    return sum(p.numel() for p in net.parameters())
