"""Python side of the hand-written DQN kernels (C ABI: include/mapf_dqn.h; kernels: csrc/mapf_encoder.hip, mapf_wgrad.hip,
mapf_wgrad0.hip, mapf_recur.hip, mapf_recur_bwd.hip, mapf_recur_wide.hip, mapf_recur_wide_bwd.hip, mapf_dqn.hip), used by
`mapf_rl_amd.model.Network` on a HIP device under bf16 autocast:

  * `PackedEncoder`, `encoder_forward`            the 8-layer observation encoder as ONE inference kernel (actor, target network);
  * `encoder_forward_train` / `_EncoderTrain`     the same forward saving layer outputs + sign bits, the backward-data chain in one
                                                  kernel, the weight-gradient kernels (3x3 layers, conv0) and the bias partials;
  * `comm_mask`                                   reference model.py:195-208 (FOV square AND 3 nearest) + the replay's packed row;
  * `PackedRecurrence`, `recurrent_infer`         GRU cell + 2 communication rounds for T steps in one launch, N <= 128 agents
                                                  (<= 48: everything of an environment in LDS; 49..128: the wide kernels);
  * `recurrent_train` / `_RecurTrain`             forward-with-saved-state + backward-through-time kernels with the tall
                                                  weight-gradient GEMMs and bias column sums formed here;
  * `bias_res_relu` / `_BiasResReLU`              fused bias + residual + ReLU epilogue of the layer-by-layer path (kept for
                                                  fp32 / FUSED_TRAINING = False runs and as the comparison path of the tests).
Nothing here falls back to a CPU implementation: without the HIP library the import of `._lib` fails."""
import contextlib
import ctypes

import torch


def warm_up_gemm_library(stream):
    """One tiny library GEMM of each kind the captured sequences contain, on `stream`, BEFORE its first capture: hipBLASLt creates its
    handle / workspace at the first call and refuses to do that while the stream is capturing ('operation not permitted when stream is
    capturing').  Until round 4 the actors' own projection GEMM had always run first; it is a hand-written kernel now."""
    with torch.cuda.stream(stream):
        dev = stream.device
        a = torch.zeros((64, 64), dtype=torch.bfloat16, device=dev)
        torch.mm(a, a)
        torch.mm(a, a, out_dtype=torch.float32)
        torch.bmm(a.view(1, 64, 64), a.view(1, 64, 64), out_dtype=torch.float32)
        h = a.to(torch.float16)
        torch.mm(h, h, out_dtype=torch.float32)
        torch.mm(a.float(), a.float())
    stream.synchronize()


def capture_mode():
    """Error mode of a stream capture: "global" (PyTorch's default: an unsafe runtime call from ANY thread invalidates the capture) on a
    single rank; "thread_local" once a process group exists -- its watchdog thread queries the events of collectives from outside
    (hipEventQuery), which a global-mode capture on the main thread must not be failed by."""
    import torch.distributed as dist

    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


@contextlib.contextmanager
def no_gc_during_capture():
    """Keeps Python's cyclic garbage collector from running while a stream is being captured.  A collection that happens to fall into
    a capture can destroy objects that own HIP resources (graphs, events, streams of earlier learners / actors): their destructors call
    the runtime, which a capture in global error mode turns into an exception inside a destructor -> abort (seen once in the full test
    suite, in the backward stage's capture).  Reference counting still frees what the capture itself drops."""
    import gc

    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

from ._lib import check, lib


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


MM_ROW_CHUNK = 32768


def row_bucket(n):
    """Allocation size for a buffer of `n` rows: 8-16 sizes per octave (<= 12.5 % slack)."""
    n = max(int(n), 1)
    g = 1 << max(n.bit_length() - 4, 0)
    return -(-n // g) * g


def rows_buffer(lead, rows, tail, dtype, device):
    """torch.empty((*lead, rows, *tail)) backed by an allocation sized for row_bucket(rows).  The learner's row counts (entries that
    can reach agent 0, distinct observations among them) differ from batch to batch; with exact sizes the caching allocator meets a
    new size for its multi-GB blocks every update and keeps growing its reserve (measured at 128 agents with every observation
    encoded: 158 -> 226 GB and a 1.8 s update whenever it had to go to the driver)."""
    lead, tail = tuple(lead), tuple(tail)
    per = 1
    for v in tail:
        per *= v
    L = 1
    for v in lead:
        L *= v
    flat = torch.empty(L * row_bucket(rows) * per, dtype=dtype, device=device)
    return flat[:L * rows * per].view(*lead, rows, *tail)


def mm_rows(x, w, transpose_w=True):
    """x [n, k] @ w^T (w [m, k]; transpose_w=False: x @ w, w [k, m]) in bf16, in chunks of MM_ROW_CHUNK rows.
    Why chunks: for n beyond ~10^5 rows hipBLASLt selects a STREAM-K kernel (`..._SK3_...MT256x256x64`): a persistent grid whose
    workgroups spin-wait on partial tiles of peer workgroups, i.e. it assumes all of them are resident.  Next to other resident
    work -- the learner's second stream, a second process on the GPU -- that is not guaranteed, and two ranks sharing a GPU did hang
    in exactly these GEMMs (round 3, tools/hang_repro.py).  At <= 32,768 rows the library picks plain tiled kernels."""
    n = x.shape[0]
    if not x.is_cuda:
        return torch.mm(x, w.t() if transpose_w else w)
    m = w.shape[0] if transpose_w else w.shape[1]
    out = rows_buffer((), n, (m,), x.dtype, x.device)
    if n <= MM_ROW_CHUNK:
        if n == 0:
            return out
        # the learner's row counts differ from batch to batch, and every new GEMM shape costs a heuristic lookup in the library
        # (~90 us of host time per call, measured at 6 agents where the update is host-bound): when `x` sits in a bucket-sized
        # buffer (rows_buffer), multiply the whole bucket -- a handful of shapes; the rows past n hold garbage on both sides
        nb = row_bucket(n)
        if nb > n and x.is_contiguous() and nb <= MM_ROW_CHUNK and \
                x.untyped_storage().nbytes() >= (x.storage_offset() + nb * x.shape[1]) * x.element_size():
            torch.mm(torch.as_strided(x, (nb, x.shape[1]), (x.shape[1], 1)), w.t() if transpose_w else w, out=torch.as_strided(out, (nb, m), (m, 1)))
            return out
        return torch.mm(x, w.t() if transpose_w else w, out=out)
    parts = -(-n // MM_ROW_CHUNK)
    step = -(-n // parts)
    for i in range(0, n, step):
        torch.mm(x[i:i + step], w.t() if transpose_w else w, out=out[i:i + step])
    return out


# ---- this library's own dense products / reductions for the learner (csrc/mapf_gemm.hip) ----
_TALL_WS = {}  # (device index, stream) -> workspace f32: launches on one stream are serial, streams must not share


def _tall_ws(dev):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _TALL_WS.get(key)
    if ws is None:
        # per output slab at most 2 MiB of partial slabs + one (mapf_tall_tn_plan); the widest output here, 768 x 784, has 42 slabs: 128 MiB
        ws = _TALL_WS[key] = torch.empty(32 << 20, dtype=torch.float32, device=dev)
    return ws


def prepare_tall_ws(dev, streams):
    """Allocates the workspaces of `streams` now -- outside any capture: a workspace first touched while a stream is capturing would come
    from the capture's private pool."""
    for s in streams:
        if s is not None:
            with torch.cuda.stream(s):
                _tall_ws(dev)


def tall_tn_into(out, a, b, scale=None, accumulate=False):
    """out[m, n] (f32, contiguous) = [out +] scale * a^T b for 16-bit a [K, m], b [K, n] (rows may be strided views) with K in the
    10^4 .. 10^6 range: mapf_tall_tn -- split over K, fp32 partial slabs summed in partition order by a second small launch.
    scale: the int32 [2] tensor of mapf_encoder_backward (bits of 1 / loss scale at [1]) or None."""
    K, m = a.shape
    n = b.shape[1]
    assert b.shape[0] == K and out.shape == (m, n) and out.is_contiguous() and out.dtype == torch.float32 and a.dtype == b.dtype
    assert a.stride(1) == 1 and b.stride(1) == 1
    ws = _tall_ws(a.device)
    check(lib.mapf_tall_tn(_ptr(a), a.stride(0) if K > 1 else m, _ptr(b), b.stride(0) if K > 1 else n, K, m, n, int(a.dtype == torch.float16), _ptr(out),
                           _ptr(scale), int(accumulate), _ptr(ws), ws.numel(), _stream(a.device)), "mapf_tall_tn")
    return out


def sum_parts_into(outs, parts, scale=None):
    """outs[g] (f32, n elements each) = scale * parts[g].sum(0) for up to 8 (parts [P, ...], out) pairs in one launch."""
    P = parts[0].shape[0]
    n = outs[0].numel()
    assert all(p.shape[0] == P and p.numel() == P * n and p.is_contiguous() and o.numel() == n and o.is_contiguous() for p, o in zip(parts, outs))
    check(lib.mapf_sum_parts((ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts]), (ctypes.c_void_p * len(outs))(*[o.data_ptr() for o in outs]),
                             len(parts), P, n, _ptr(scale), _stream(outs[0].device)), "mapf_sum_parts")


LATGRAD_PACKED_ELEMS = 602112


def latent_grad_rows(d_gi, packed_wt, out=None):
    """g_lat [rows, 784] bf16 = d_gi [rows, 768] @ W_ih (mapf_latent_grad_rows; packed_wt from mapf_latent_grad_pack)."""
    rows = d_gi.shape[0]
    assert d_gi.dtype == torch.bfloat16 and d_gi.is_contiguous() and d_gi.shape[1] == 768
    if out is None:
        out = rows_buffer((), rows, (784,), torch.bfloat16, d_gi.device)
    check(lib.mapf_latent_grad_rows(_ptr(d_gi), rows, _ptr(packed_wt), _ptr(out), _stream(d_gi.device)), "mapf_latent_grad_rows")
    return out


class _BiasResReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, res):
        C = y.shape[1]
        check(lib.mapf_bias_res_relu_fwd(_ptr(y), _ptr(bias), _ptr(res), y.numel(), C, _stream(y.device)), "mapf_bias_res_relu_fwd")
        ctx.mark_dirty(y)
        ctx.save_for_backward(y)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        C = y.shape[1]
        g = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(y, memory_format=torch.channels_last)
        gb = torch.zeros(C, dtype=torch.float32, device=y.device)
        check(lib.mapf_bias_res_relu_bwd(_ptr(g), _ptr(y), _ptr(gx), _ptr(gb), y.numel(), C, _stream(y.device)), "mapf_bias_res_relu_bwd")
        return gx, gb, (gx if ctx.has_res else None)


def bias_res_relu(conv_out, bias, res=None):
    """conv_out: bf16 NHWC activation straight out of a bias-free convolution (modified in place);
    bias: f32 [C]; res: optional bf16 NHWC tensor of the same shape."""
    assert conv_out.dtype == torch.bfloat16 and _is_nhwc(conv_out) and bias.dtype == torch.float32
    if res is not None:
        assert res.dtype == torch.bfloat16 and _is_nhwc(res) and res.shape == conv_out.shape
    return _BiasResReLU.apply(conv_out, bias.contiguous(), res)


# ---------------------------------------------------------------------------------------------------------
# Fused inference encoder (include/mapf_dqn.h: mapf_encoder_pack / mapf_encoder_forward)
# ---------------------------------------------------------------------------------------------------------
ENC_PACKED_ELEMS = 894976
ENC_BIAS_ELEMS = 912
ENC_PACKED_BWD_ELEMS = 888832
ENC_WGRAD0_PARTS = 512
ENC_WGRAD_PARTS = 128
ENC_OBS_PER_BLOCK = 4
ENC_PACKED_OBS_STRIDE = 488  # include/mapf_dqn.h: MAPF_ENC_PACKED_OBS_STRIDE
_ENC_OBS_U8, _ENC_OBS_BF16 = 0, 1
# what the encoder kernels keep between layers and save for the backward pass: IEEE half (csrc/mapf_encoder.hip) -- activations,
# packed weights, pre-activation gradients.  The latent they output and the gradient they receive for it are bf16.
ENC_ELEMENT = torch.float16


def tall_tn_f32(a, b, rows=8192):
    """a^T b in fp32 for 16-bit a [K, m], b [K, n] with K in the 10^5..10^6 range (model._tall_tn: K split into batches so that the
    library GEMM fills the chip) -- with fp32 partial products: f16 operands carry a loss scale, a 16-bit partial could overflow."""
    K, m = a.shape
    S = K // rows
    if S > 1:
        out = torch.bmm(a[:S * rows].view(S, rows, m).transpose(1, 2), b[:S * rows].view(S, rows, -1), out_dtype=torch.float32).sum(dim=0)
        if K > S * rows:
            out += torch.mm(a[S * rows:].t(), b[S * rows:], out_dtype=torch.float32)
        return out
    return torch.mm(a.t(), b, out_dtype=torch.float32)


def encoder_convs(obs_encoder):
    """The 8 convolutions of `Network.obs_encoder` (reference model.py:147-162) in the order the packer expects."""
    e = obs_encoder
    return [e[0], e[2].block1, e[2].block2, e[3].block1, e[3].block2, e[4].block1, e[4].block2, e[5]]


def _conv_weight_ptrs(convs):
    """(tensors kept alive, nhwc flag, host array of 8 device pointers) for the pack kernels: fp32 weights are used where they
    lie -- PyTorch's channels_last memory ([co][kh][kw][ci]: how Network stores them) or plain contiguous --, anything else is copied."""
    ws = [c.weight.detach() for c in convs]
    if all(w.dtype == torch.float32 and w.is_contiguous(memory_format=torch.channels_last) for w in ws):
        nhwc = 1
    else:
        nhwc = 0
        ws = [w.to(torch.float32).contiguous() for w in ws]
    return ws, nhwc, (ctypes.c_void_p * 8)(*[w.data_ptr() for w in ws])


class PackedEncoder:
    """bf16 MFMA-fragment image of an encoder's weights; re-packed (one small kernel) whenever a parameter changed
    (optimizer step, load_state_dict, .to()), detected through the tensors' version counters and storage pointers and the owner's
    `epoch` (the learner's fused Adam kernel writes the parameters without passing through a PyTorch operation)."""

    def __init__(self):
        self.key = None
        self.weights = None
        self.bias = None

    def get(self, obs_encoder, epoch=0, force=False):
        """force: pack even if the key says nothing changed (a launch sequence being captured into a graph must contain the pack)."""
        convs = encoder_convs(obs_encoder)
        params = [c.weight for c in convs] + [c.bias for c in convs]
        key = (epoch,) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self.key or force:
            dev = params[0].device
            assert dev.type == "cuda", "the fused encoder needs a HIP device"
            ws, nhwc, wp = _conv_weight_ptrs(convs)
            bs = [c.bias.detach().to(torch.float32).contiguous() for c in convs]
            assert [tuple(w.shape) for w in ws] == [(128, 6, 3, 3)] + [(128, 128, 3, 3)] * 6 + [(16, 128, 1, 1)]
            if self.weights is None or self.weights.device != dev:
                self.weights = torch.empty(ENC_PACKED_ELEMS, dtype=ENC_ELEMENT, device=dev)
                self.bias = torch.empty(ENC_BIAS_ELEMS, dtype=torch.float32, device=dev)
            bp = (ctypes.c_void_p * 8)(*[b.data_ptr() for b in bs])
            check(lib.mapf_encoder_pack(wp, bp, nhwc, _ptr(self.weights), _ptr(self.bias), _stream(dev)), "mapf_encoder_pack")
            self.key = key  # `ws`/`bs` temporaries stay alive until the stream has consumed them (caching allocator)
        return self.weights, self.bias


def pack_encoder_backward(obs_encoder_or_params):
    """bf16 image of the TRANSPOSED convolutions for mapf_encoder_backward (one kernel; no copies for channels_last fp32 weights)."""
    if isinstance(obs_encoder_or_params, (list, tuple)):
        class _C:  # the 8 weights of a saved parameter list [w0, b0, w1, b1, ...]
            def __init__(self, w):
                self.weight = w
        convs = [_C(obs_encoder_or_params[2 * i]) for i in range(8)]
    else:
        convs = encoder_convs(obs_encoder_or_params)
    ws, nhwc, wp = _conv_weight_ptrs(convs)
    dev = ws[0].device
    wpt = torch.empty(ENC_PACKED_BWD_ELEMS, dtype=ENC_ELEMENT, device=dev)
    check(lib.mapf_encoder_pack_bwd(wp, nhwc, _ptr(wpt), _stream(dev)), "mapf_encoder_pack_bwd")
    return wpt


def encoder_forward(obs, packed_weights, packed_bias):
    """obs [M, 6, 9, 9] uint8 / bool / bf16 on a HIP device -> latent bf16 [M, 784]; no autograd."""
    assert obs.dim() == 4 and tuple(obs.shape[1:]) == (6, 9, 9) and obs.is_cuda
    if obs.dtype == torch.bool:
        obs = obs.view(torch.uint8)
    if obs.dtype not in (torch.uint8, torch.bfloat16):
        obs = obs.to(torch.bfloat16)
    obs = obs.contiguous()
    M = obs.shape[0]
    out = torch.empty((M, 784), dtype=torch.bfloat16, device=obs.device)
    kind = _ENC_OBS_U8 if obs.dtype == torch.uint8 else _ENC_OBS_BF16
    check(lib.mapf_encoder_forward(_ptr(obs), kind, M, _ptr(packed_weights), _ptr(packed_bias), _ptr(out), _stream(obs.device)),
          "mapf_encoder_forward")
    return out


class LatentCache:
    """Exact reuse in the actor loop: an agent whose 6x9x9 observation is the same as at the previous step has the same encoding (the
    encoder is a deterministic per-observation function, reference model.py:147-162), so only the rows that changed go through the
    encoder kernel -- 34 % repeat under the bench's tape policy, far more once agents wait on their goals
    (tools/obs_reuse_probe.py).  Valid while the encoder's weights do not change (checked per call: parameter versions + the
    owner's `weights_epoch`) and the caller passes the same observation buffer; otherwise everything is encoded again.
    No host synchronisation: the list of changed rows and its length stay on the device (mapf_obs_changed,
    mapf_encoder_forward_rows)."""

    def __init__(self):
        self.key = None
        self.lat = self.prev = self.packed = self.list = self.count = self.gi = None
        self.calls = self.full = 0

    def encode(self, obs, packed: "PackedEncoder", obs_encoder, epoch=0, proj=None):
        """obs uint8 [R, 6, 9, 9] (contiguous; the SAME buffer at every call) -> latent bf16 [R, 784] (owned by the cache).
        proj = (packed W_ih, its key) (PackedRecurrence.input_weight_packed): the cache also keeps the recurrent cell's input projection
        of every row, self.gi bf16 [R, 768], recomputed for the same rows as the latent (an unchanged latent has an unchanged projection)."""
        assert obs.dtype == torch.uint8 and obs.is_contiguous() and tuple(obs.shape[1:]) == (6, 9, 9)
        R, dev = obs.shape[0], obs.device
        wp, bp = packed.get(obs_encoder, epoch)
        key = (packed.key, obs.data_ptr(), R, None if proj is None else proj[1])
        st = _stream(dev)
        self.calls += 1
        if key != self.key:
            if self.lat is None or self.lat.shape[0] != R or self.lat.device != dev:
                self.lat = torch.empty((R, 784), dtype=torch.bfloat16, device=dev)
                self.prev = torch.empty_like(obs)
                self.packed = torch.empty((R, ENC_PACKED_OBS_STRIDE), dtype=torch.uint8, device=dev)  # rows padded to dwords
                self.list = torch.empty(R, dtype=torch.int32, device=dev)
                self.count = torch.zeros(1, dtype=torch.int32, device=dev)
            check(lib.mapf_encoder_forward(_ptr(obs), _ENC_OBS_U8, R, _ptr(wp), _ptr(bp), _ptr(self.lat), st), "mapf_encoder_forward")
            self.prev.copy_(obs)
            self.count.fill_(R)
            if proj is not None:
                if self.gi is None or self.gi.shape[0] != R or self.gi.device != dev:
                    self.gi = torch.empty((R, 768), dtype=torch.bfloat16, device=dev)
                input_proj_rows(self.lat, proj[0], self.gi)
            self.key = key
            self.full += 1
        else:
            check(lib.mapf_obs_changed(_ptr(obs), _ptr(self.prev), R, _ptr(self.list), _ptr(self.count), _ptr(self.packed), st), "mapf_obs_changed")
            check(lib.mapf_encoder_forward_rows(_ptr(self.packed), R, _ptr(self.list), _ptr(self.count), _ptr(wp), _ptr(bp), _ptr(self.lat), st),
                  "mapf_encoder_forward_rows")
            if proj is not None:
                input_proj_rows(self.lat, proj[0], self.gi, self.list, self.count)
        return self.lat

    def last_encoded(self):
        """Rows the last call encoded (reads the device counter: synchronises; statistics only)."""
        return int(self.count.item()) if self.count is not None else 0


def window_relevance(comm_mask_bt, steps):
    """comm_mask_bt bool/uint8 [B, T, N, N], steps int64 [B] (1-based) on a HIP device -> bool [T, B, N]:
    include/mapf_dqn.h mapf_window_relevance (the entries of a training window that can reach agent 0's Q-value)."""
    assert comm_mask_bt.is_cuda and comm_mask_bt.dim() == 4 and comm_mask_bt.dtype in (torch.bool, torch.uint8)
    m = comm_mask_bt.contiguous()
    B, T, N, _ = m.shape
    st = steps.to(device=m.device, dtype=torch.int64).contiguous()
    rel = torch.empty((T, B, N), dtype=torch.bool, device=m.device)
    check(lib.mapf_window_relevance(_ptr(m), _ptr(st), T, B, N, _ptr(rel), _stream(m.device)), "mapf_window_relevance")
    return rel


def comm_mask(pos, obs_radius=4, max_comm=3, packed_words=0, out_mask=None, out_packed=None):
    """pos int16 [E, N, 2] on a HIP device -> (bool [E, N, N], int32 [E, N, packed_words] or None):
    include/mapf_dqn.h mapf_comm_mask (reference model.py:195-208).  out_mask (bool / uint8, E*N*N elements) / out_packed (int32,
    E*N*packed_words): write into the caller's buffers."""
    assert pos.is_cuda and pos.dtype == torch.int16 and pos.dim() == 3 and pos.shape[2] == 2
    pos = pos.contiguous()
    E, N, _ = pos.shape
    if out_mask is not None:
        assert out_mask.is_contiguous() and out_mask.numel() == E * N * N and out_mask.element_size() == 1
        mask = out_mask.view(torch.bool).view(E, N, N)
    else:
        mask = torch.empty((E, N, N), dtype=torch.bool, device=pos.device)
    if out_packed is not None:
        assert packed_words and out_packed.is_contiguous() and out_packed.dtype == torch.int32 and out_packed.numel() == E * N * packed_words
        packed = out_packed.view(E, N, packed_words)
    else:
        packed = torch.empty((E, N, packed_words), dtype=torch.int32, device=pos.device) if packed_words else None
    check(lib.mapf_comm_mask(_ptr(pos), E, N, obs_radius, max_comm, _ptr(mask), _ptr(packed), packed_words, _stream(pos.device)),
          "mapf_comm_mask")
    return mask, packed


# ---------------------------------------------------------------------------------------------------------
# Training forward of the encoder through the fused kernel (it also stores the 7 layer outputs); the backward-data
# chain is one more kernel of the same structure (mapf_encoder_backward) that starts from the gradient of the latent and
# emits the ReLU-masked pre-activation gradient of every layer; the weight gradients are a third kernel
# (mapf_encoder_wgrad, per 3x3 128->128 layer, on (layer input, that gradient)) and a fourth for conv0 (mapf_encoder_wgrad0).
# ---------------------------------------------------------------------------------------------------------
class _EncoderTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, packed_w, packed_b, *params):
        M = obs.shape[0]
        acts = torch.empty((7, M, 7, 7, 128), dtype=ENC_ELEMENT, device=obs.device)
        out = torch.empty((M, 784), dtype=torch.bfloat16, device=obs.device)
        bits = torch.empty((7, M, 49, 4), dtype=torch.int32, device=obs.device)  # ReLU sign bits for the backward chain
        kind = _ENC_OBS_U8 if obs.dtype == torch.uint8 else _ENC_OBS_BF16
        check(lib.mapf_encoder_forward_save(_ptr(obs), kind, M, _ptr(packed_w), _ptr(packed_b), _ptr(out), _ptr(acts),
                                            _ptr(bits), _stream(obs.device)), "mapf_encoder_forward_save")
        ctx.save_for_backward(obs, acts, out, bits, *params)
        return out

    @staticmethod
    def backward(ctx, g):
        obs, acts, out, bits = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        cl = torch.channels_last
        dev = obs.device
        M = obs.shape[0]
        gws = [None] * 8
        # the whole backward-data chain in one kernel: ReLU mask of the 1x1 layer on the incoming gradient, then the masked
        # pre-activation gradients of all 7 layers below it and per-workgroup bias-gradient partials
        wpt = pack_encoder_backward(list(params))
        g = g.to(torch.bfloat16).contiguous()
        nblk = -(-M // ENC_OBS_PER_BLOCK)
        gz = torch.empty_like(acts)
        gz7 = torch.empty((M * 49, 16), dtype=ENC_ELEMENT, device=dev)
        gb_part = torch.empty((7, nblk, 128), dtype=torch.float32, device=dev)
        gb7_part = torch.empty((4 * nblk, 16), dtype=torch.float32, device=dev)
        # gz / gz7 come back multiplied by the chain's loss scale S (include/mapf_dqn.h); scale[1] = the bits of 1 / S
        scale = torch.empty(2, dtype=torch.int32, device=dev)
        check(lib.mapf_encoder_backward(_ptr(g), _ptr(out), M, _ptr(bits), _ptr(wpt), _ptr(gz), _ptr(gb_part), _ptr(gz7), _ptr(gb7_part),
                                        _ptr(scale), _stream(dev)), "mapf_encoder_backward")
        # per-workgroup partial bias gradients -> [7][128]: in two stages (a reduction over the middle axis of [7, 30720, 128] in
        # one call runs at 0.4 TB/s; the 240-row inner stage is contiguous per output and the outer stage tiny)
        pad = (-nblk) % 256
        gp = gb_part if pad == 0 else torch.cat([gb_part, gb_part.new_zeros((7, pad, 128))], dim=1)
        gbs = list(gp.view(7, -1, 256, 128).sum(dim=2).sum(dim=1).unbind(0)) + [gb7_part.sum(dim=0)]
        # weight gradients of the six 3x3 128->128 layers: one streaming MFMA kernel per layer (mapf_encoder_wgrad),
        # partial sums per observation partition, added here
        ws = torch.empty((ENC_WGRAD_PARTS, 128, 3, 3, 128), dtype=torch.float32, device=dev)
        for k in range(1, 7):
            check(lib.mapf_encoder_wgrad(_ptr(gz[k]), _ptr(acts[k - 1]), M, _ptr(scale), _ptr(ws), _stream(dev)), "mapf_encoder_wgrad")
            gws[k] = ws.sum(dim=0).permute(0, 3, 1, 2)  # [co][ky][kx][ci] memory == channels_last [co, ci, 3, 3]
        # conv0 (6 -> 128 on the raw 9x9 observation): its own streaming kernel (csrc/mapf_wgrad0.hip) -- the im2col matrix a
        # library GEMM would need is 0.65 GB at the learner's shape; the 1x1 layer (16 outputs) is a plain split-K GEMM
        ws0 = torch.empty((ENC_WGRAD0_PARTS, 128, 64), dtype=torch.float32, device=dev)
        kind = _ENC_OBS_U8 if obs.dtype == torch.uint8 else _ENC_OBS_BF16
        check(lib.mapf_encoder_wgrad0(_ptr(gz[0]), _ptr(obs), kind, M, _ptr(scale), _ptr(ws0), _stream(dev)), "mapf_encoder_wgrad0")
        gws[0] = ws0.sum(dim=0)[:, :54].reshape(128, 6, 3, 3)

        # the 1x1 head (16 x 128 outputs over 6 M rows): the library's split-K GEMM streams its 1.7 GB of input at 5.3-5.9 TB/s
        # (0.33 ms at 40 agents); a hand-written vector-ALU streaming kernel tried in round 2 reached 3.2-3.8 TB/s and was dropped
        gws[7] = (tall_tn_f32(gz7, acts[6].reshape(M * 49, 128)) * scale.view(torch.float32)[1]).view(16, 128, 1, 1)
        grads = []
        for i in range(8):
            grads += [gws[i].to(params[2 * i].dtype), gbs[i].to(params[2 * i + 1].dtype)]
        return (None, None, None, *grads)


def encoder_forward_train(obs, obs_encoder, packed: PackedEncoder, epoch=0):
    """obs [M, 6, 9, 9] uint8 / bool / bf16 -> latent bf16 [M, 784] with autograd into the encoder's parameters."""
    assert obs.dim() == 4 and tuple(obs.shape[1:]) == (6, 9, 9) and obs.is_cuda
    if obs.dtype == torch.bool:
        obs = obs.view(torch.uint8)
    if obs.dtype not in (torch.uint8, torch.bfloat16):
        obs = obs.to(torch.bfloat16)
    wp, bp = packed.get(obs_encoder, epoch)
    params = [t for c in encoder_convs(obs_encoder) for t in (c.weight, c.bias)]
    return _EncoderTrain.apply(obs.contiguous(), wp, bp, *params)


# ---------------------------------------------------------------------------------------------------------
# Inference recurrence behind the encoder (include/mapf_dqn.h: mapf_recurrent_infer)
# ---------------------------------------------------------------------------------------------------------
RECUR_MAX_AGENTS = 128         # include/mapf_dqn.h: the fused recurrence kernels (inference, forward-save, backward)
RECUR_NARROW_AGENTS = 48       # up to here one workgroup keeps every image in LDS and the forward saves the attention weights P


def _recurrence_param_ptrs(params):
    """(tensors kept alive, host array of 14 device pointers) of `recurrence_params(net)` as fp32 contiguous tensors."""
    ts = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().to(torch.float32).contiguous() for p in params]
    return ts, (ctypes.c_void_p * 14)(*[t.data_ptr() for t in ts])


class PackedRecurrence:
    """bf16 fragment image + f32 biases of Network.recurrent / Network.comm for mapf_recurrent_infer (one kernel:
    mapf_recurrent_pack); rebuilt when a parameter changed (same rule as PackedEncoder)."""

    def __init__(self):
        self.key = None
        self.weights = None
        self.bias = None

    def get(self, net, inplace=False, force=False):
        """inplace: re-pack into the buffers of the last pack (their address is held by a captured graph; the caller packs on the stream
        that reads them).  force: pack even if the key says nothing changed (a launch sequence being captured must contain the pack)."""
        params = recurrence_params(net)
        key = (getattr(net, "weights_epoch", 0),) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self.key or force:
            dev = params[0].device
            ts, ptrs = _recurrence_param_ptrs(params)
            if not (inplace and self.weights is not None and self.weights.device == dev):
                # fresh buffers per pack: a launch of the stream that packed before may still be reading the old image
                self.weights = torch.empty(RECUR_WEIGHT_ELEMS, dtype=torch.bfloat16, device=dev)
                self.bias = torch.empty(RECUR_BIAS_ELEMS, dtype=torch.float32, device=dev)
            check(lib.mapf_recurrent_pack(ptrs, _ptr(self.weights), _ptr(self.bias), None, _stream(dev)), "mapf_recurrent_pack")
            self.key = key
        return self.weights, self.bias

    def input_weight_packed(self, net, inplace=False):
        """(recurrent.weight_ih as bf16 MFMA fragments for input_proj_rows, the key of this pack): mapf_input_proj_pack, repacked when the
        parameter changed; inplace: as in get()."""
        p = net.recurrent.weight_ih
        key = (getattr(net, "weights_epoch", 0), p.data_ptr(), p._version)
        if key != getattr(self, "key_ihp", None):
            if not (inplace and getattr(self, "w_ihp", None) is not None and self.w_ihp.device == p.device):
                self.w_ihp = torch.empty(INPROJ_PACKED_ELEMS, dtype=torch.bfloat16, device=p.device)
            src = p.detach()
            src = src if (src.dtype == torch.float32 and src.is_contiguous()) else src.float().contiguous()
            check(lib.mapf_input_proj_pack(_ptr(src), _ptr(self.w_ihp), _stream(p.device)), "mapf_input_proj_pack")
            self.key_ihp = key
        return self.w_ihp, self.key_ihp

    def input_weight(self, net, inplace=False):
        """bf16 copy of Network.recurrent.weight_ih (the input projection is a library GEMM in front of the kernel); converted when the
        parameter changed, not once per step (it was a 2.4 MB element-wise launch in every actor iteration).  inplace: as in get()."""
        p = net.recurrent.weight_ih
        key = (getattr(net, "weights_epoch", 0), p.data_ptr(), p._version)
        if key != getattr(self, "key_ih", None):
            if inplace and getattr(self, "w_ih", None) is not None and self.w_ih.device == p.device:
                self.w_ih.copy_(p.detach())
            else:
                self.w_ih = p.detach().to(torch.bfloat16)
            self.key_ih = key
        return self.w_ih


RECUR_WEIGHT_ELEMS = 548864
INPROJ_PACKED_ELEMS = 614400


def input_proj_rows(latent, packed_w, out=None, row_list=None, row_count=None):
    """gi bf16 [R, 768] = W_ih latent (no bias) through mapf_input_proj_rows (csrc/mapf_inproj.hip): for all R rows, or -- row_list int32
    [>= count], row_count int32 [1] on the device -- only for the listed ones, the other rows of `out` staying as they are."""
    R = latent.shape[0]
    assert latent.is_cuda and latent.dtype == torch.bfloat16 and latent.is_contiguous() and latent.shape[1] == 784
    assert packed_w.dtype == torch.bfloat16 and packed_w.numel() == INPROJ_PACKED_ELEMS
    if out is None:
        assert row_list is None, "a row list updates an existing gi buffer"
        out = torch.empty((R, 768), dtype=torch.bfloat16, device=latent.device)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and tuple(out.shape) == (R, 768)
    check(lib.mapf_input_proj_rows(_ptr(latent), R, _ptr(row_list), _ptr(row_count), _ptr(packed_w), _ptr(out), _stream(latent.device)),
          "mapf_input_proj_rows")
    return out


RECUR_BIAS_ELEMS = 3456


def recurrent_infer(gi, h0, comm, weights, bias, want_agent0=False, out=None):
    """gi bf16 [T, E, N, 768]; h0 bf16 [E, N, 256] or None; comm bool/u8 [T, E, N, N]
    -> (hidden bf16 [E, N, 256], agent-0 states bf16 [T, E, 256] or None).  out: optional bf16 [E, N, 256] buffer for the hidden states
    (not h0's: a workgroup writes its environment's rows while others may still be reading theirs -- fine -- but callers keep h0)."""
    T, E, N, _ = gi.shape
    assert gi.is_cuda and gi.dtype == torch.bfloat16 and gi.shape[3] == 768 and N <= RECUR_MAX_AGENTS
    assert tuple(comm.shape) == (T, E, N, N)
    gi = gi.contiguous()
    comm = comm.contiguous()
    comm = comm.view(torch.uint8) if comm.dtype == torch.bool else comm.to(torch.uint8)
    if h0 is not None:
        h0 = h0.to(torch.bfloat16).reshape(E, N, 256).contiguous()
    if out is not None:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == E * N * 256
        h_out = out.view(E, N, 256)
    else:
        h_out = torch.empty((E, N, 256), dtype=torch.bfloat16, device=gi.device)
    a0 = torch.empty((T, E, 256), dtype=torch.bfloat16, device=gi.device) if want_agent0 else None
    check(lib.mapf_recurrent_infer(_ptr(gi), _ptr(h0), _ptr(comm), _ptr(weights), _ptr(bias), T, E, N, _ptr(h_out), _ptr(a0), None, 0,
                                   _stream(gi.device)), "mapf_recurrent_infer")
    return h_out, a0


def q_head_infer(hidden, adv, state, q_out=None, act_out=None):
    """Dueling head + arg-max of the policy's forward in one launch (include/mapf_dqn.h: mapf_q_head): hidden bf16 [rows, 256],
    adv / state the network's nn.Linear modules (fp32 parameters) -> (q f32 [rows, 5], actions int64 [rows])."""
    rows = hidden.shape[0]
    assert hidden.is_cuda and hidden.dtype == torch.bfloat16 and hidden.is_contiguous() and hidden.shape[1] == 256
    ps = (adv.weight, adv.bias, state.weight, state.bias)
    assert all(p.dtype == torch.float32 and p.is_contiguous() for p in ps)
    q = torch.empty((rows, 5), dtype=torch.float32, device=hidden.device) if q_out is None else q_out
    act = torch.empty(rows, dtype=torch.int64, device=hidden.device) if act_out is None else act_out
    assert q.dtype == torch.float32 and q.is_contiguous() and q.numel() == rows * 5 and act.dtype == torch.int64 and act.is_contiguous()
    check(lib.mapf_q_head(_ptr(hidden), rows, _ptr(ps[0]), _ptr(ps[1]), _ptr(ps[2]), _ptr(ps[3]), _ptr(q), _ptr(act), _stream(hidden.device)), "mapf_q_head")
    return q, act


def recurrent_infer_multi(gi_all, h0_all, comm_all, weights, bias, envtab, out):
    """One policy step of environments of different agent counts (<= 16 each) in one launch (include/mapf_dqn.h:
    mapf_recurrent_infer_multi): gi_all bf16 [rows, 768], h0_all bf16 [rows, 256] or None, comm_all uint8 [bytes], envtab int32 [E, 4],
    out bf16 [rows, 256] (not h0_all)."""
    assert gi_all.dtype == torch.bfloat16 and gi_all.is_contiguous() and gi_all.shape[1] == 768 and out.dtype == torch.bfloat16 and out.is_contiguous()
    assert envtab.dtype == torch.int32 and envtab.is_contiguous() and envtab.shape[1] == 4 and comm_all.dtype == torch.uint8
    assert h0_all is None or (h0_all.dtype == torch.bfloat16 and h0_all.is_contiguous() and h0_all.data_ptr() != out.data_ptr())
    check(lib.mapf_recurrent_infer_multi(_ptr(gi_all), _ptr(h0_all), _ptr(comm_all), _ptr(weights), _ptr(bias), envtab.shape[0], _ptr(envtab), _ptr(out),
                                         _stream(gi_all.device)), "mapf_recurrent_infer_multi")
    return out


# ---------------------------------------------------------------------------------------------------------
# Training recurrence: forward with saved state + backward through time (mapf_recurrent_forward_save /
# mapf_recurrent_backward); weight and bias gradients are tall GEMMs / column sums formed here.
# ---------------------------------------------------------------------------------------------------------
def _pack_frag(m):
    """[O, K] -> MFMA A-fragment order [O/16][K/32][lane = 16*((k%32)//8) + o%16][k%8], flattened."""
    return m.reshape(m.shape[0] // 16, 16, m.shape[1] // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def recurrence_params(net):
    rc, at, uc = net.recurrent, net.comm.self_attn, net.comm.update_cell
    return [rc.weight_hh, rc.bias_ih, rc.bias_hh, at.W_Q.weight, at.W_K.weight, at.W_V.weight, at.W_Q.bias, at.W_K.bias, at.W_V.bias,
            at.W_O.weight, uc.weight_ih, uc.weight_hh, uc.bias_ih, uc.bias_hh]


def pack_recurrence_transposed(params):
    """The transposed matrices of the backward kernel in fragment order (include/mapf_dqn.h), one kernel."""
    ts, ptrs = _recurrence_param_ptrs(params)
    out = torch.empty(RECUR_WEIGHT_ELEMS, dtype=torch.bfloat16, device=ts[0].device)
    check(lib.mapf_recurrent_pack(ptrs, None, None, _ptr(out), _stream(out.device)), "mapf_recurrent_pack")
    return out


def pack_recurrence_transposed_torch(params):
    """The same image through PyTorch operations (the statement the pack kernel is tested against)."""
    w_hh, _, _, wq, wk, wv, _, _, _, w_o, u_ih, u_hh, _, _ = [p.detach() for p in params]
    parts = [_pack_frag(u_ih[256 * g:256 * (g + 1)].t().contiguous()) for g in range(3)]
    parts += [_pack_frag(u_hh[256 * g:256 * (g + 1)].t().contiguous()) for g in range(3)]
    parts += [_pack_frag(w_hh[256 * g:256 * (g + 1)].t().contiguous()) for g in range(3)]
    parts += [_pack_frag(w_o.t().contiguous()), _pack_frag(torch.cat([wq, wk, wv], dim=0).t().contiguous())]
    out = torch.cat(parts).to(torch.bfloat16).contiguous()
    assert out.numel() == 548864
    return out


def pack_recurrence_torch(params):
    """Forward image + biases through PyTorch operations (tests)."""
    w_hh, b_ih, b_hh, wq, wk, wv, bq, bk, bv, w_o, u_ih, u_hh, ub_ih, ub_hh = [p.detach() for p in params]
    mats = [w_hh, torch.cat([wq, wk, wv], dim=0), w_o, u_ih, u_hh]
    w = torch.cat([_pack_frag(m) for m in mats]).to(torch.bfloat16).contiguous()
    b = torch.cat([t.reshape(-1) for t in (b_ih, b_hh, bq, bk, bv, ub_ih, ub_hh)]).to(torch.float32).contiguous()
    return w, b


class _RecurTrain(torch.autograd.Function):
    """agent-0 states [T, E, 256] of the T-step GRU + CommBlock recurrence from gi = W_ih latent (no bias) [T, E, N, 768]."""

    @staticmethod
    def forward(ctx, gi, h0, comm, w_pack, b_pack, *params):
        T, E, N, _ = gi.shape
        dev, R = gi.device, gi.shape[0] * gi.shape[1] * gi.shape[2]
        bf = torch.bfloat16
        saves = [torch.empty((R, 256), dtype=bf, device=dev), torch.empty((R, 1024), dtype=bf, device=dev),
                 torch.empty((2, R, 256), dtype=bf, device=dev), torch.empty((2, R, 384), dtype=bf, device=dev),
                 torch.empty((2, R, 128), dtype=bf, device=dev), torch.empty((2, R, 64), dtype=bf, device=dev),
                 torch.empty((2, R, 1024), dtype=bf, device=dev),
                 # attention weights: saved for N <= 48; wider environments recompute them in the backward kernel (placeholder)
                 torch.empty((2, T * E, 2, 48, 64) if N <= RECUR_NARROW_AGENTS else (8,), dtype=bf, device=dev)]
        h_out = torch.empty((E, N, 256), dtype=bf, device=dev)
        a0 = torch.empty((T, E, 256), dtype=bf, device=dev)
        sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in saves])
        check(lib.mapf_recurrent_forward_save(_ptr(gi), _ptr(h0), _ptr(comm), _ptr(w_pack), _ptr(b_pack), T, E, N, _ptr(h_out), _ptr(a0), sp,
                                              None, 0, _stream(dev)), "mapf_recurrent_forward_save")
        ctx.save_for_backward(comm, *saves, *params)
        ctx.shape = (T, E, N)
        return a0

    @staticmethod
    def backward(ctx, g_a0):
        from .model import _tall_tn

        T, E, N = ctx.shape
        comm = ctx.saved_tensors[0]
        saves = ctx.saved_tensors[1:9]
        params = ctx.saved_tensors[9:]
        dev, R, bf = comm.device, T * E * N, torch.bfloat16
        wt = pack_recurrence_transposed(params)
        outs = [torch.empty((R, 768), dtype=bf, device=dev), torch.empty((R, 768), dtype=bf, device=dev),
                torch.empty((2, R, 768), dtype=bf, device=dev), torch.empty((2, R, 768), dtype=bf, device=dev),
                torch.empty((2, R, 64), dtype=bf, device=dev), torch.empty((2, R, 384), dtype=bf, device=dev),
                torch.empty((E, 2432), dtype=torch.float32, device=dev)]
        g_a0 = g_a0.to(bf).contiguous()
        sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in saves])
        op = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in outs])
        check(lib.mapf_recurrent_backward(sp, _ptr(comm), _ptr(g_a0), _ptr(wt), T, E, N, op, None, 0, _stream(dev)), "mapf_recurrent_backward")
        d_gi1, d_gh1, d_gi2, d_gh2, d_info, d_qkv, bsum = outs
        hin0, _, hr, _, ctxs, info, _, _ = saves
        d_gi2f, d_gh2f, d_qkvf, hrf = d_gi2.view(2 * R, 768), d_gh2.view(2 * R, 768), d_qkv.view(2 * R, 384), hr.view(2 * R, 256)
        g_whh = _tall_tn(d_gh1, hin0)
        g_qkv = _tall_tn(d_qkvf, hrf)
        bs = bsum.sum(dim=0)  # bias gradients: the kernel's per-environment column sums
        u, r, b_qkv = bs[:1024], bs[1024:2048], bs[2048:]
        grads = [g_whh, r[:768], torch.cat([r[:512], r[768:]]), g_qkv[:128], g_qkv[128:256], g_qkv[256:], b_qkv[:128],
                 b_qkv[128:256], b_qkv[256:], _tall_tn(d_info.view(2 * R, 64), ctxs.view(2 * R, 128)), _tall_tn(d_gi2f, info.view(2 * R, 64)),
                 _tall_tn(d_gh2f, hrf), u[:768], torch.cat([u[:512], u[768:]])]
        grads = [g.to(p.dtype).reshape(p.shape) for g, p in zip(grads, params)]
        return (d_gi1.view(T, E, N, 768), None, None, None, None, *grads)


def recurrent_train(gi, h0, comm, w_pack, b_pack, params):
    """gi bf16 [T, E, N, 768] (autograd input); h0 bf16 [E, N, 256] or None; comm bool/u8 [T, E, N, N] -> agent-0 states [T, E, 256]."""
    T, E, N, _ = gi.shape
    assert gi.dtype == torch.bfloat16 and N <= RECUR_MAX_AGENTS and tuple(comm.shape) == (T, E, N, N)
    comm = comm.contiguous()
    comm = comm.view(torch.uint8) if comm.dtype == torch.bool else comm.to(torch.uint8)
    if h0 is not None:
        h0 = h0.detach().to(torch.bfloat16).reshape(E, N, 256).contiguous()
    return _RecurTrain.apply(gi.contiguous(), h0, comm, w_pack, b_pack, *params)
