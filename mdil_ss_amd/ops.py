"""Block-level operators of the ERFNet-RAP hot path: ``torch.autograd.Function``s whose forward
and backward are sequences of calls into libmdil_hip.so (C ABI, include/mdil_hip.h).

Activations are NHWC fp32 tensors ``[N, H, W, C]`` (contiguous); weights stay in the reference's
PyTorch layouts and are re-packed on device (cached per optimizer step).  Nothing here touches
``oracle/`` and nothing falls back to eager torch math: a missing extension raises.

Reference semantics implemented (file:line into prachigarg23/MDIL-SS):
  DownFn   DownsamplerBlock.forward        models/erfnet_RA_parallel.py:21-25
  NbFn     non_bottleneck_1d(_RAP).forward models/erfnet_RA_parallel.py:48-64, 90-113
  UpFn     UpsamplerBlock.forward          models/erfnet_RA_parallel.py:159-162
  OutFn    Decoder.output_conv             models/erfnet_RA_parallel.py:179-180,188
  CEFn     CrossEntropyLoss2d              train_new_task_step2.py:84-92
  KLDFn    KLDivLoss()(softmax, softmax)   train_new_task_step2.py:241,296-297
"""
import ctypes as C
import weakref as _weakref

import torch

from . import _lib
from ._lib import Epilogue, Geom

# (tests/helpers.kernel_build_id hashes the device sources AND this file's code -- with engine.py, _lib.py
# and the model containers -- so recorded mIoU samples of a build whose launches differ are excluded from
# the parity statistic without anybody having to remember a counter.)
BN_EPS = 1e-3
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _stream():
    """hipStream_t of torch's current stream (raw handle: torch.cuda.current_stream() builds a
    Python Stream object through several layers and costs ~8 us per call)."""
    return _raw_stream(_cur_device())


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, name="tensor"):
    if t is None:
        return
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"mdil op: {name} must be a contiguous fp32 device tensor "
                           f"(got {t.dtype}, {t.device}, contiguous={t.is_contiguous()})")


_ws = {}


def workspace(nbytes, device):
    """Per-device scratch buffer handed to the library (it allocates nothing itself)."""
    dev = _cur_device()
    key = (dev, _raw_stream(dev))                         # one scratch buffer per (device, stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mdil: scratch buffer would have to grow during graph capture; run "
                               "one eager iteration on the same streams first")
        buf = torch.empty(max(int(nbytes), 64 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# TICKETS (include/mdil_hip.h): one zeroed device word per (device, stream); with it the BatchNorm
# finalize steps ride inside their producing launches (the last-arriving work-group runs them).
# MDIL_NO_BNFIN=1: separate finalize launches (A/B; results are bit-identical).
BN_FIN = __import__("os").environ.get("MDIL_NO_BNFIN") is None
_tickets = {}


def _ticket(device):
    """-> device pointer of this stream's ticket word (None when the fused finalize is off)."""
    if not BN_FIN:
        return None
    dev = _cur_device()
    key = (dev, _raw_stream(dev))
    t = _tickets.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mdil: a ticket word would have to be created during graph capture; run "
                               "one eager iteration on the same streams first")
        t = _tickets[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return t.data_ptr()


def reset_tickets():
    """Zero every stream's ticket word (every engine calls this at construction: a launch that faulted
    between its arrivals would otherwise leave a count behind that no later launch completes).  The device
    is drained first: a launch still queued on another stream (a previous engine's, the pipelined frozen
    model) must not have its arrival count reset in flight -- its finalize would never fire, or fire early."""
    if not _tickets:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("mdil: ticket words cannot be reset during graph capture")
    torch.cuda.synchronize()
    for t in _tickets.values():
        t.zero_()
    torch.cuda.synchronize()


_bn_ws_bytes = {}


def _bn_ws(lib, npix, Cc, device):
    n = _bn_ws_bytes.get((npix, Cc))
    if n is None:
        n = _bn_ws_bytes[(npix, Cc)] = lib.mdil_bn_workspace(npix, Cc)
    return workspace(n, device)


def _r16(v):
    return (v + 15) // 16 * 16


_geom_cache = {}


def make_geom(N, HO, WO, HI, WI, taps, in_pitch, OH, OW, out_pitch, ihs=1, iws=1, ohs=1, oho=0,
              ows=1, owo=0, coff=0):
    """mdil_geom for one launch; memoised (the structs are read-only and a training step builds
    the same ~200 geometries over and over)."""
    key = (N, HO, WO, HI, WI, tuple(taps), in_pitch if isinstance(in_pitch, int) else tuple(in_pitch),
           OH, OW, out_pitch, ihs, iws, ohs, oho, ows, owo, coff)
    g = _geom_cache.get(key)
    if g is not None:
        return g
    g = _make_geom(N, HO, WO, HI, WI, taps, in_pitch, OH, OW, out_pitch, ihs, iws, ohs, oho, ows,
                   owo, coff)
    _geom_cache[key] = g
    return g


def _make_geom(N, HO, WO, HI, WI, taps, in_pitch, OH, OW, out_pitch, ihs, iws, ohs, oho, ows, owo,
               coff):
    g = Geom()
    g.N, g.HO, g.WO, g.HI, g.WI = N, HO, WO, HI, WI
    g.ihs, g.iws = ihs, iws
    g.ntaps = len(taps)
    for i, (dh, dw, s) in enumerate(taps):
        g.dh[i], g.dw[i], g.src[i] = dh, dw, s
    if isinstance(in_pitch, int):
        in_pitch = (in_pitch, in_pitch)
    g.in_pitch[0], g.in_pitch[1] = in_pitch
    g.OH, g.OW = OH, OW
    g.ohs, g.oho, g.ows, g.owo = ohs, oho, ows, owo
    g.out_pitch, g.out_coff = out_pitch, coff
    return g


# Diagnostic tap for the parity tests: when GATE_LOG is a list, every block operator appends the
# ReLU masks of its train-mode forward ([N,C,H,W] bool, in the order the block applies its ReLUs),
# so a checker can replay exactly these gates.  A dict {slot: list} keeps the graphs of a
# multi-stream step apart (keyed by SINK_SLOT: 0 = new-task graph, 1 = old-task graph).  The tap
# only READS the activations the shipped path produced (after the block-level C-ABI call): it does
# not change which kernels run.  Off (None) in normal operation.
GATE_LOG = None


def _log_gates(*acts):
    if GATE_LOG is not None:
        log = GATE_LOG[SINK_SLOT] if isinstance(GATE_LOG, dict) else GATE_LOG
        for t in acts:
            log.append((t > 0).permute(0, 3, 1, 2))


# Per-launch timing (bench.py's roofline leg): the library itself brackets every conv / weight-
# gradient launch with HIP events on the launch stream between profile_begin() and profile_end()
# (csrc/prof.cpp), so what is timed is the shipped call sequence -- block-level C ABI, deferred
# reductions -- not a re-orchestrated copy of it.
_PROF_CONV = ("tapconv", "sconv", "wconv", "c16conv", "w4conv")
_PROF_WGRAD = ("wgrad", "wgrad2", "wgradw")


def profile_begin(capacity=8192):
    _lib.check(_lib.load().mdil_profile_begin(capacity), "mdil_profile_begin")


def profile_end(capacity=8192):
    """-> [(kernel family, cin, cout, ntaps, algorithmic (direct-form) flops, seconds)] in launch
    order; synchronises the timed launches."""
    buf = (_lib.ProfileRecord * capacity)()
    n = _lib.load().mdil_profile_end(buf, capacity)
    if n < 0:
        _lib.check(n, "mdil_profile_end")
    out = []
    for r in buf[:n]:
        if r.ms < 0:
            continue
        kind = (_PROF_WGRAD if r.kind else _PROF_CONV)[r.path]
        out.append((kind, r.cin, r.cout, r.ntaps, 2.0 * r.npix * r.ntaps * r.cin * r.cout, r.ms * 1e-3))
    return out


def tapconv(g, cin, cout, in0, in1, wpk, out, bias=None, scale=None, shift=None, res=None,
            res_gate=None, gate=None, relu=False, bias2=None):
    e = Epilogue(_p(bias), _p(scale), _p(shift), _p(res), _p(res_gate), _p(gate), 1 if relu else 0,
                 _p(bias2))
    lib = _lib.load()
    _lib.check(lib.mdil_tapconv(C.byref(g), cin, cout, _p(in0), _p(in1), _p(wpk), C.byref(e),
                                _p(out), _stream()), "mdil_tapconv")
    return out


_stat_blocks = {}
BN_BWD_UNFUSED = __import__("os").environ.get("MDIL_NO_BNFUSE") is not None   # A/B: separate reductions


def tapconv_bn(g, cin, cout, in0, in1, wpk, out, gamma, beta, rm, rv, nbt, bias=None, bias2=None):
    """conv (+bias) -> train-mode BatchNorm statistics of its output: ``out`` is written and the
    coefficient table [4][C] (save_mean, save_invstd, scale, shift) returned; running statistics
    are updated in place.  One foreign call (``mdil_tapconv_bn_train``): where the streaming conv
    kernel covers the call the statistics ride in its epilogue and the launch's last-arriving
    work-group turns them into the coefficients; otherwise the output is re-read by the statistics
    kernel inside the call."""
    lib = _lib.load()
    npix = out.numel() // cout
    ws = _bn_ws(lib, npix, cout, out.device)
    e = Epilogue(_p(bias), None, None, None, None, None, 0, _p(bias2))
    coef = torch.empty(4, cout, dtype=torch.float32, device=out.device)
    bn = _lib.BnTrain(_p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt), BN_EPS, BN_MOMENTUM, coef.data_ptr())
    bn_stats_touched(rm)
    _lib.check(lib.mdil_tapconv_bn_train(C.byref(g), cin, cout, _p(in0), _p(in1), _p(wpk), C.byref(e),
                                         _p(out), C.byref(bn), ws.data_ptr(), ws.numel(),
                                         _ticket(out.device), _stream()), "mdil_tapconv_bn_train")
    return coef


def _ktap_arr(ktap):
    return (C.c_int * len(ktap))(*ktap)


_pack_cache = {}     # key -> packed image (persistent: refreshed in place, never re-allocated)
_pack_jobs = []      # (weak ref of the source tensor, packed image, PackJob) of every cached image
_pack_table = None   # (device uint8 tensor holding the PackJob array, number of jobs)
_pack_table_tr = None  # the same for the images of trainable sources: (table, selection, number of jobs)


def pack_into(dst, w, ktap, M, K, s_m, s_k, stem=False, register=True):
    """dst: [len(ktap), M_P, K_P] fp32 (device).  dst[t][m][k] = w.flat[m*s_m + k*s_k + ktap[t]].
    The job is remembered so ``refresh_packs`` can redo every image in one launch."""
    lib = _lib.load()
    _lib.check(lib.mdil_pack_weights(_p(w), _p(dst), len(ktap), _ktap_arr(ktap), M, K,
                                     dst.shape[1], dst.shape[2], s_m, s_k, 1 if stem else 0,
                                     _stream()), "mdil_pack_weights")
    if register and not torch.cuda.is_current_stream_capturing():
        j = _lib.PackJob(w.data_ptr(), dst.data_ptr(), len(ktap), M, K, dst.shape[1], dst.shape[2],
                         s_m, s_k, 1 if stem else 0)
        for i, k in enumerate(ktap):
            j.ktap[i] = k
        _pack_jobs.append((_weakref.ref(w), dst, j))
    return dst


_pack_gen = 0        # bumped whenever cached image POINTERS may have changed


def invalidate_packs():
    """Forget every packed image (parameter storage changed: new model / re-homed parameters)."""
    global _pack_table, _pack_table_tr, _pack_gen
    _pack_gen += 1
    _pack_table_tr = None
    if not any(st.n for st in _defer_states.values()):
        _defer_states.clear()            # arenas of streams that no longer exist (a new engine was built)
    _pack_cache.clear()
    _pack_src.clear()
    del _pack_jobs[:]
    _pack_table = None


def refresh_packs(trainable_only=False, updated=None):
    """Parameter VALUES changed in place (optimizer step, load_state_dict): redo every cached
    packed image with ONE launch over a job table resident in device memory.
    ``trainable_only`` (the fused optimizer's call): only the images of the parameters the optimizer
    just rewrote -- a frozen model's images are not rewritten while another stream may still be reading
    them (the pipelined frozen-model forward of Step2Engine runs past the end of ``iteration``).
    ``updated = (first byte, byte count)`` of the optimizer's flat parameter buffer: an image is selected
    when its source storage lies inside it, whatever the tensor's CURRENT ``requires_grad`` says (a
    parameter frozen after the optimizer was built still moves in the flat buffer -- weight decay -- and
    the C-ABI update does not bump ``._version``); without it, by ``requires_grad`` (plain callers)."""
    global _pack_table, _pack_table_tr, PARAM_GEN
    PARAM_GEN += 1
    _purge_dead_packs()
    if not _pack_jobs:
        return
    lib = _lib.load()

    def table(jobs):
        arr = (_lib.PackJob * len(jobs))(*jobs)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host.to(_pack_jobs[0][1].device)

    if trainable_only:
        if updated is not None:
            lo, hi = int(updated[0]), int(updated[0]) + int(updated[1])
            sel = tuple(lo <= r().data_ptr() < hi for r, _, _ in _pack_jobs)
        else:
            sel = tuple(bool(r().requires_grad) for r, _, _ in _pack_jobs)
        if not any(sel):
            return
        if _pack_table_tr is None or _pack_table_tr[1] != sel:
            _pack_table_tr = (table([j for (_, _, j), k in zip(_pack_jobs, sel) if k]), sel, sum(sel))
        _lib.check(lib.mdil_pack_weights_batch(_pack_table_tr[0].data_ptr(), _pack_table_tr[2], _stream()),
                   "mdil_pack_weights_batch")
        return
    if _pack_table is None or _pack_table[1] != len(_pack_jobs):
        _pack_table = (table([j for _, _, j in _pack_jobs]), len(_pack_jobs))
    _lib.check(lib.mdil_pack_weights_batch(_pack_table[0].data_ptr(), _pack_table[1], _stream()),
               "mdil_pack_weights_batch")


_pack_src = {}       # key -> (source tensors, their ._version when the image was last known fresh)
_pack_dead = [False]


def _source_died(key):
    """weakref.finalize callback of a weight tensor a packed image was built from: the caches are
    keyed by raw data pointers, and the allocator may hand the same address to an unrelated weight
    later -- the image, its re-pack job and every block descriptor that holds its pointer go."""
    global _pack_gen
    _pack_cache.pop(key, None)
    _pack_src.pop(key, None)
    _pack_dead[0] = True
    _pack_gen += 1              # cached block descriptors hold the image's pointer: none may be re-hit
    _nb_templates.clear()


def _purge_dead_packs():
    global _pack_table, _pack_table_tr, _pack_gen
    if not _pack_dead[0] and all(r() is not None for r, _, _ in _pack_jobs):
        return
    _pack_dead[0] = False
    _pack_jobs[:] = [(r, d, j) for r, d, j in _pack_jobs if r() is not None]
    _pack_table = _pack_table_tr = None
    _pack_gen += 1                                   # cached block descriptors hold image pointers
    _nb_templates.clear()



def _cached(key, builder, srcs=()):
    """Packed image for ``key``.  ``srcs`` are the weight tensors it was built from: if torch has
    modified one of them in place since (``p.data.copy_``, a foreign optimizer, EMA -- anything
    that bumps ``._version``; the fused Adam kernel and ``load_state_dict`` refresh the images
    themselves), every cached image is refreshed with one launch before it is used."""
    t = _pack_cache.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mdil: a packed weight image is missing during graph capture; run one "
                               "eager iteration before capturing")
        t = builder()
        _pack_cache[key] = t
        # weak references: the cache must not keep the weights of discarded (eval-only) models alive
        _pack_src[key] = (tuple(_weakref.ref(s) for s in srcs), tuple(s._version for s in srcs))
        for s in srcs:
            _weakref.finalize(s, _source_died, key)
        return t
    rec = _pack_src.get(key)
    if rec is not None:
        for r_, v in zip(rec[0], rec[1]):
            s_ = r_()
            if s_ is not None and s_._version != v:
                _stale_refresh()
                break
    return t


def _stale_refresh():
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("mdil: a weight changed in place under a captured graph")
    refresh_packs()
    for k, (refs, old) in list(_pack_src.items()):
        _pack_src[k] = (refs, tuple(o if r() is None else r()._version for r, o in zip(refs, old)))


def pack_conv(w, mode, ktap=None, k_pad=None):
    """Packed image of a conv ([CO,CI,KH,KW]) or transposed-conv ([CI,CO,KH,KW]) weight.
    mode: 'fwd' | 'dgrad' (conv), 't_fwd' | 't_dgrad' (transposed conv)."""
    T = w.shape[2] * w.shape[3]
    if ktap is None:
        ktap = tuple(range(T))
    key = (w.data_ptr(), mode, tuple(ktap), k_pad)

    def build():
        if mode == "fwd":
            M, K, s_m, s_k = w.shape[0], w.shape[1], w.shape[1] * T, T
        elif mode == "dgrad":
            M, K, s_m, s_k = w.shape[1], w.shape[0], T, w.shape[1] * T
        elif mode == "t_fwd":
            M, K, s_m, s_k = w.shape[1], w.shape[0], T, w.shape[1] * T
        elif mode == "t_dgrad":
            M, K, s_m, s_k = w.shape[0], w.shape[1], w.shape[1] * T, T
        else:
            raise ValueError(mode)
        dst = torch.empty(len(ktap), _r16(M), k_pad or _r16(K), dtype=torch.float32, device=w.device)
        return pack_into(dst, w, ktap, M, K, s_m, s_k)

    return _cached(key, build, (w,))


def pack_pair(w3, wa, mode):
    """[4][C][C] image: 3 taps of a factorised conv + the 1x1 adapter as 4th tap (or just the 3
    taps when there is no adapter)."""
    if wa is None:
        return pack_conv(w3, mode)
    key = (w3.data_ptr(), wa.data_ptr(), mode, "pair")

    def build():
        Cc = w3.shape[0]
        dst = torch.empty(4, Cc, Cc, dtype=torch.float32, device=w3.device)
        if mode == "fwd":
            pack_into(dst[:3], w3, (0, 1, 2), Cc, Cc, Cc * 3, 3)
            pack_into(dst[3:], wa, (0,), Cc, Cc, Cc, 1)
        else:
            raise ValueError(mode)
        return dst

    return _cached(key, build, (w3, wa))


SINK_SLOT = 0   # which of a parameter's gradient sinks new autograd nodes will accumulate into
SINK_GEN = 0    # bumped by whoever (re)installs gradient sinks: cached block descriptors hold their pointers


def sinks_changed():
    global SINK_GEN
    SINK_GEN += 1


def _sink(p):
    """Flat-gradient view installed by engine.FlatAdam: kernels accumulate into it directly and
    autograd gets None back (no per-parameter AccumulateGrad launches).  A parameter may own two
    sinks (engine's two-stream mode: the CE graph and the KD graph run their backward passes
    concurrently and must not read-modify-write the same buffer); SINK_SLOT selects."""
    if p is None:
        return None
    s = getattr(p, "_mdil_grad_sink", None)
    if s is not None and SINK_SLOT == 1:
        s2 = getattr(p, "_mdil_grad_sink2", None)
        return s2 if s2 is not None else s
    return s


def _grad_target(w, b, cout, dw, db):
    """-> (weight target, bias target, sunk?) for a wgrad launch."""
    sw, sb = _sink(w), _sink(b)
    if sw is not None and (b is None or sb is not None):
        return sw, sb, True
    tw = dw if dw is not None else torch.zeros_like(w)
    tb = db if db is not None else (torch.zeros(cout, dtype=torch.float32, device=w.device)
                                    if b is not None else None)
    return tw, tb, False


# Weight-gradient launches feed nothing but the optimizer, so they need not sit on the backward's
# critical path (bn-backward -> dgrad -> dgrad ...).  With ASYNC_WGRAD (set by the engine, and only
# effective for parameters that own gradient sinks) each stream hands its wgrad launches to a
# companion side stream; join_side_streams() makes a stream wait for all of them.
ASYNC_WGRAD = False
_side_streams = {}


def _side_stream(cur):
    s = _side_streams.get(cur.cuda_stream)
    if s is None:
        s = torch.cuda.Stream()
        _side_streams[cur.cuda_stream] = s
    return s


def join_side_streams(stream=None):
    stream = stream or torch.cuda.current_stream()
    for s in _side_streams.values():
        stream.wait_stream(s)


_wgrad_ws_bytes = {}


def wgrad(g, cin, cout, in0, in1, gout, ktap, s_co, s_ci, w, b, dw=None, db=None, second=None):
    """Weight/bias gradient of one tap-conv launch, ACCUMULATED (only the taps in ``ktap`` are
    touched, so the parity classes of a transposed conv can share one buffer).
    ``second = (ntaps2, s_co2, s_ci2, w2, b2)`` routes the trailing taps to another weight (the
    1x1 adapter riding as 4th tap of a 1x3 conv).
    -> (dw, db[, dw2, db2]) for autograd; None where the parameter owns a gradient sink."""
    lib = _lib.load()
    tw, tb, sunk = _grad_target(w, b, cout, dw, db)
    n2, sc2, si2, tw2, tb2, sunk2 = 0, 0, 0, None, None, True
    if second is not None:
        n2, sc2, si2, w2, b2 = second
        tw2, tb2, sunk2 = _grad_target(w2, b2, cout, None, None)
    kt = _ktap_arr(ktap) if ktap is not None else None

    def launch():
        need = _wgrad_ws_bytes.get((id(g), cin, cout))     # geometries are memoised: id is stable
        if need is None:
            need = _wgrad_ws_bytes[(id(g), cin, cout)] = lib.mdil_wgrad_workspace(C.byref(g), cin, cout)
        ws = workspace(need, w.device)
        _lib.check(lib.mdil_wgrad(C.byref(g), cin, cout, _p(in0), _p(in1), _p(gout), kt, s_co, s_ci,
                                  _p(tw), _p(tb), n2, sc2, si2, _p(tw2), _p(tb2), 1, ws.data_ptr(),
                                  ws.numel(), _stream()), "mdil_wgrad")

    if ASYNC_WGRAD and sunk and sunk2:
        cur = torch.cuda.current_stream()
        side = _side_stream(cur)
        side.wait_stream(cur)                 # inputs (gout, activations) are complete on `cur`
        with torch.cuda.stream(side):
            launch()
        for t in (in0, in1, gout):            # keep their storage alive until the side stream is done
            if t is not None:
                t.record_stream(side)
    else:
        launch()
    out = (None, None) if sunk else (tw, tb)
    if second is not None:
        out = out + ((None, None) if sunk2 else (tw2, tb2))
    return out


def bn_train_stats(z, gamma, beta, rm, rv, nbt):
    """-> coef [4][C]: save_mean, save_invstd, scale, shift (running stats updated in place)."""
    lib = _lib.load()
    Cc = z.shape[-1]
    npix = z.numel() // Cc
    coef = torch.empty(4, Cc, dtype=torch.float32, device=z.device)
    ws = _bn_ws(lib, npix, Cc, z.device)
    c0, row = coef.data_ptr(), 4 * Cc
    bn_stats_touched(rm)
    _lib.check(lib.mdil_bn_train_stats(_p(z), npix, Cc, _p(gamma), _p(beta), _p(rm), _p(rv),
                                       _p(nbt), BN_EPS, BN_MOMENTUM, c0, c0 + row,
                                       c0 + 2 * row, c0 + 3 * row, ws.data_ptr(), ws.numel(),
                                       _ticket(z.device), _stream()), "mdil_bn_train_stats")
    return coef


# Eval-mode coefficients are a pure function of (gamma, beta, running_mean, running_var): a frozen
# model's (the step-2 teacher: 39 BatchNorms) never change, yet round 3 recomputed them with one
# launch per BatchNorm and forward.  They are cached per BatchNorm and recomputed only when one of
# the four tensors may have changed: torch-side in-place writes bump ``._version``; writes through the
# C ABI do not, so train-mode launches mark the statistics they rewrite (``bn_stats_touched``) and
# the fused optimizer bumps ``PARAM_GEN`` (``refresh_packs``), which counts for trainable affines only.
_bn_gen = {}            # running_mean.data_ptr() -> train-mode launches that rewrote the statistics there
PARAM_GEN = 0           # optimizer steps through the C ABI (parameter VALUES changed in place)
_eval_coefs = {}        # (gamma ptr, running_mean ptr) -> [weak refs, stamp, coef, event, streams that may read]
EVAL_COEF_CACHE = __import__("os").environ.get("MDIL_NO_EVALCACHE") is None
EVAL_COEF_COUNT = {"computed": 0, "cached": 0}


def bn_stats_touched(*rms):
    """A train-mode launch is about to rewrite these running statistics through raw pointers."""
    for rm in rms:
        if rm is not None:
            k = rm.data_ptr()
            _bn_gen[k] = _bn_gen.get(k, 0) + 1


def _eval_stamp(gamma, beta, rm, rv):
    trainable = gamma.requires_grad or beta.requires_grad
    return (gamma._version, beta._version, rm._version, rv._version, _bn_gen.get(rm.data_ptr(), 0),
            PARAM_GEN if trainable else -1)


def bn_eval_coeffs(gamma, beta, rm, rv):
    """-> [2][C]: scale, shift from running statistics (cached while the four tensors are unchanged;
    the returned tensor is shared -- read only)."""
    lib = _lib.load()
    Cc = gamma.numel()
    st = _stream()
    key = (gamma.data_ptr(), rm.data_ptr())
    rec = _eval_coefs.get(key) if EVAL_COEF_CACHE else None
    if rec is not None:
        refs, stamp, coef, ev, readers = rec
        same = all(r() is t for r, t in zip(refs, (gamma, beta, rm, rv)))
        if same and stamp == _eval_stamp(gamma, beta, rm, rv) and not torch.cuda.is_current_stream_capturing():
            if st not in readers:               # first use on another stream: order it behind the launch
                torch.cuda.current_stream().wait_event(ev)
                readers.add(st)
            EVAL_COEF_COUNT["cached"] += 1
            return coef
    coef = torch.empty(2, Cc, dtype=torch.float32, device=gamma.device)
    _lib.check(lib.mdil_bn_eval_coeffs(Cc, _p(gamma), _p(beta), _p(rm), _p(rv), BN_EPS,
                                       coef.data_ptr(), coef.data_ptr() + 4 * Cc, st),
               "mdil_bn_eval_coeffs")
    EVAL_COEF_COUNT["computed"] += 1
    if EVAL_COEF_CACHE and not torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()
        ev.record()
        if rec is not None:
            # the replaced table may still be read by launches queued on other streams: it goes back to
            # the allocator only behind them (record_stream), not at this stream's position
            for s_ in rec[4]:
                if s_ != st:
                    rec[2].record_stream(torch.cuda.ExternalStream(s_))
        else:
            # entries are keyed by raw pointers: evict when the BatchNorm's tensors die (a discarded
            # model would otherwise leave its tables behind, and the address may be handed out again)
            _weakref.finalize(gamma, _evict_eval_coef, key, rm.data_ptr())
        _eval_coefs[key] = [tuple(_weakref.ref(t) for t in (gamma, beta, rm, rv)),
                            _eval_stamp(gamma, beta, rm, rv), coef, ev, {st}]
    return coef


def _evict_eval_coef(key, rm_ptr):
    _eval_coefs.pop(key, None)
    _bn_gen.pop(rm_ptr, None)


def bn_apply(z, scale, shift, drop=None, res=None, relu=True, out=None):
    lib = _lib.load()
    Cc = z.shape[-1]
    npix = z.numel() // Cc
    if out is None:
        out = torch.empty_like(z)
    _lib.check(lib.mdil_bn_apply(_p(z), npix, npix // z.shape[0], Cc, _p(scale), _p(shift),
                                 _p(drop), _p(res), 1 if relu else 0, _p(out), _stream()),
               "mdil_bn_apply")
    return out


def bn_backward(gy, relu_src, drop, z, gamma, beta, coef, want_affine, out=None):
    """-> (gz, dgamma, dbeta); the affine grads are None when not wanted or when they were
    accumulated into the parameters' gradient sinks."""
    lib = _lib.load()
    Cc = z.shape[-1]
    npix = z.numel() // Cc
    gz = torch.empty_like(z) if out is None else out
    dg = db = None
    sunk = False
    if want_affine:
        sg, sb = _sink(gamma), _sink(beta)
        sunk = sg is not None and sb is not None
        if sunk:
            dg, db = sg, sb
        else:
            dgb = torch.zeros(2, Cc, dtype=torch.float32, device=z.device)
            dg, db = dgb[0], dgb[1]
    ws = _bn_ws(lib, npix, Cc, z.device)
    _lib.check(lib.mdil_bn_backward(_p(gy), _p(relu_src), _p(drop), _p(z), npix,
                                    npix // z.shape[0], Cc, _p(gamma), coef.data_ptr(),
                                    coef.data_ptr() + 4 * Cc,
                                    _p(dg), _p(db), 1, _p(gz), ws.data_ptr(), ws.numel(),
                                    _ticket(z.device), _stream()), "mdil_bn_backward")
    if sunk or not want_affine:
        return gz, None, None
    return gz, dg, db


def _affine_sinks(gamma, beta, want_affine):
    """-> (dgamma ptr tensor, dbeta, ok): the gradient sinks of a BatchNorm's affine parameters.
    ok is False when the gradients are wanted but a parameter owns no sink (plain autograd users):
    the finalize then cannot ride in a producer launch."""
    if not want_affine:
        return None, None, True
    sg, sb = _sink(gamma), _sink(beta)
    return sg, sb, sg is not None and sb is not None


def tapconv_bnred(g, cin, cout, in0, in1, wpk, out, gate, z, coef, fin=None):
    """dgrad launch that stores g = conv(...) * (gate > 0) AND emits the BatchNorm-backward
    reductions of g against the BN input ``z`` (``coef`` = the forward's [4][C] table).
    -> (g, partial pointer, nblk), or None when the streaming kernel does not cover the call.
    ``fin = (gamma, dgamma sink, dbeta sink)``: the launch also FINALIZES the reductions (its
    last-arriving work-group, ``_ticket``): -> (g, coef3 [3][C], 0) for ``bn_backward_apply``."""
    lib = _lib.load()
    key = (id(g), cin, cout)
    nblk = _stat_blocks.get(key)
    if nblk is None:
        nblk = _stat_blocks[key] = lib.mdil_tapconv_stat_blocks(C.byref(g), cin, cout)
    if nblk == 0:
        return None
    npix = out.numel() // cout
    ws = _bn_ws(lib, npix, cout, out.device)
    partial = ws.data_ptr()
    e = Epilogue(None, None, None, None, None, _p(gate), 0, None)
    fp, coef3 = None, None
    if fin is not None:
        coef3 = torch.empty(3, cout, dtype=torch.float32, device=out.device)
        fs = _lib.BnGrad(_p(fin[0]), _p(fin[1]), _p(fin[2]), 1, coef3.data_ptr(), _ticket(out.device))
        fp = C.byref(fs)
    _lib.check(lib.mdil_tapconv_bnred(C.byref(g), cin, cout, _p(in0), _p(in1), _p(wpk), C.byref(e),
                                      _p(out), _p(z), coef.data_ptr(), coef.data_ptr() + 4 * cout,
                                      partial, fp, _stream()), "mdil_tapconv_bnred")
    if fin is not None:
        return out, coef3, 0
    return out, partial, nblk


def bn_backward_apply(g, z, coef, coef3, drop=None, out=None):
    """Last pass of a BatchNorm backward whose reductions were finalized by their producer."""
    lib = _lib.load()
    Cc = z.shape[-1]
    npix = z.numel() // Cc
    gz = torch.empty_like(z) if out is None else out
    _lib.check(lib.mdil_bn_backward_apply(_p(g), _p(drop), _p(z), npix, npix // z.shape[0], Cc,
                                          coef.data_ptr(), coef.data_ptr() + 4 * Cc, coef3.data_ptr(),
                                          _p(gz), _stream()), "mdil_bn_backward_apply")
    return gz


def bn_backward_partials(g, z, gamma, beta, coef, want_affine, partial, nblk, out=None):
    """BatchNorm backward given the reductions (``tapconv_bnred``) and the already gated gradient
    ``g``: finalize + apply.  -> (gz, dgamma, dbeta) like ``bn_backward``."""
    lib = _lib.load()
    Cc = z.shape[-1]
    npix = z.numel() // Cc
    gz = torch.empty_like(z) if out is None else out
    dg = db = None
    sunk = False
    if want_affine:
        sg, sb = _sink(gamma), _sink(beta)
        sunk = sg is not None and sb is not None
        if sunk:
            dg, db = sg, sb
        else:
            dgb = torch.zeros(2, Cc, dtype=torch.float32, device=z.device)
            dg, db = dgb[0], dgb[1]
    ws = _bn_ws(lib, npix, Cc, z.device)
    cws = ws.data_ptr() + (256 * 2 * Cc + 256) * 4          # behind the partial region
    _lib.check(lib.mdil_bn_backward_partials(_p(g), None, _p(z), npix, npix // z.shape[0], Cc, _p(gamma),
                                             coef.data_ptr(), coef.data_ptr() + 4 * Cc, partial, nblk,
                                             _p(dg), _p(db), 1, _p(gz), cws, 3 * Cc * 4, _stream()),
               "mdil_bn_backward_partials")
    if sunk or not want_affine:
        return gz, None, None
    return gz, dg, db


# ----------------------------------------------------------------------------------------------
# geometry of the reference's convolutions
# ----------------------------------------------------------------------------------------------
def _g_s1(N, H, W, Cc, taps, two_src=False):
    return make_geom(N, H, W, H, W, taps, Cc, H, W, Cc)


def _taps_3x1(d, flip=False):
    s = -1 if flip else 1
    return [(s * (k - 1) * d, 0, 0) for k in range(3)]


def _taps_1x3(d, flip=False):
    s = -1 if flip else 1
    return [(0, s * (k - 1) * d, 0) for k in range(3)]


# parity classes of a 3x3 / stride-2 transposed convolution (and of a stride-2 conv's dgrad):
# output row 2i+a takes kernel rows kh with input row i+dh:  a=0 -> (kh=1, dh=0);
# a=1 -> (kh=0, dh=+1), (kh=2, dh=0).   Same along W.
_PAR = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}


def _class_taps(a, b):
    taps, ktap = [], []
    for kh, dh in _PAR[a]:
        for kw, dw in _PAR[b]:
            taps.append((dh, dw, 0))
            ktap.append(kh * 3 + kw)
    return taps, ktap


# ----------------------------------------------------------------------------------------------
# DownsamplerBlock
# ----------------------------------------------------------------------------------------------
class DownFn(torch.autograd.Function):
    """relu(bn_ini[task](cat[conv3x3 s2 p1 (x), maxpool2x2 (x)])) on NHWC tensors."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rm, rv, nbt, train, link_out):
        lib = _lib.load()
        ctx.sink_slot = SINK_SLOT
        ctx.link_out = link_out
        _chk(x, "x")
        N, H, W, cin = x.shape
        cc = w.shape[0]
        cout = cc + cin
        HO, WO = H // 2, W // 2
        stem = cin == 3
        z = torch.empty(N, HO, WO, cout, dtype=torch.float32, device=x.device)
        if stem:
            wp = _cached((w.data_ptr(), "stem"), lambda: pack_into(
                torch.empty(1, 16, 32, dtype=torch.float32, device=w.device), w, (0,), cc, 27,
                27, 1, stem=True), (w,))
            g = make_geom(N, HO, WO, H, W, [(0, 0, 0)], 3, HO, WO, cout, ihs=2, iws=2)
            tapconv(g, 27, cc, x, None, wp, z, bias=b)
        else:
            taps = [(kh - 1, kw - 1, 0) for kh in range(3) for kw in range(3)]
            g = make_geom(N, HO, WO, H, W, taps, cin, HO, WO, cout, ihs=2, iws=2)
            tapconv(g, cin, cc, x, None, pack_conv(w, "fwd"), z, bias=b)
        _lib.check(lib.mdil_maxpool_concat_fwd(_p(x), N, H, W, cin, _p(z), cout, cc, _stream()),
                   "mdil_maxpool_concat_fwd")
        if train:
            coef = bn_train_stats(z, gamma, beta, rm, rv, nbt)
            y = bn_apply(z, coef[2], coef[3], relu=True)
            ctx.save_for_backward(x, w, b, gamma, beta, z, y, coef)
            _log_gates(y)
            _attach_tail(link_out, y, z, coef, gamma, beta)
        else:
            ec = bn_eval_coeffs(gamma, beta, rm, rv)
            y = bn_apply(z, ec[0], ec[1], relu=True)
        ctx.stem = stem
        return y

    @staticmethod
    def backward(ctx, gy):
        global SINK_SLOT
        SINK_SLOT = ctx.sink_slot
        lib = _lib.load()
        x, w, b, gamma, beta, z, y, coef = ctx.saved_tensors
        gy = gy.contiguous()
        N, H, W, cin = x.shape
        cc = w.shape[0]
        cout = cc + cin
        HO, WO = H // 2, W // 2
        need = ctx.needs_input_grad
        gz, dgamma, dbeta = _bn_backward_maybe_fused(ctx.link_out, gy, y, z, gamma, beta, coef, need[3] or need[4])
        dw = db = gx = None
        if need[1] or need[2]:
            if ctx.stem:
                g = make_geom(N, HO, WO, H, W, [(0, 0, 0)], 3, HO, WO, cout, ihs=2, iws=2)
                dw, db = wgrad(g, 27, cc, x, None, gz, None, 0, 0, w, b)
            else:
                taps = [(kh - 1, kw - 1, 0) for kh in range(3) for kw in range(3)]
                g = make_geom(N, HO, WO, H, W, taps, cin, HO, WO, cout, ihs=2, iws=2)
                dw, db = wgrad(g, cin, cc, x, None, gz, tuple(range(9)), cin * 9, 9, w, b)
        if need[0]:
            gx = torch.empty_like(x)
            _lib.check(lib.mdil_maxpool_concat_bwd(_p(x), _p(gz), N, H, W, cin, cout, cc, _p(gx),
                                                   _stream()), "mdil_maxpool_concat_bwd")
            for a in (0, 1):
                for bb in (0, 1):
                    taps, ktap = _class_taps(a, bb)
                    g = make_geom(N, HO, WO, HO, WO, taps, cout, H, W, cin, ohs=2, oho=a, ows=2,
                                  owo=bb)
                    tapconv(g, cc, cin, gz, None, pack_conv(w, "dgrad", tuple(ktap)), gx, res=gx)
        return gx, dw, db, dgamma, dbeta, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# non_bottleneck_1d / non_bottleneck_1d_RAP
# ----------------------------------------------------------------------------------------------
def _pack_pair_dgrad(w31, pw):
    """dgrad image of a 3x1 conv with the adapter^T as 4th tap (read from the second source)."""
    if pw is None:
        return pack_conv(w31, "dgrad")
    Cc = w31.shape[0]

    def build():
        dst = torch.empty(4, Cc, Cc, dtype=torch.float32, device=w31.device)
        pack_into(dst[:3], w31, (0, 1, 2), Cc, Cc, 3, Cc * 3)
        pack_into(dst[3:], pw, (0,), Cc, Cc, 1, Cc)
        return dst

    return _cached((w31.data_ptr(), pw.data_ptr(), "pair_dgrad"), build, (w31, pw))


# One foreign call per block and direction (mdil_nb_block_forward / _backward) instead of one per
# launch.  The per-launch Python orchestration below stays as the A/B path (MDIL_PY_BLOCKS=1, the
# side-stream weight-gradient experiment); the parity tests run both against each other.  The gate
# log (parity tests, smoke) and the launch profiler (bench.py) ride on the block-ABI path itself.
BLOCK_ABI = __import__("os").environ.get("MDIL_PY_BLOCKS") is None
_nb_ws_bytes = {}


# Deferred weight-gradient reductions (mdil_wgrad_reduce_batch).  While DEFER_WGRAD is on, the block
# backward leaves the partial sums of its weight-gradient launches in a per-stream arena and queues
# their reductions; flush_wgrad() reduces a stream's queue with one launch per 16 jobs.  Only the
# engines turn it on: whoever does must flush every stream that ran a backward before the flat
# gradient buffer is read (all-reduce, optimizer step).
DEFER_WGRAD = False
WGRAD_BATCH = int(__import__("os").environ.get("MDIL_WGRAD_BATCH", "16"))   # jobs queued before a flush is forced
WGRAD_ARENA_BYTES = WGRAD_BATCH * (20 << 20)     # per stream: every queued launch holds <= 17 MB of partial sums


class _DeferState:
    __slots__ = ("jobs", "n", "cursor", "arena", "nout", "used")

    def __init__(self, device):
        self.jobs = (_lib.WgradJob * (WGRAD_BATCH + 4))()
        self.n = 0
        self.cursor = 0
        self.arena = torch.empty(WGRAD_ARENA_BYTES, dtype=torch.uint8, device=device)
        self.nout = C.c_int(0)
        self.used = C.c_size_t(0)


_defer_states = {}


def _defer_state(device):
    key = (_cur_device(), _raw_stream(_cur_device()))
    st = _defer_states.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mdil: the weight-gradient arena would be allocated during graph capture; "
                               "run one eager iteration on the same streams first")
        st = _defer_states[key] = _DeferState(device)
    return st


def flush_wgrad():
    """Reduce the weight-gradient partial sums queued on the CURRENT stream (no-op when none)."""
    st = _defer_states.get((_cur_device(), _raw_stream(_cur_device())))
    if st is None or st.n == 0:
        return
    _lib.check(_lib.load().mdil_wgrad_reduce_batch(st.jobs, st.n, _stream()), "mdil_wgrad_reduce_batch")
    C.memset(st.jobs, 0, st.n * 192)   # a stale record must never pass for a queued one
    st.n = 0
    st.cursor = 0          # later launches on this stream are ordered behind the reduction: reuse


def pending_wgrad():
    """Streams (raw handles) that still hold queued reductions -- a consistency check for callers."""
    return [k for k, st in _defer_states.items() if st.n]


_nb_templates = {}   # key -> (descriptor with its static fields filled, generations, weight versions)


def _nb_template(key, srcs):
    """Cached block descriptor whose static half (packed images, parameters, gradient sinks) is
    still valid: same pack / sink generation and no weight modified in place since."""
    rec = _nb_templates.get(key)
    if rec is None or rec[1] != (_pack_gen, SINK_GEN):
        return None
    for s_, v in zip(srcs, rec[2]):
        if s_ is not None and s_._version != v:
            return None
    return rec[0]


def _nb_template_store(key, b, srcs):
    _nb_templates[key] = (b, (_pack_gen, SINK_GEN), tuple(0 if s_ is None else s_._version for s_ in srcs))


# Block-boundary fusion of the outer BatchNorm backward (mdil_tapconv_tail, include/mdil_hip.h).
# The chain is EXPLICIT: the model creates one ``Boundary`` per block boundary of a forward pass and
# hands it to the block in front (``link_out``) and to the block behind (``link_in``) -- nothing rides
# on tensor objects (round 3 hung this state on ``_mdil_tail`` / ``_mdil_head`` attributes).
#   forward:  the producing block records what the fusion needs (its bn2 input, statistics, dropout
#             factors, affine parameters) and WHICH tensor it returned; the consuming block accepts
#             the boundary only if that very tensor is its input;
#   backward: the consuming block's last launch gates its input gradient, emits the reductions (and,
#             with a ticket, finalizes them) and records WHICH gradient tensor they describe; the
#             producing block uses them only if that very tensor -- same storage, same version: not
#             a sum autograd formed for a second consumer -- arrives as its dL/dout.
# A boundary that does not match is ignored and the unfused path runs (storing the gated gradient
# is harmless on its own: every consumer applies the same gate again; where the tail launch already
# finalized dgamma / dbeta into the sinks, _bn_backward_maybe_fused takes that contribution back out).  The C ABI sees the same
# chain as mdil_nb_block.tail / .head_coef / .head_partial.
class Boundary:
    __slots__ = ("z", "coef", "drop", "gamma", "beta", "stream", "out_ptr", "out_version", "shape",
                 "partial", "nblk", "coef3", "gx_ptr", "gx_version", "gx_stream")

    def __init__(self):
        self.clear()

    def clear(self):
        for k in self.__slots__:
            setattr(self, k, None)

    def record(self, out, z, coef, drop, gamma, beta):
        """Producer, forward: ``out = relu(bn(z) * drop [+ x])`` is what this block returns."""
        self.clear()
        self.z, self.coef, self.drop, self.gamma, self.beta = z, coef, drop, gamma, beta
        self.stream, self.out_ptr, self.out_version, self.shape = _stream(), out.data_ptr(), out._version, tuple(out.shape)

    def feeds(self, x):
        """Consumer, forward: is ``x`` the tensor the producer returned (untouched, same stream)?"""
        return (self.z is not None and self.out_ptr == x.data_ptr() and self.shape == tuple(x.shape)
                and self.out_version == x._version and self.stream == _stream())

    def emitted(self, gx, partial, nblk, coef3):
        """Consumer, backward: the reductions (``partial`` rows, or finalized ``coef3``) describe ``gx``."""
        self.partial, self.nblk, self.coef3 = partial, nblk, coef3
        self.gx_ptr, self.gx_version, self.gx_stream = gx.data_ptr(), gx._version, _stream()

    def describes(self, gy):
        """Producer, backward: is ``gy`` exactly the gradient the reductions were taken of?"""
        return (self.partial is not None and self.gx_ptr == gy.data_ptr() and self.gx_version == gy._version
                and self.gx_stream == _stream() and self.shape == tuple(gy.shape))


def boundaries(n):
    """n + 1 boundaries for a chain of n blocks: block k gets (links[k], links[k + 1])."""
    return [Boundary() for _ in range(n + 1)]


BN_TAIL = __import__("os").environ.get("MDIL_NO_BNTAIL") is None and not BN_BWD_UNFUSED
TAIL_COUNT = {"tail": 0, "head": 0}     # launches that emitted / blocks that consumed reductions (tests)
_tail_blocks = {}


def _tail_nblk(N, H, W, Cc, rap):
    key = (N, H, W, Cc, rap)
    n = _tail_blocks.get(key)
    if n is None:
        n = _tail_blocks[key] = _lib.load().mdil_nb_block_tail_blocks(N, H, W, Cc, int(rap))
    return n


def _attach_tail(link, y, z, coef, gamma=None, beta=None):
    """DownsamplerBlock / UpsamplerBlock outputs y = relu(bn(z)): the same block-boundary fusion as
    between two factorised blocks (no dropout factor, no residual)."""
    if link is not None and BN_TAIL and y.shape[3] in (64, 128):
        link.record(y, z, coef, None, gamma, beta)


def _tail_fin(b, tail, Cc, device):
    """Fill ``b.tail`` from the boundary in front of this block; when the producing block's affine
    gradients have sinks (or are not wanted) the tail launch finalizes the reductions itself.
    -> the [3][C] table it will write (None: partial rows only)."""
    b.tail.z, b.tail.save_mean = tail.z.data_ptr(), tail.coef.data_ptr()
    b.tail.save_invstd, b.tail.drop = tail.coef.data_ptr() + 4 * Cc, _p(tail.drop)
    gamma, beta = tail.gamma, tail.beta
    b.tail.fin.coef = None
    if BN_FIN and gamma is not None:
        want = gamma.requires_grad or beta.requires_grad
        sg, sb, ok = _affine_sinks(gamma, beta, want)
        if ok:
            coef3 = torch.empty(3, Cc, dtype=torch.float32, device=device)
            b.tail.fin.gamma, b.tail.fin.dgamma, b.tail.fin.dbeta = gamma.data_ptr(), _p(sg), _p(sb)
            b.tail.fin.accumulate, b.tail.fin.coef = 1, coef3.data_ptr()
            b.tail.fin.ticket = _ticket(device)
            return coef3
    return None


def _bn_backward_maybe_fused(link, gy, y, z, gamma, beta, coef, want_affine):
    """BatchNorm backward of y = relu(bn(z)) given dL/dy: with the next block's reductions for exactly
    this gradient (already gated by y > 0) finalize + apply only -- or, when that block's tail
    launch finalized them too, the apply pass alone --, else the three-pass form."""
    if BN_TAIL and link is not None and link.describes(gy):
        TAIL_COUNT["head"] += 1
        coef3, partial, nblk = link.coef3, link.partial, link.nblk
        link.clear()
        if coef3 is not None:
            return bn_backward_apply(gy, z, coef, coef3), None, None
        return bn_backward_partials(gy, z, gamma, beta, coef, want_affine, partial.data_ptr(), nblk)
    _undo_tail_finalize(link, gy, gamma, beta, want_affine)
    return bn_backward(gy, y, None, z, gamma, beta, coef, want_affine)


def _undo_tail_finalize(link, gy, gamma, beta, want_affine):
    """The boundary behind a block does not describe the gradient that arrives (a second consumer's
    gradient was added, a hook replaced it, another stream) -- but the consumer's tail launch may have
    FINALIZED its reductions already: dgamma / dbeta of ITS gradient then sit in this BatchNorm's sinks,
    and the unfused backward that follows accumulates dgamma / dbeta of what arrives, which contains
    that contribution again.  Take it back out first (coef3 rows 1, 2 are sum(g) / n and sum(g xhat) / n
    of the tail's gradient).  Rare path: a full synchronisation orders the tail's stream against this one."""
    if not BN_TAIL or link is None or link.coef3 is None:
        return
    coef3, n, tail_stream = link.coef3, gy.numel() // gy.shape[-1], link.gx_stream
    link.clear()
    if want_affine:
        sg, sb, ok = _affine_sinks(gamma, beta, True)
        if ok:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("mdil: a block boundary whose fused BatchNorm-backward finalize must be taken "
                                   "back (second consumer / replaced gradient) cannot be captured in a graph; "
                                   "set MDIL_NO_BNTAIL=1 for this model")
            # order this stream behind the tail launch that wrote coef3 and the sinks (its stream, not the device)
            cur = torch.cuda.current_stream()
            if tail_stream is not None and tail_stream != cur.cuda_stream:
                cur.wait_stream(torch.cuda.ExternalStream(tail_stream))
            sb.sub_(coef3[1] * n)
            sg.sub_(coef3[2] * n)


def _nb_block_dynamic(b, x, dil, rap):
    """Per-call fields: shape, input, per-stream scratch."""
    lib = _lib.load()
    N, H, W, Cc = x.shape
    key = (N, H, W, Cc, dil, rap)
    need = _nb_ws_bytes.get(key)
    if need is None:
        need = _nb_ws_bytes[key] = max(lib.mdil_bn_workspace(N * H * W, Cc),
                                       lib.mdil_nb_block_wgrad_workspace(N, H, W, Cc, dil, int(rap)))
    ws = workspace(need, x.device)
    b.N, b.H, b.W = N, H, W
    b.x = x.data_ptr()
    b.bn_workspace = b.wgrad_workspace = ws.data_ptr()
    b.bn_workspace_bytes = b.wgrad_workspace_bytes = ws.numel()
    b.head_partial, b.head_nblk, b.head_coef = None, 0, None   # descriptors are cached: always reset the
    b.tail.partial = None                                      # per-call fusion fields
    b.tail.fin.coef = None


class NbFn(torch.autograd.Function):
    """a=relu(c31_1(x)); z1=c13_1(a)+pc1(x); u=relu(bn1(z1)); b=relu(c31_2(u)); z2=c13_2(b)+pc2(u);
    out=relu(bn2(z2)*drop + x).  pc* / drop are None for the decoder's plain blocks."""

    @staticmethod
    def forward(ctx, x, w31_1, b31_1, w13_1, b13_1, pw1, pb1, g1, be1, w31_2, b31_2, w13_2, b13_2,
                pw2, pb2, g2, be2, bufs, drop, dil, train, link_in, link_out):
        ctx.sink_slot = SINK_SLOT
        ctx.link_out = None
        _chk(x, "x")
        N, H, W, Cc = x.shape
        rm1, rv1, nbt1, rm2, rv2, nbt2 = bufs
        rap = pw1 is not None
        ad = [(0, 0, 1)] if rap else []
        G31a = make_geom(N, H, W, H, W, _taps_3x1(1), Cc, H, W, Cc)
        G13a = make_geom(N, H, W, H, W, _taps_1x3(1) + ad, Cc, H, W, Cc)
        G31b = make_geom(N, H, W, H, W, _taps_3x1(dil), Cc, H, W, Cc)
        G13b = make_geom(N, H, W, H, W, _taps_1x3(dil) + ad, Cc, H, W, Cc)
        new = lambda: torch.empty_like(x)
        if BLOCK_ABI:
            srcs = (w31_1, w13_1, pw1, w31_2, w13_2, pw2)
            key = (w31_1.data_ptr(), g1.data_ptr(), _p(rm1), "fwd")
            b = _nb_template(key, srcs)
            if b is None:
                b = _lib.NbBlock()
                b.C, b.dilation, b.rap = Cc, dil, int(rap)
                b.bn_eps, b.bn_momentum = BN_EPS, BN_MOMENTUM
                for h, (w31, b31, w13, b13, pw, pb, gm, be, rm, rv, nbt) in enumerate((
                        (w31_1, b31_1, w13_1, b13_1, pw1, pb1, g1, be1, rm1, rv1, nbt1),
                        (w31_2, b31_2, w13_2, b13_2, pw2, pb2, g2, be2, rm2, rv2, nbt2))):
                    p = b.half[h]
                    p.wp31, p.wp13 = pack_conv(w31, "fwd").data_ptr(), pack_pair(w13, pw, "fwd").data_ptr()
                    p.b31, p.b13, p.pb = _p(b31), _p(b13), _p(pb)
                    p.gamma, p.beta = gm.data_ptr(), be.data_ptr()
                    p.running_mean, p.running_var, p.num_batches_tracked = _p(rm), _p(rv), _p(nbt)
                _nb_template_store(key, b, srcs)
            _nb_block_dynamic(b, x, dil, rap)
            b.train = 1 if train else 0
            b.drop = _p(drop)
            b.ticket = _ticket(x.device)
            if train:
                bn_stats_touched(rm1, rm2)
                coef = torch.empty(2, 4, Cc, dtype=torch.float32, device=x.device)
                c0 = coef.data_ptr()
                b.half[0].coef, b.half[1].coef = c0, c0 + 16 * Cc
                b.eval_coef_ready = 0
            else:       # folded-BN coefficients: cached per BatchNorm (one launch each when they changed)
                e1, e2 = bn_eval_coeffs(g1, be1, rm1, rv1), bn_eval_coeffs(g2, be2, rm2, rv2)
                b.half[0].coef, b.half[1].coef = e1.data_ptr(), e2.data_ptr()
                b.eval_coef_ready = 1
            a1, u, out = new(), new(), new()
            b.a1, b.u, b.out = a1.data_ptr(), u.data_ptr(), out.data_ptr()
            if train:
                z1, a2, z2 = new(), new(), new()
                b.z1, b.a2, b.z2 = z1.data_ptr(), a2.data_ptr(), z2.data_ptr()
            else:
                b.z1 = b.a2 = b.z2 = None
            _lib.check(_lib.load().mdil_nb_block_forward(C.byref(b), _stream()), "mdil_nb_block_forward")
            if train:
                ctx.save_for_backward(x, a1, z1, u, a2, z2, out, coef[0], coef[1], drop, w31_1, w13_1,
                                      pw1, g1, w31_2, w13_2, pw2, g2, b31_1, b13_1, pb1, be1, b31_2,
                                      b13_2, pb2, be2)
                ctx.dil = dil
                _log_gates(a1, u, a2, out)
                # block-boundary fusion: the boundary in front of us (accepted only if our input IS
                # the tensor its producer returned) / the boundary behind us
                ctx.tail = link_in if (BN_TAIL and link_in is not None and link_in.feeds(x)
                                       and _tail_nblk(N, H, W, Cc, rap) > 0) else None
                if BN_TAIL and link_out is not None and Cc in (64, 128):
                    link_out.record(out, z2, coef[1], drop, g2, be2)
                    ctx.link_out = link_out
            return out
        a1 = tapconv(G31a, Cc, Cc, x, None, pack_conv(w31_1, "fwd"), new(), bias=b31_1, relu=True)
        if train:
            z1 = new()
            c1 = tapconv_bn(G13a, Cc, Cc, a1, x, pack_pair(w13_1, pw1, "fwd"), z1, g1, be1, rm1, rv1,
                            nbt1, bias=b13_1, bias2=pb1)
            u = bn_apply(z1, c1[2], c1[3], relu=True)
            a2 = tapconv(G31b, Cc, Cc, u, None, pack_conv(w31_2, "fwd"), new(), bias=b31_2, relu=True)
            z2 = new()
            c2 = tapconv_bn(G13b, Cc, Cc, a2, u, pack_pair(w13_2, pw2, "fwd"), z2, g2, be2, rm2, rv2,
                            nbt2, bias=b13_2, bias2=pb2)
            out = bn_apply(z2, c2[2], c2[3], drop=drop, res=x, relu=True)
            ctx.save_for_backward(x, a1, z1, u, a2, z2, out, c1, c2, drop, w31_1, w13_1, pw1, g1,
                                  w31_2, w13_2, pw2, g2, b31_1, b13_1, pb1, be1, b31_2, b13_2,
                                  pb2, be2)
            ctx.dil = dil
            _log_gates(a1, u, a2, out)
        else:
            e1 = bn_eval_coeffs(g1, be1, rm1, rv1)
            e2 = bn_eval_coeffs(g2, be2, rm2, rv2)
            u = tapconv(G13a, Cc, Cc, a1, x, pack_pair(w13_1, pw1, "fwd"), new(), bias=b13_1,
                        bias2=pb1, scale=e1[0], shift=e1[1], relu=True)
            a2 = tapconv(G31b, Cc, Cc, u, None, pack_conv(w31_2, "fwd"), a1, bias=b31_2, relu=True)
            out = tapconv(G13b, Cc, Cc, a2, u, pack_pair(w13_2, pw2, "fwd"), new(), bias=b13_2,
                          bias2=pb2, scale=e2[0], shift=e2[1], res=x, relu=True)
        return out

    @staticmethod
    def backward(ctx, gy):
        global SINK_SLOT
        SINK_SLOT = ctx.sink_slot
        (x, a1, z1, u, a2, z2, out, c1, c2, drop, w31_1, w13_1, pw1, g1, w31_2, w13_2, pw2,
         g2, b31_1, b13_1, pb1, be1, b31_2, b13_2, pb2, be2) = ctx.saved_tensors
        gy = gy.contiguous()
        N, H, W, Cc = x.shape
        need = ctx.needs_input_grad
        if BLOCK_ABI and not ASYNC_WGRAD:
            srcs = (w31_1, w13_1, pw1, w31_2, w13_2, pw2)
            key = (w31_1.data_ptr(), g1.data_ptr(), SINK_SLOT, need, "bwd")
            res = [None] * 23
            b = _nb_template(key, srcs)
            all_sunk = b is not None           # only descriptors with every gradient sunk are cached
            if b is None:
                b = _lib.NbBlock()
                b.C, b.dilation, b.rap = Cc, ctx.dil, int(pw1 is not None)
                all_sunk = True
                for h, (i0, w31, b31, w13, b13, pw, pb, gm, be) in enumerate((
                        (1, w31_1, b31_1, w13_1, b13_1, pw1, pb1, g1, be1),
                        (9, w31_2, b31_2, w13_2, b13_2, pw2, pb2, g2, be2))):
                    p = b.half[h]
                    p.wp13 = pack_conv(w13, "dgrad").data_ptr()
                    p.wp31 = _pack_pair_dgrad(w31, pw).data_ptr()
                    p.gamma = gm.data_ptr()
                    for j, (w, bb, fw, fb) in enumerate(((w31, b31, "dw31", "db31"), (w13, b13, "dw13", "db13"),
                                                         (pw, pb, "dpw", "dpb"))):
                        if w is not None and (need[i0 + 2 * j] or need[i0 + 2 * j + 1]):
                            tw, tb, sunk = _grad_target(w, bb, Cc, None, None)
                            setattr(p, fw, _p(tw))
                            setattr(p, fb, _p(tb))
                            if not sunk:
                                all_sunk = False
                                res[i0 + 2 * j], res[i0 + 2 * j + 1] = tw, tb
                    if need[i0 + 6] or need[i0 + 7]:
                        sg, sb = _sink(gm), _sink(be)
                        if sg is None or sb is None:
                            all_sunk = False
                            dgb = torch.zeros(2, Cc, dtype=torch.float32, device=x.device)
                            sg, sb = dgb[0], dgb[1]
                            res[i0 + 6], res[i0 + 7] = sg, sb
                        p.dgamma, p.dbeta = sg.data_ptr(), sb.data_ptr()
                if all_sunk:        # fresh gradient tensors would differ from call to call
                    _nb_template_store(key, b, srcs)
            _nb_block_dynamic(b, x, ctx.dil, pw1 is not None)
            b.drop = _p(drop)
            b.half[0].coef, b.half[1].coef = c1.data_ptr(), c2.data_ptr()
            b.a1, b.z1, b.u, b.a2, b.z2, b.out = (a1.data_ptr(), z1.data_ptr(), u.data_ptr(),
                                                  a2.data_ptr(), z2.data_ptr(), out.data_ptr())
            gz2, ga, gu, gx = (torch.empty_like(x) for _ in range(4))
            b.gy, b.gz2, b.ga, b.gu, b.gx = (gy.data_ptr(), gz2.data_ptr(), ga.data_ptr(),
                                             gu.data_ptr(), gx.data_ptr())
            b.ticket = _ticket(x.device)
            # (the boundary behind us describes ONE gradient tensor: if autograd has accumulated a second
            # consumer's gradient -- a new tensor, or the same one at a later version -- it does not match)
            head = getattr(ctx, "link_out", None)
            if BN_TAIL and head is not None and head.describes(gy):
                if head.coef3 is not None:                    # finalized by the tail launch: apply only
                    b.head_coef = head.coef3.data_ptr()
                else:
                    b.head_partial, b.head_nblk = head.partial.data_ptr(), head.nblk
                ctx.head_keep = (head.partial, head.coef3)    # alive until this call's launches are enqueued
                head.clear()
                TAIL_COUNT["head"] += 1
            else:
                _undo_tail_finalize(head, gy, g2, be2, need[15] or need[16])
            tail, tail_partial, tail_coef = getattr(ctx, "tail", None), None, None
            if tail is not None and need[0]:
                nblk = _tail_nblk(N, H, W, Cc, pw1 is not None)
                tail_partial = torch.empty(nblk * 2 * Cc, dtype=torch.float32, device=x.device)
                tail_coef = _tail_fin(b, tail, Cc, x.device)
                b.tail.partial = tail_partial.data_ptr()
                TAIL_COUNT["tail"] += 1
            ws_need = _nb_ws_bytes[(N, H, W, Cc, ctx.dil, pw1 is not None)] * 4 + 4096
            if DEFER_WGRAD and all_sunk and ws_need <= WGRAD_ARENA_BYTES:
                # partial sums stay in this stream's arena; their reductions are batched.  (A block
                # whose partial sums would not fit the arena at all reduces immediately instead.)
                ds = _defer_state(x.device)
                if ds.n + 4 > WGRAD_BATCH or ds.cursor + ws_need > ds.arena.numel():
                    flush_wgrad()
                b.wgrad_workspace = ds.arena.data_ptr() + ds.cursor
                b.wgrad_workspace_bytes = ds.arena.numel() - ds.cursor
                _lib.check(_lib.load().mdil_nb_block_backward_deferred(
                    C.byref(b), C.cast(C.byref(ds.jobs, ds.n * 192), C.POINTER(_lib.WgradJob)),
                    C.byref(ds.nout), C.byref(ds.used), _stream()), "mdil_nb_block_backward_deferred")
                ds.n += ds.nout.value
                ds.cursor += ds.used.value
            else:
                _lib.check(_lib.load().mdil_nb_block_backward(C.byref(b), _stream()),
                           "mdil_nb_block_backward")
            if tail_partial is not None:
                tail.emitted(gx, tail_partial, nblk, tail_coef)
            res[0] = gx
            for i in range(17):
                if not need[i]:
                    res[i] = None
            return tuple(res)

        def conv_wgrad(taps, inp, gout, w, b):
            g = make_geom(N, H, W, H, W, taps, Cc, H, W, Cc)
            nt = len(taps)
            return wgrad(g, Cc, Cc, inp, None, gout, tuple(range(nt)), Cc * nt, nt, w, b)

        def pair_bwd(gz, a, inp, w31, b31, w13, b13, pw, pb, dil, n31, n13, npw, res_in, res_gate,
                     bnred=None):
            """Backward of  z = c13(relu(c31(inp))) [+ pw(inp)]  given gz = dL/dz.
            -> (dL/dinp [+ res_in gated by res_gate], dw31, db31, dw13, db13, dpw, dpb)."""
            dw31 = db31 = dw13 = db13 = dpw = dpb = None
            if n13 and pw is not None and npw:
                # one launch: 3 taps of the 1x3 (source a) + the adapter as 4th tap (source inp)
                G4 = make_geom(N, H, W, H, W, _taps_1x3(dil) + [(0, 0, 1)], Cc, H, W, Cc)
                dw13, db13, dpw, dpb = wgrad(G4, Cc, Cc, a, inp, gz, (0, 1, 2, 0), Cc * 3, 3, w13,
                                             b13, second=(1, Cc, 1, pw, pb))
            else:
                if n13:
                    dw13, db13 = conv_wgrad(_taps_1x3(dil), a, gz, w13, b13)
                if pw is not None and npw:
                    dpw, dpb = conv_wgrad([(0, 0, 0)], inp, gz, pw, pb)
            # dgrad through the 1x3 (taps mirrored), gated by relu(a):  ga = c13^T(gz) * (a > 0)
            G = make_geom(N, H, W, H, W, _taps_1x3(dil, flip=True), Cc, H, W, Cc)
            ga = tapconv(G, Cc, Cc, gz, None, pack_conv(w13, "dgrad"), torch.empty_like(gz), gate=a)
            if n31:
                dw31, db31 = conv_wgrad(_taps_3x1(dil), inp, ga, w31, b31)
            # dgrad through the 3x1 (+ adapter^T applied to gz as a 4th tap from source 1)
            taps = _taps_3x1(dil, flip=True)
            wpk = _pack_pair_dgrad(w31, pw)
            if pw is not None:
                taps = taps + [(0, 0, 1)]
            G = make_geom(N, H, W, H, W, taps, Cc, H, W, Cc)
            fused = None
            if bnred is not None and res_in is None and not BN_BWD_UNFUSED:
                # the result feeds a BatchNorm backward: gate it here and let the reductions of
                # that backward ride in this launch's epilogue (bnred = (relu source, BN input, coef))
                fused = tapconv_bnred(G, Cc, Cc, ga, gz, wpk, torch.empty_like(gz), *bnred)
            if fused is not None:
                ginp = fused
            else:
                ginp = tapconv(G, Cc, Cc, ga, gz, wpk, torch.empty_like(gz), res=res_in,
                               res_gate=res_gate)
            return ginp, dw31, db31, dw13, db13, dpw, dpb

        # second half:  out = relu(bn2(z2)*drop + x)
        gz2, dg2, dbe2 = bn_backward(gy, out, drop, z2, g2, be2, c2, need[15] or need[16])
        gu, dw31_2, db31_2, dw13_2, db13_2, dpw2, dpb2 = pair_bwd(
            gz2, a2, u, w31_2, b31_2, w13_2, b13_2, pw2, pb2, ctx.dil, need[9] or need[10], need[11] or need[12],
            need[13] or need[14], None, None, bnred=(u, z1, c1))
        # first half:  u = relu(bn1(z1));  the block input also receives gy * (out > 0)
        if isinstance(gu, tuple):       # (gated gradient, reductions): finalize + apply only
            g_, partial, nblk = gu
            gz1, dg1, dbe1 = bn_backward_partials(g_, z1, g1, be1, c1, need[7] or need[8], partial,
                                                  nblk, out=g_)
        else:
            gz1, dg1, dbe1 = bn_backward(gu, u, None, z1, g1, be1, c1, need[7] or need[8], out=gu)
        gx, dw31_1, db31_1, dw13_1, db13_1, dpw1, dpb1 = pair_bwd(
            gz1, a1, x, w31_1, b31_1, w13_1, b13_1, pw1, pb1, 1, need[1] or need[2], need[3] or need[4],
            need[5] or need[6], gy, out)
        res = [gx, dw31_1, db31_1, dw13_1, db13_1, dpw1, dpb1, dg1, dbe1,
               dw31_2, db31_2, dw13_2, db13_2, dpw2, dpb2, dg2, dbe2, None, None, None, None, None, None]
        for i in range(17):
            if not need[i]:
                res[i] = None
        return tuple(res)


# ----------------------------------------------------------------------------------------------
# UpsamplerBlock
# ----------------------------------------------------------------------------------------------
class UpFn(torch.autograd.Function):
    """relu(bn(convT 3x3 s2 p1 op1 (x))): one tap-conv launch per output parity class."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rm, rv, nbt, train, link_out):
        ctx.sink_slot = SINK_SLOT
        ctx.link_out = link_out
        _chk(x, "x")
        N, H, W, cin = x.shape
        cout = w.shape[1]
        z = torch.empty(N, 2 * H, 2 * W, cout, dtype=torch.float32, device=x.device)
        for a in (0, 1):
            for bb in (0, 1):
                taps, ktap = _class_taps(a, bb)
                g = make_geom(N, H, W, H, W, taps, cin, 2 * H, 2 * W, cout, ohs=2, oho=a, ows=2,
                              owo=bb)
                tapconv(g, cin, cout, x, None, pack_conv(w, "t_fwd", tuple(ktap)), z, bias=b)
        if train:
            coef = bn_train_stats(z, gamma, beta, rm, rv, nbt)
            y = bn_apply(z, coef[2], coef[3], relu=True)
            ctx.save_for_backward(x, w, b, gamma, beta, z, y, coef)
            _log_gates(y)
            _attach_tail(link_out, y, z, coef, gamma, beta)
        else:
            ec = bn_eval_coeffs(gamma, beta, rm, rv)
            y = bn_apply(z, ec[0], ec[1], relu=True, out=z)
        return y

    @staticmethod
    def backward(ctx, gy):
        global SINK_SLOT
        SINK_SLOT = ctx.sink_slot
        x, w, b, gamma, beta, z, y, coef = ctx.saved_tensors
        gy = gy.contiguous()
        N, H, W, cin = x.shape
        cout = w.shape[1]
        need = ctx.needs_input_grad
        gz, dgamma, dbeta = _bn_backward_maybe_fused(ctx.link_out, gy, y, z, gamma, beta, coef, need[3] or need[4])
        dw = db = gx = None
        if need[1] or need[2]:
            for a in (0, 1):
                for bb in (0, 1):
                    taps, ktap = _class_taps(a, bb)
                    g = make_geom(N, H, W, H, W, taps, cin, 2 * H, 2 * W, cout, ohs=2, oho=a, ows=2,
                                  owo=bb)
                    # each parity class owns distinct (kh,kw) taps of dW and a quarter of the
                    # pixels of dbias: all four accumulate into the same buffers
                    dw, db = wgrad(g, cin, cout, x, None, gz, tuple(ktap), 9, cout * 9, w, b, dw, db)
        if need[0]:
            taps = [(kh - 1, kw - 1, 0) for kh in range(3) for kw in range(3)]
            g = make_geom(N, H, W, 2 * H, 2 * W, taps, cout, H, W, cin, ihs=2, iws=2)
            gx = tapconv(g, cout, cin, gz, None, pack_conv(w, "t_dgrad"), torch.empty_like(x))
        return gx, dw, db, dgamma, dbeta, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# Decoder.output_conv : ConvTranspose2d(16, nc, 2, stride 2)
# ----------------------------------------------------------------------------------------------
def _r4(v):
    return (v + 3) // 4 * 4


class OutFn(torch.autograd.Function):
    """ConvTranspose2d(16, nc, 2, stride 2): four 1-tap launches (one per output parity class).
    A pixel's nc logits occupy a row of r4(nc) floats (28 for the 27-class head) so that every
    consumer keeps 16-byte aligned rows; the returned tensor is the [..., :nc] view of it."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.sink_slot = SINK_SLOT
        _chk(x, "x")
        N, H, W, cin = x.shape
        nc = w.shape[1]
        P = _r4(nc)
        y = torch.empty(N, 2 * H, 2 * W, P, dtype=torch.float32, device=x.device)
        if cin == 16 and w.is_contiguous() and b is not None:
            # one pass for the four output parity classes (x read once, pad entries written 0)
            _lib.check(_lib.load().mdil_outconv_fwd(_p(x), _p(w), _p(b), N, H, W, nc, P, _p(y),
                                                    _stream()), "mdil_outconv_fwd")
        else:
            if P != nc:
                y.zero_()                                          # the pad entries must read as 0
            for a in (0, 1):
                for bb in (0, 1):
                    g = make_geom(N, H, W, H, W, [(0, 0, 0)], cin, 2 * H, 2 * W, P, ohs=2, oho=a,
                                  ows=2, owo=bb)
                    tapconv(g, cin, nc, x, None, pack_conv(w, "t_fwd", (a * 2 + bb,)), y, bias=b)
        ctx.save_for_backward(x, w, b)
        return y if P == nc else y[..., :nc]

    @staticmethod
    def backward(ctx, gy):
        global SINK_SLOT
        SINK_SLOT = ctx.sink_slot
        x, w, b = ctx.saved_tensors
        N, H, W, cin = x.shape
        nc = w.shape[1]
        P = _r4(nc)
        if P == nc:
            gy = gy.contiguous()
        elif not (gy.stride(3) == 1 and gy.stride(2) == P and gy.stride(1) == 2 * W * P
                  and gy.stride(0) == 4 * H * W * P and gy.storage_offset() % 4 == 0):
            pad = torch.zeros(N, 2 * H, 2 * W, P, dtype=torch.float32, device=gy.device)
            pad[..., :nc].copy_(gy)                          # rows of P floats, pad entry zero
            gy = pad
        need = ctx.needs_input_grad
        dw = db = gx = None
        if need[1] or need[2]:
            for a in (0, 1):
                for bb in (0, 1):
                    g = make_geom(N, H, W, H, W, [(0, 0, 0)], cin, 2 * H, 2 * W, P, ohs=2, oho=a,
                                  ows=2, owo=bb)
                    dw, db = wgrad(g, cin, nc, x, None, gy, (a * 2 + bb,), 4, nc * 4, w, b, dw, db)
        if need[0]:
            taps = [(a, bb, 0) for a in (0, 1) for bb in (0, 1)]
            g = make_geom(N, H, W, 2 * H, 2 * W, taps, P, H, W, cin, ihs=2, iws=2)
            gx = tapconv(g, nc, cin, gy, None, pack_conv(w, "t_dgrad", k_pad=32),
                         torch.empty_like(x))
        return gx, dw, db


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def _nhwc_logits(t):
    """[N,C,H,W] logits whose storage is NHWC rows of `pitch` floats (what Net.forward returns:
    pitch = C, or 28 for the 27-class head) -> (tensor positioned at the first row, C, pitch).
    Anything else is re-laid out into such rows."""
    if t.dim() != 4:
        raise RuntimeError("logits must be 4-D")
    N, Cc, H, W = t.shape
    P = _r4(Cc)
    if (t.stride(1) == 1 and t.stride(3) == P and t.stride(2) == W * P and t.stride(0) == H * W * P
            and t.storage_offset() % 4 == 0):
        return t, Cc, P
    rows = torch.zeros(N, H, W, P, dtype=torch.float32, device=t.device)
    rows[..., :Cc].copy_(t.permute(0, 2, 3, 1))
    return rows[..., :Cc].permute(0, 3, 1, 2), Cc, P


def _grad_rows(like, Cc, P):
    """Gradient buffer with the same row layout, returned as the [N,C,H,W] view of it."""
    N, _, H, W = like.shape
    buf = torch.empty(N, H, W, P, dtype=torch.float32, device=like.device)
    return buf, (buf if P == Cc else buf[..., :Cc]).permute(0, 3, 1, 2)


_label_err = {}


def _label_counter(device):
    """Per-device int32 counter the loss / metric kernels bump for every label outside [0, C)."""
    t = _label_err.get(device.index)
    if t is None:
        t = _label_err[device.index] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def check_labels():
    """Raise if any loss / metric launch since the last call met a label outside [0, C) -- the
    reference raises a device assert at that point (an un-relabelled 255, another dataset's ids
    with the wrong --num-classes).  Reads a device counter, i.e. synchronises: the trainers call
    it where they read the losses anyway; ``MDIL_CHECK_LABELS=1`` checks after every loss call."""
    for t in _label_err.values():
        n = int(t.item())
        if n:
            t.zero_()
            raise RuntimeError(f"mdil: {n} target label(s) outside [0, num_classes) reached the loss / "
                               "metric kernels (un-relabelled ignore id? wrong --num-classes?)")


_EAGER_LABEL_CHECK = __import__("os").environ.get("MDIL_CHECK_LABELS") == "1"


def _chk_target(target, what):
    if target.dtype != torch.int64 or not target.is_cuda:
        raise RuntimeError(f"mdil {what}: target must be an int64 device tensor (got {target.dtype}, "
                           f"{target.device})")


class CEFn(torch.autograd.Function):
    """Weighted per-pixel cross entropy; the gradient kernel re-reads the logits and takes the
    upstream gradient as a device scalar (no host sync)."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        lib = _lib.load()
        x, Cc, P = _nhwc_logits(logits)
        npix = x.shape[0] * x.shape[2] * x.shape[3]
        _chk_target(target, "cross_entropy2d")
        _chk(weight, "class weights")
        if weight.numel() != Cc or target.numel() != npix:
            raise RuntimeError(f"mdil cross_entropy2d: {weight.numel()} class weights / {target.numel()} "
                               f"targets for logits with {Cc} classes and {npix} pixels")
        target = target.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        ws = workspace(lib.mdil_loss_workspace(npix), x.device)
        _lib.check(lib.mdil_ce_loss(_p(x), _p(target), _p(weight), npix, Cc, P, None, _p(loss), None,
                                    _p(_label_counter(x.device)), ws.data_ptr(), ws.numel(),
                                    _stream()), "mdil_ce_loss")
        if _EAGER_LABEL_CHECK:
            check_labels()
        ctx.save_for_backward(x, target, weight)
        ctx.cp = (Cc, P)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, target, weight = ctx.saved_tensors
        Cc, P = ctx.cp
        npix = x.shape[0] * x.shape[2] * x.shape[3]
        g = g.reshape(1).contiguous().float()
        buf, dx = _grad_rows(x, Cc, P)
        scratch = torch.empty(1, dtype=torch.float32, device=x.device)
        ws = workspace(lib.mdil_loss_workspace(npix), x.device)
        _lib.check(lib.mdil_ce_loss(_p(x), _p(target), _p(weight), npix, Cc, P, _p(g), _p(scratch),
                                    _p(buf), None, ws.data_ptr(), ws.numel(), _stream()),
                   "mdil_ce_loss")
        return dx, None, None


class KLDFn(torch.autograd.Function):
    """mean(t * (log t - p_s)) with p_s = softmax(student), t = softmax(teacher)  (the
    reference passes probabilities, not log-probabilities, to KLDivLoss)."""

    @staticmethod
    def forward(ctx, s_logits, t_logits):
        lib = _lib.load()
        s, Cc, P = _nhwc_logits(s_logits)
        t, Ct, Pt = _nhwc_logits(t_logits)
        if (Cc, P) != (Ct, Pt) or s.shape != t.shape:
            raise RuntimeError("kld_prob: student / teacher logits differ in shape")
        npix = s.shape[0] * s.shape[2] * s.shape[3]
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        ws = workspace(lib.mdil_loss_workspace(npix), s.device)
        _lib.check(lib.mdil_kld_loss(_p(s), _p(t), npix, Cc, P, None, _p(loss), None, ws.data_ptr(),
                                     ws.numel(), _stream()), "mdil_kld_loss")
        ctx.save_for_backward(s, t)
        ctx.cp = (Cc, P)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        s, t = ctx.saved_tensors
        Cc, P = ctx.cp
        npix = s.shape[0] * s.shape[2] * s.shape[3]
        g = g.reshape(1).contiguous().float()
        buf, ds = _grad_rows(s, Cc, P)
        scratch = torch.empty(1, dtype=torch.float32, device=s.device)
        ws = workspace(lib.mdil_loss_workspace(npix), s.device)
        _lib.check(lib.mdil_kld_loss(_p(s), _p(t), npix, Cc, P, _p(g), _p(scratch), _p(buf),
                                     ws.data_ptr(), ws.numel(), _stream()), "mdil_kld_loss")
        return ds, None


# ----------------------------------------------------------------------------------------------
# Decoder.output_conv fused with the loss (csrc/head.hip): the training step never materialises
# the logits.  x: NHWC decoder features [N,H,W,16]; w, b: the ConvTranspose2d(16, nc, 2, stride 2)
# parameters in the reference's layout.
# ----------------------------------------------------------------------------------------------
HEAD_FUSE = __import__("os").environ.get("MDIL_NO_HEADFUSE") is None


def _head_ws(lib, device):
    return workspace(lib.mdil_head_workspace(), device)


def _head_args(x, w, b):
    _chk(x, "head features")
    if x.dim() != 4 or x.shape[3] != 16 or w.dim() != 4 or w.shape[0] != 16 or tuple(w.shape[2:]) != (2, 2) \
            or not w.is_contiguous() or b is None or b.numel() != w.shape[1]:
        raise RuntimeError("mdil fused head: expects NHWC features [N,H,W,16] and ConvTranspose2d(16, nc, 2, 2) "
                           f"parameters (got x {tuple(x.shape)}, w {tuple(w.shape)})")
    return x.shape[0], x.shape[1], x.shape[2], w.shape[1]


class HeadCEFn(torch.autograd.Function):
    """CrossEntropyLoss2d(weight)(output_conv(x), target) (models/erfnet_RA_parallel.py:188 +
    train_new_task_step2.py:84-92,293); backward recomputes the logits from x.
    -> loss, or (loss, logits [N,nc,2H,2W] view) with ``want_logits`` (e.g. --iouTrain)."""

    @staticmethod
    def forward(ctx, x, w, b, target, weight, want_logits):
        lib = _lib.load()
        ctx.sink_slot = SINK_SLOT
        N, H, W, nc = _head_args(x, w, b)
        _chk_target(target, "head_ce")
        _chk(weight, "class weights")
        if weight.numel() != nc or target.numel() != N * 4 * H * W:
            raise RuntimeError(f"mdil head_ce: {weight.numel()} class weights / {target.numel()} targets for "
                               f"{nc} classes and {N * 4 * H * W} output pixels")
        target = target.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=x.device)          # loss, wsum
        logits = torch.empty(N, 2 * H, 2 * W, _r4(nc), dtype=torch.float32, device=x.device) if want_logits else None
        ws = _head_ws(lib, x.device)
        _lib.check(lib.mdil_head_ce(_p(x), _p(w), _p(b), N, H, W, nc, _p(target), _p(weight), None,
                                    out.data_ptr(), out.data_ptr() + 4, None, None, None, 0, _p(logits),
                                    _p(_label_counter(x.device)), ws.data_ptr(), ws.numel(), _stream()),
                   "mdil_head_ce")
        if _EAGER_LABEL_CHECK:
            check_labels()
        ctx.save_for_backward(x, w, b, target, weight, out)
        if not want_logits:
            return out[0]
        lg = (logits if _r4(nc) == nc else logits[..., :nc]).permute(0, 3, 1, 2)
        ctx.mark_non_differentiable(lg)
        return out[0], lg

    @staticmethod
    def backward(ctx, g, *unused):
        global SINK_SLOT
        SINK_SLOT = ctx.sink_slot
        lib = _lib.load()
        x, w, b, target, weight, out = ctx.saved_tensors
        N, H, W, nc = _head_args(x, w, b)
        need = ctx.needs_input_grad
        g = g.reshape(1).contiguous().float()
        gx = torch.empty_like(x)
        tw = tb = None
        sunk = True
        if need[1] or need[2]:
            tw, tb, sunk = _grad_target(w, b, nc, None, None)
        ws = _head_ws(lib, x.device)
        _lib.check(lib.mdil_head_ce(_p(x), _p(w), _p(b), N, H, W, nc, _p(target), _p(weight), _p(g), None,
                                    out.data_ptr() + 4, _p(gx), _p(tw), _p(tb), 1, None, None,
                                    ws.data_ptr(), ws.numel(), _stream()), "mdil_head_ce")
        return (gx if need[0] else None, None if sunk else tw, None if sunk else tb, None, None, None)


class HeadKLDFn(torch.autograd.Function):
    """KLDivLoss()(softmax(output_conv_s(xs)), softmax(output_conv_t(xt))) -- probabilities as the
    input, the reference's quirk (train_new_task_step2.py:241,296-297); gradients flow to the
    student side only (the previous model is frozen)."""

    @staticmethod
    def forward(ctx, xs, ws_, bs, xt, wt, bt):
        lib = _lib.load()
        ctx.sink_slot = SINK_SLOT
        N, H, W, nc = _head_args(xs, ws_, bs)
        if _head_args(xt, wt, bt) != (N, H, W, nc):
            raise RuntimeError("mdil head_kld: student / teacher heads differ in shape")
        loss = torch.empty(1, dtype=torch.float32, device=xs.device)
        ws = _head_ws(lib, xs.device)
        _lib.check(lib.mdil_head_kld(_p(xs), _p(ws_), _p(bs), _p(xt), _p(wt), _p(bt), N, H, W, nc, None,
                                     _p(loss), None, None, None, 0, ws.data_ptr(), ws.numel(), _stream()),
                   "mdil_head_kld")
        ctx.save_for_backward(xs, ws_, bs, xt, wt, bt)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        global SINK_SLOT
        SINK_SLOT = ctx.sink_slot
        lib = _lib.load()
        xs, ws_, bs, xt, wt, bt = ctx.saved_tensors
        N, H, W, nc = _head_args(xs, ws_, bs)
        need = ctx.needs_input_grad
        g = g.reshape(1).contiguous().float()
        gx = torch.empty_like(xs)
        tw = tb = None
        sunk = True
        if need[1] or need[2]:
            tw, tb, sunk = _grad_target(ws_, bs, nc, None, None)
        ws = _head_ws(lib, xs.device)
        _lib.check(lib.mdil_head_kld(_p(xs), _p(ws_), _p(bs), _p(xt), _p(wt), _p(bt), N, H, W, nc, _p(g),
                                     None, _p(gx), _p(tw), _p(tb), 1, ws.data_ptr(), ws.numel(), _stream()),
                   "mdil_head_kld")
        return (gx if need[0] else None, None if sunk else tw, None if sunk else tb, None, None, None)


def head_ce(x, w, b, target, weight, want_logits=False):
    return HeadCEFn.apply(x, w, b, target, weight, want_logits)


def head_kld(xs, ws, bs, xt, wt, bt):
    return HeadKLDFn.apply(xs, ws, bs, xt, wt, bt)


def cross_entropy2d(logits, target, weight):
    return CEFn.apply(logits, target, weight)


def kld_prob(student_logits, teacher_logits):
    return KLDFn.apply(student_logits, teacher_logits)


def argmax_confusion(logits, target, ignore, counts):
    """counts: int64 [3][C] (tp, fp, fn), accumulated in place."""
    lib = _lib.load()
    x, Cc, P = _nhwc_logits(logits)
    npix = x.shape[0] * x.shape[2] * x.shape[3]
    _chk_target(target, "argmax_confusion")
    if target.numel() != npix or counts.dtype != torch.int64 or counts.numel() != 3 * Cc:
        raise RuntimeError("mdil argmax_confusion: target / counts do not match the logits")
    _lib.check(lib.mdil_argmax_confusion(_p(x), _p(target.contiguous()), npix, Cc, P, ignore,
                                         _p(counts), _p(_label_counter(x.device)), _stream()),
               "mdil_argmax_confusion")
    if _EAGER_LABEL_CHECK:
        check_labels()
    return counts


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0, grad_scale=1.0):
    """torch.optim.Adam (L2) on flat fp32 segments; ``step`` is the 1-based step count."""
    lib = _lib.load()
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _lib.check(lib.mdil_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(),
                                  lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale,
                                  _stream()), "mdil_adam_step")


# ----------------------------------------------------------------------------------------------
# input pipeline, device half
# ----------------------------------------------------------------------------------------------
def to_nhwc(images):
    """[N,C,H,W] float images (the reference's ToTensor layout) -> dense NHWC fp32 for the stem.  A tensor
    that already sits in channels-last storage (``augment_batch`` returns such views) is passed through."""
    if not images.is_cuda:
        raise RuntimeError("mdil to_nhwc: input must be a device tensor (no CPU path)")
    x = images.permute(0, 2, 3, 1)
    if images.dtype == torch.float32 and x.is_contiguous():
        return x
    if images.dtype != torch.float32 or not images.is_contiguous() or images.shape[1] > 32 or \
            (images.requires_grad and torch.is_grad_enabled()):
        return x.contiguous().float()         # (ATen: differentiable -- a gradient with respect to the input image)
    N, Cc, H, W = images.shape
    out = torch.empty(N, H, W, Cc, dtype=torch.float32, device=images.device)
    _lib.check(_lib.load().mdil_nchw_to_nhwc(images.data_ptr(), N, Cc, H, W, out.data_ptr(), _stream()),
               "mdil_nchw_to_nhwc")
    return out


def dropout_factors(uniform, keep, inv_keep):
    """One uniform draw -> Dropout2d factors of every encoder block (kept: 1 / (1 - p), dropped: 0)."""
    out = torch.empty_like(uniform)
    _lib.check(_lib.load().mdil_dropout_factors(uniform.data_ptr(), keep.data_ptr(), inv_keep.data_ptr(),
                                                out.data_ptr(), uniform.numel(), _stream()), "mdil_dropout_factors")
    return out



def augment_batch(img_u8, lab_u8, params, num_classes):
    """MyCoTransform's flip / shift / ToTensor / ToLabel / Relabel(255 -> num_classes-1) for a whole
    batch (train_new_task_step2.py:59-79).  img_u8 [N,H,W,3] uint8, lab_u8 [N,H,W] uint8,
    params [N,3] int32 (hflip, transX, transY) -- all on the device.
    -> (images f32 [N,3,H,W] as a view of NHWC storage, labels i64 [N,1,H,W])."""
    lib = _lib.load()
    if not (img_u8.is_cuda and lab_u8.is_cuda and params.is_cuda):
        raise RuntimeError("mdil augment_batch: inputs must be device tensors (no CPU path)")
    if img_u8.dtype != torch.uint8 or lab_u8.dtype != torch.uint8 or params.dtype != torch.int32:
        raise RuntimeError("mdil augment_batch: expected uint8 image / uint8 label / int32 params")
    N, H, W, _ = img_u8.shape
    img_u8, lab_u8, params = img_u8.contiguous(), lab_u8.contiguous(), params.contiguous()
    out = torch.empty(N, H, W, 3, dtype=torch.float32, device=img_u8.device)
    lab = torch.empty(N, 1, H, W, dtype=torch.int64, device=img_u8.device)
    _lib.check(lib.mdil_augment_batch(img_u8.data_ptr(), lab_u8.data_ptr(), params.data_ptr(), N, H,
                                      W, 255, num_classes - 1, out.data_ptr(), lab.data_ptr(),
                                      _stream()), "mdil_augment_batch")
    return out.permute(0, 3, 1, 2), lab
