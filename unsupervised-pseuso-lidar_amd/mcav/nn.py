"""Host-side building blocks of the MI355X networks: ctypes mirrors of include/mcav_conv.h and thin op wrappers.

Everything here launches HIP kernels through the C ABI on torch's current stream.  Activations are NHWC fp32
torch tensors ([B, H, W, C]); parameters keep the reference's shapes (OIHW conv weights) so state_dicts are
interchangeable, and are re-packed to the kernels' [Np][taps][Kp] form whenever they change.
"""
import ctypes
import os
import os as _os

import torch

from . import lib as L

c_p, c_i, c_f, c_sz, c_d = L.c_p, L.c_i, L.c_f, L.c_sz, ctypes.c_double

G_DIRECT, G_SMALLC, G_ADJ_REFLECT, G_ADJ_STRIDE2 = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3
DACT_AFTER_ADDEND = 0x100       # (conv + addend) * act'(aux) instead of conv * act'(aux) + addend


class IgemmDesc(ctypes.Structure):
    _fields_ = [("x1", c_p), ("x2", c_p), ("B", c_i), ("Hs", c_i), ("Ws", c_i), ("C1", c_i), ("C2", c_i), ("up1", c_i),
                ("w", c_p), ("kh", c_i), ("kw", c_i), ("Np", c_i), ("Kp", c_i),
                ("mode", c_i), ("stride", c_i), ("sign", c_i), ("offset", c_i), ("pad_mode", c_i),
                ("y", c_p), ("Hd", c_i), ("Wd", c_i), ("Cd", c_i), ("n_begin", c_i), ("n_count", c_i), ("y_choff", c_i),
                ("bias", c_p), ("act", c_i), ("dact_aux", c_p), ("dact", c_i), ("addend", c_p), ("pool", c_i), ("stats", c_p),
                ("tile", c_i), ("groups", c_i), ("w_upmerge", c_p), ("mma", c_i), ("w16", c_p),
                ("stats_x", c_p), ("stats_mean", c_p), ("stats_invstd", c_p), ("w_stem", c_p)]


class WgradDesc(ctypes.Structure):
    _fields_ = [("x1", c_p), ("x2", c_p), ("B", c_i), ("Hs", c_i), ("Ws", c_i), ("C1", c_i), ("C2", c_i), ("up1", c_i),
                ("kh", c_i), ("kw", c_i), ("Kp", c_i),
                ("mode", c_i), ("stride", c_i), ("sign", c_i), ("offset", c_i), ("pad_mode", c_i),
                ("dy", c_p), ("Hd", c_i), ("Wd", c_i), ("Cdy", c_i), ("dy_choff", c_i), ("Cout", c_i), ("Cin", c_i),
                ("dw_oihw", c_p), ("accumulate", c_i), ("dbias", c_p), ("tile", c_i), ("upm", c_i), ("Cin_total", c_i), ("ci_offset", c_i),
                ("mma", c_i)]


class WgradReduceItem(ctypes.Structure):
    """include/mcav_conv.h: mcav_wgrad_reduce_item (a pending slab reduction of one weight-gradient launch)."""
    _fields_ = [("slab", c_p), ("pre", c_p), ("dw", c_p), ("dbias", c_p), ("elems", ctypes.c_ulonglong)] + \
               [(n, c_i) for n in ("splits", "groups", "per_group", "Ktot", "slabN", "Kp", "taps", "Cout", "Cin", "ci_t", "accumulate", "upm", "cin_total",
                                   "ci_off", "pre_bx", "red_gx", "red_gy", "pre_first", "red_first", "reserved")]


L.register({
    "mcav_wgrad_deferred": (c_i, [ctypes.POINTER(WgradDesc), c_p, c_sz, ctypes.POINTER(WgradReduceItem), c_p]),
    "mcav_wgrad_reduce_plan": (c_i, [ctypes.POINTER(WgradReduceItem), c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_sz)]),
    "mcav_wgrad_reduce_multi": (c_i, [c_p, c_i, c_i, c_i, c_sz, c_p]),
    "mcav_igemm_mtiles": (c_i, [ctypes.POINTER(IgemmDesc)]),
    "mcav_igemm": (c_i, [ctypes.POINTER(IgemmDesc), c_p]),
    "mcav_igemm_uses_bf16": (c_i, [ctypes.POINTER(IgemmDesc)]),
    "mcav_wgrad_uses_bf16": (c_i, [ctypes.POINTER(WgradDesc)]),
    "mcav_f32_to_bf16": (c_i, [c_p, c_p, c_sz, c_p]),
    "mcav_f32_to_bf16_planes": (c_i, [c_p, c_p, c_sz, c_p]),
    "mcav_pack_stem_weights": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    "mcav_wgrad_workspace_bytes": (c_sz, [ctypes.POINTER(WgradDesc)]),
    "mcav_wgrad": (c_i, [ctypes.POINTER(WgradDesc), c_p, c_sz, c_p]),
    "mcav_pack_weights": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "mcav_pack_weights_multi": (c_i, [c_p, c_i, c_i, c_p]),
    "mcav_pack_weights_blocks": (c_i, [c_i, c_i, c_i, c_i]),
    "mcav_pack_weights_upmerge": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "mcav_pack_weights_upmerge_adj": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "mcav_upsample_adj_fold": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p]),
    "mcav_nchw_to_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "mcav_nchw3_to_nhwc": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "mcav_nhwc_to_nchw": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "mcav_bn_finalize": (c_i, [c_p, c_i, c_i, c_d, c_p, c_p, c_f, c_f] + [c_p] * 6 + [c_i, c_p, c_sz, c_p]),
    "mcav_bn_finalize_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "mcav_bn_eval_coeffs": (c_i, [c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p]),
    "mcav_bn_apply": (c_i, [c_p, c_p, c_p, c_p, c_i, c_sz, c_i, c_p, c_sz, c_p]),
    "mcav_bn_bwd_workspace_bytes": (c_sz, [c_sz, c_i, c_i]),
    "mcav_bn_bwd_reduce": (c_i, [c_p] * 5 + [c_i, c_sz, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_sz, c_p]),
    "mcav_bn_bwd_apply": (c_i, [c_p] * 7 + [c_i, c_sz, c_i, c_p, c_p, c_i, c_i, c_p]),
    "mcav_bn_bwd_finalize": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p]),
    "mcav_maxpool3s2_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "mcav_maxpool3s2_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "mcav_act_bwd": (c_i, [c_p, c_p, c_i, c_sz, c_p, c_i, c_p]),
    "mcav_act_bwd_strided": (c_i, [c_p, c_p, c_i, c_sz, c_p, c_i, c_p]),
    "mcav_conv3x3r_c1_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p]),
    "mcav_conv3x3r_c1_bwd_workspace_bytes": (c_sz, [c_i]),
    "mcav_conv3x3r_c1_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_sz, c_p]),
    "mcav_add": (c_i, [c_p, c_p, c_sz, c_p, c_p]),
    "mcav_spatial_mean": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "mcav_spatial_mean_bwd": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "mcav_upsample_nearest2x": (c_i, [c_p, c_sz, c_i, c_i, c_p, c_p]),
    "mcav_upsample_nearest2x_bwd": (c_i, [c_p, c_sz, c_i, c_i, c_p, c_p]),
    "mcav_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_f, c_f, c_f, c_f, c_i, c_f, c_p]),
    "mcav_adam_step_dev": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_f, c_f, c_f, c_p, c_p]),
    "mcav_event_create": (c_i, [ctypes.POINTER(c_p)]),
    "mcav_event_destroy": (c_i, [c_p]),
    "mcav_event_record_external": (c_i, [c_p, c_p]),
    "mcav_event_wait_external": (c_i, [c_p, c_p]),
    "mcav_stream_wait_event": (c_i, [c_p, c_p]),
    "mcav_kernel_timer_begin": (c_i, []),
    "mcav_kernel_timer_count": (c_i, []),
    "mcav_kernel_timer_end": (c_i, [c_p, c_i]),
})


def kernel_timer_begin():
    L.check(L.lib().mcav_kernel_timer_begin(), "mcav_kernel_timer_begin")


def kernel_timer_end():
    """-> per-dispatch durations (ms) of the conv kernels launched since kernel_timer_begin(), in launch order."""
    n = L.lib().mcav_kernel_timer_count()
    buf = (ctypes.c_float * max(1, n))()
    got = L.lib().mcav_kernel_timer_end(buf, n)
    if got < 0:
        L.check(got, "mcav_kernel_timer_end")
    return [float(buf[i]) for i in range(min(n, got))]


# bench.py's instrumented step: when PROFILE is a list (and mcav_kernel_timer_begin() was called) every conv kernel is dispatched with its
# own start/stop event pair inside the library (csrc/kernel_timer.h); a record notes which of those dispatches a call issued
PROFILE = None
PROFILE_LOSS = None
PROFILE_TAGS = None


import collections as _collections
ProfileRec = _collections.namedtuple("ProfileRec", "kind flops i0 i1 executed bf16_planes abytes")


class _Timed:
    """flops: the reference's algorithmic FLOPs of the launch (2 M N K of the convolution it stands for).  executed: what the MFMA pipe really
    does -- less for the merged-tap forms (4 taps instead of 9 on the upsampled half), more where K is padded (the stem's 168 k for 147)."""

    def __init__(self, kind, flops, tag="", executed=None, bf16_planes=0, abytes=0.0):
        """bf16_planes: plane products per algorithmic product when the launch runs on the bf16 MFMA pipe (1 = plain bf16, 6 = the fp32
        contraction on split operands), 0 = an fp32-MFMA (or non-MFMA) launch.  abytes: the launch's ALGORITHMIC bytes -- every operand
        read once, the result written once, fp32 (the per-launch roofline is min(MFMA peak, flops / abytes x HBM peak))."""
        self.kind, self.flops, self.tag = kind, flops, tag
        self.executed = flops if executed is None else executed
        self.bf16_planes, self.abytes = bf16_planes, abytes

    def __enter__(self):
        if PROFILE is not None:
            self.i0 = L.lib().mcav_kernel_timer_count()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            PROFILE.append(ProfileRec(self.kind, self.flops, self.i0, L.lib().mcav_kernel_timer_count(), self.executed, self.bf16_planes, self.abytes))
            if PROFILE_TAGS is not None:
                PROFILE_TAGS.append(self.tag)
        return False


def up16(v):
    return (v + 15) // 16 * 16


def empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def P(t):
    return t.data_ptr() if t is not None else None


# ------------------------------------------------------------------------------------------------ conv parameters
class _PackItem(ctypes.Structure):
    _fields_ = [("src", c_p), ("dst", c_p), ("Cout", c_i), ("Cin", c_i), ("taps", c_i), ("transposed", c_i), ("Np", c_i), ("Kp", c_i),
                ("Kstride", c_i), ("first_block", c_i)]


MMA_FP32, MMA_BF16, MMA_SPLIT, MMA_SPLIT_ALL = 0, 1, 2, 3      # (3: the split form on every launch the bf16 kernels cover, not only where it pays: tests)
# What the networks' convolutions run on when nothing else is asked for (round 4): fp32 arithmetic with the kernel chosen per launch -- the
# 3x3 stride-1 zero-padded convolutions of the ResNet trunk and their data gradients as fp32 contractions on split operands (six bf16-MFMA
# plane products per fp32 product, fp32 accumulation: results at least as close to float64 as the fp32 MFMA's, tests/test_split_gpu.py),
# every other launch on the fp32 MFMA kernels.  set_compute_dtype(m, "fp32-mfma") pins every launch to v_mfma_f32_32x32x2_f32.
# A ConvSpec built directly (kernel-level tests, micro-benchmarks) stays on MMA_FP32 unless its .mma is set.
DEFAULT_MMA = MMA_SPLIT
# kind -> (transposed, bf16 planes: 0 = an fp32 copy, 1 = bf16, 3 = the planes h, m, l of the fp32 contraction on split operands)
_KINDS = {"f": (False, 0), "b": (True, 0), "f16": (False, 1), "b16": (True, 1), "f16s": (False, 3), "b16s": (True, 3)}


class PackRegistry:
    """Every packed filter copy that has been requested so far.  After an optimiser step all of them are stale; the first one
    that is asked for re-derives ALL of them with one mcav_pack_weights_multi launch (instead of ~80 small ones)."""

    def __init__(self):
        self.entries = []          # (weakref to spec, kind)
        self.table = None
        self.retired = []
        self.nblocks = 0
        self.signature = None

    def add(self, spec, kind):
        import weakref
        self.entries.append((weakref.ref(spec), kind))
        self.signature = None          # the next repack rebuilds the table

    def _build(self, device, live):
        items = (_PackItem * len(live))()
        blk = 0
        for it, (spec, kind) in zip(items, live):
            if kind in ("um", "ua"):                   # merged-tap copies of the decoder's upsample convolutions: C1 travels in `taps`
                buf = spec._packs[kind]
                it.src, it.dst = spec.weight.data_ptr(), buf.data_ptr()
                it.Cout, it.Cin, it.taps = spec.cout, spec.cin, spec._merge_c1
                if kind == "um":
                    it.transposed, it.Np, it.Kp = 8, spec.np, 0
                else:
                    it.transposed, it.Np, it.Kp = 16, buf.shape[0], buf.shape[1] // 16
                it.Kstride, it.first_block = 0, blk
                blk += L.lib().mcav_pack_weights_blocks(it.taps, it.transposed, it.Np, max(it.Kp, 1))
                continue
            tr, h = _KINDS[kind]
            buf = spec._packs[kind]
            np_, kp_ = (up16(spec.cin), up16(spec.cout)) if tr else (spec.np, spec.kp)
            it.src, it.dst = spec.weight.data_ptr(), buf.data_ptr()
            it.Cout, it.Cin, it.taps, it.transposed = spec.cout, spec.cin, spec.kh * spec.kw, int(tr) | (4 if h == 3 else 2 if h else 0)
            it.Np, it.Kp, it.Kstride, it.first_block = np_, kp_, buf.shape[1], blk      # (buf.shape[0] = Np rows per plane x planes)
            blk += L.lib().mcav_pack_weights_blocks(it.taps, int(tr), np_, kp_)
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        if self.table is not None:
            self.retired.append(self.table)      # a captured step's re-packing launch reads the table it was captured with: never freed
        self.table = raw.to(device)
        self.nblocks = blk

    def repack_all(self, device):
        self.entries = [(r, k) for r, k in self.entries if r() is not None]          # drop specs of deleted modules
        live = [(r(), k) for r, k in self.entries]
        live = [(s, k) for s, k in live if s is not None and s.weight.device == device and s._packs.get(k) is not None]
        if not live:
            return
        sig = tuple((s.weight.data_ptr(), s._packs[k].data_ptr()) for s, k in live)
        if self.table is None or sig != self.signature or self.table.device != device:
            self._build(device, live)
            self.signature = sig
        L.check(L.lib().mcav_pack_weights_multi(P(self.table), len(live), self.nblocks, L.stream()), "mcav_pack_weights_multi")
        for s, k in live:
            s._keys[k] = s._key()


PACKS = PackRegistry()


def refresh_packed_weights(device):
    """Re-derive every stale packed filter copy NOW, on the current stream.  Call before forking work onto other streams: the lazy
    refresh inside packed_fwd()/packed_bwd() runs on whichever stream asks first, which the other streams would not wait for."""
    for ref, k in PACKS.entries:
        s = ref()
        if s is None or s.weight.device != device:
            continue
        if s._keys.get(k) != s._key():
            PACKS.repack_all(device)
            return


class ConvSpec:
    """Static description of one convolution + the packed copies of its weight (kept fresh lazily)."""

    def __init__(self, weight, bias, stride, pad, pad_mode, smallc=False):
        self.weight, self.bias = weight, bias           # nn.Parameters, OIHW / [Cout]
        self.cout, self.cin, self.kh, self.kw = weight.shape
        self.stride, self.pad, self.pad_mode = stride, pad, pad_mode
        self.smallc = smallc                             # the 3-channel image stem: source is NHWC4
        self.kp = 4 if smallc else up16(self.cin)        # K padding of the forward filter
        self.np = up16(self.cout)
        self.mma = MMA_FP32                              # MMA_BF16: eligible launches run the bf16 MFMA kernels (set_compute_dtype)
        self._packs, self._keys = {}, {}                 # kind ("f", "b", "f16", "b16") -> packed copy / the weight version it was made from

    def _key(self):
        w = self.weight
        return (w.data_ptr(), w._version, getattr(w, "_mcav_epoch", lambda: 0)())

    def _packed(self, kind):
        tr, h = _KINDS[kind]
        key = self._key()
        buf = self._packs.get(kind)
        if buf is None or self._keys.get(kind) != key or buf.device != self.weight.device:
            taps = self.kh * self.kw
            if buf is None or buf.device != self.weight.device:
                # first use: derive this one copy (a full repack per new copy would cost ~80 whole-registry launches in the first step)
                shape = (up16(self.cin), taps * up16(self.cout)) if tr else (self.np, up16(taps * self.kp))
                buf = torch.empty((shape[0] * max(h, 1), shape[1]), dtype=torch.bfloat16 if h else torch.float32, device=self.weight.device)
                f32 = torch.empty(shape, dtype=torch.float32, device=self.weight.device) if h else buf
                np_, kp_ = (up16(self.cin), up16(self.cout)) if tr else (self.np, self.kp)
                L.check(L.lib().mcav_pack_weights(P(self.weight), self.cout, self.cin, self.kh, self.kw, int(tr), P(f32), np_, kp_, L.stream()),
                        "mcav_pack_weights")
                if h == 3:
                    L.check(L.lib().mcav_f32_to_bf16_planes(P(f32), P(buf), f32.numel(), L.stream()), "mcav_f32_to_bf16_planes")
                elif h:
                    L.check(L.lib().mcav_f32_to_bf16(P(f32), P(buf), f32.numel(), L.stream()), "mcav_f32_to_bf16")
                self._packs[kind] = buf
                self._keys[kind] = key
                PACKS.add(self, kind)
            else:
                PACKS.repack_all(self.weight.device)       # one launch refreshes every registered copy (this one included)
        return self._packs[kind]

    def packed_fwd(self):
        return self._packed("f")

    def packed_bwd(self):
        """Data-gradient filter: rows = input channels (padded to 16), K = output channels (padded to 16)."""
        return self._packed("b")

    def packed_fwd16(self):
        return self._packed("f16")

    def packed_bwd16(self):
        return self._packed("b16")

    def packed_fwd16s(self):
        """[3][Np][Kstride] bf16: the planes h, m, l of the forward filter (MMA_SPLIT)."""
        return self._packed("f16s")

    def packed_bwd16s(self):
        return self._packed("b16s")

    def is_stem(self):
        """The two 7x7 stride-2 stems the patch-in-LDS kernels of csrc/conv_stem.hip cover: the depth net's image stem and PoseNet conv1."""
        return ((self.kh, self.kw, self.stride, self.pad, self.pad_mode) == (7, 7, 2, 3, PAD_ZERO)
                and ((self.smallc and (self.cout, self.cin) == (64, 3)) or (not self.smallc and (self.cout, self.cin) == (16, 9))))

    def packed_stem(self):
        """[Cin * 56][Cout] copy of a stem filter in the K order of csrc/conv_stem.hip (mcav_pack_stem_weights); one tiny launch when stale."""
        key = self._key()
        if getattr(self, "_stem", None) is None or self._key_s != key or self._stem.device != self.weight.device:
            if getattr(self, "_stem", None) is None or self._stem.device != self.weight.device:
                self._stem = empty((self.cin * 56, self.cout), self.weight)
            L.check(L.lib().mcav_pack_stem_weights(P(self.weight), self.cout, self.cin, P(self._stem), L.stream()), "mcav_pack_stem_weights")
            self._key_s = key
        return self._stem

    def packed_upmerge(self, c1):
        """Merged-tap copy for conv(cat(up2(x1), x2)) with reflection padding: [4 classes][Np][4 taps][c1] pre-summed filters of the
        first c1 input channels (mcav_pack_weights_upmerge).  First use: one small launch; afterwards a record of the registry's ONE re-packing
        launch per optimiser step (round 4)."""
        return self._merged("um", c1, lambda: empty((4, self.np, 4, c1), self.weight),
                            lambda buf: L.lib().mcav_pack_weights_upmerge(P(self.weight), self.cout, self.cin, c1, P(buf), self.np, L.stream()))

    def _merged(self, kind, c1, make, pack):
        key = self._key()
        buf = self._packs.get(kind)
        if buf is None or buf.device != self.weight.device or getattr(self, "_merge_c1", c1) != c1:
            if buf is not None and getattr(self, "_merge_c1", c1) != c1:
                raise L.MCAVError("a convolution's merged-tap copies are built for ONE split of its input channels (got %d after %d)" % (c1, self._merge_c1))
            self._merge_c1 = c1
            first = buf is None
            buf = self._packs[kind] = make()
            L.check(pack(buf), "mcav_pack_weights_upmerge(%s)" % kind)
            self._keys[kind] = key
            if first:
                PACKS.add(self, kind)
        elif self._keys.get(kind) != key:
            PACKS.repack_all(self.weight.device)       # one launch refreshes every registered copy (this one included)
        return self._packs[kind]

    def packed_upmerge_adj(self, c1, bf16=False):
        """[up16(c1)][16 taps][up16(cout)]: the 4x4 stride-2 filter of the pooled upsample adjoint (mcav_pack_weights_upmerge_adj);
        bf16: its rounded copy for the bf16 MFMA kernel (mcav_f32_to_bf16)."""
        npd, kpd = up16(c1), up16(self.cout)
        upa = self._merged("ua", c1, lambda: empty((npd, 16 * kpd), self.weight),
                           lambda buf: L.lib().mcav_pack_weights_upmerge_adj(P(self.weight), self.cout, self.cin, c1, P(buf), npd, kpd, L.stream()))
        self._upa = upa
        key = self._key() + (c1,)
        if not bf16:
            return self._upa
        planes = 3 if bf16 == 3 else 1                      # bf16 = 3: the planes h, m, l (MMA_SPLIT)
        if getattr(self, "_upa16", None) is None or self._key_ua16 != key or self._upa16.shape[0] != planes * self._upa.shape[0]:
            if getattr(self, "_upa16", None) is None or self._upa16.shape[0] != planes * self._upa.shape[0]:
                self._upa16 = torch.empty((planes * self._upa.shape[0], self._upa.shape[1]), dtype=torch.bfloat16, device=self._upa.device)
            if planes == 3:
                L.check(L.lib().mcav_f32_to_bf16_planes(P(self._upa), P(self._upa16), self._upa.numel(), L.stream()), "mcav_f32_to_bf16_planes")
            else:
                L.check(L.lib().mcav_f32_to_bf16(P(self._upa), P(self._upa16), self._upa.numel(), L.stream()), "mcav_f32_to_bf16")
            self._key_ua16 = key
        return self._upa16


# MMA_SPLIT weight gradients: 1 = every launch the bf16 kernels cover on the split kernels (wgrad_bf16_kernel<3>); 0 = the patch kernel where it
# applies, the fp32-MFMA kernel elsewhere (all are fp32 results)
SPLIT_WGRAD = _os.environ.get("MCAV_SPLIT_WGRAD", "0") != "0"
SPLIT_NAMES = ("fp32-split", "fp32_split", "f32s", "fp32s")
MFMA32_NAMES = ("fp32-mfma", "fp32_mfma", "f32-mfma")


def set_compute_dtype(module, dtype):
    """What the convolutions of `module` compute in.
    torch.float32 / "fp32" / None: the default -- fp32 results, kernel chosen per launch (DEFAULT_MMA: the trunk's 3x3 stride-1 convolutions and
    their data gradients as fp32 contractions on split operands, everything else on the fp32 MFMA).
    "fp32-mfma": every launch on the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32).
    "fp32-split": the same launches as the default, named explicitly (mcav_igemm_desc.mma = 2): every operand element split into three bf16
    planes, six plane products accumulated in fp32 -- fp32 results (at least as exact as the fp32 MFMA's) at 6 / 16 of its MFMA time.
    torch.bfloat16 / "bf16": opt-in bf16 MFMA conv tiles (BASELINE.json configs[2] / [4]): the launches the bf16 kernels cover
    (csrc/conv_bf16.hip) take bf16-rounded operands with fp32 accumulation.
    Master weights, activations in HBM, BatchNorm statistics and gradients stay fp32 in every mode."""
    bf16 = dtype in (torch.bfloat16, "bf16", "bfloat16")
    split = dtype in SPLIT_NAMES
    pinned = dtype in MFMA32_NAMES
    if not bf16 and not split and not pinned and dtype not in (torch.float32, "fp32", "f32", "float32", None):
        raise L.MCAVError("compute dtype must be fp32, fp32-mfma, fp32-split or bf16, got %r" % (dtype,))
    mma = MMA_BF16 if bf16 else MMA_SPLIT if split else MMA_FP32 if pinned else DEFAULT_MMA
    for m in module.modules():
        if hasattr(m, "weight") and getattr(m, "weight") is not None and m.weight.dim() == 4:
            m._mcav_mma = mma
            spec = getattr(m, "_mcav_spec", None)
            if spec is not None:
                spec.mma = m._mcav_mma
    return module


def stats_blocks_stem(spec, stats):
    return stats and not spec.smallc           # (only the image stem's kernel carries the BatchNorm statistics epilogue)


def _weights_for(spec, d, transposed):
    """Fills d.w / d.w16 / d.mma of an IgemmDesc whose geometry is already set: the bf16 copy where the launch runs on the bf16 kernels."""
    if spec.mma in (MMA_BF16, MMA_SPLIT, MMA_SPLIT_ALL):
        d.mma, d.w16, d.w = spec.mma, d.x1, d.x1          # placeholders: the eligibility test looks at the geometry only
        if L.lib().mcav_igemm_uses_bf16(ctypes.byref(d)):
            if spec.mma >= MMA_SPLIT:
                w16 = spec.packed_bwd16s() if transposed else spec.packed_fwd16s()
            else:
                w16 = spec.packed_bwd16() if transposed else spec.packed_fwd16()
            d.w16 = d.w = P(w16)                           # (w is never read on the bf16 path; it only has to be non-null)
            return
    d.mma, d.w16 = 0, None
    d.w = P(spec.packed_bwd() if transposed else spec.packed_fwd())


def _planes(d):
    """Plane products per algorithmic product of a filled IgemmDesc / WgradDesc: 6 on the split form, 1 on plain bf16, 0 on the fp32 kernels."""
    return 6 if d.mma >= MMA_SPLIT else 1 if d.mma == MMA_BF16 else 0


def _pipe_tag(d):
    return " [fp32 on split operands: 6 bf16-MFMA plane products]" if d.mma >= MMA_SPLIT else " [bf16 MFMA]" if d.mma == MMA_BF16 else ""


def out_size(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def conv_fwd(spec, x1, x2=None, up1=False, act=ACT_NONE, stats=False, tile=0, groups=1):
    """y = act(conv(cat(up2?(x1), x2)) + bias).  x1/x2 NHWC.  Returns y, or (y, stats_slab, mtiles) when stats."""
    B = x1.shape[0]
    Hs, Ws = (x1.shape[1] * 2, x1.shape[2] * 2) if up1 else (x1.shape[1], x1.shape[2])
    C1 = x1.shape[3]
    C2 = x2.shape[3] if x2 is not None else 0
    Hd, Wd = out_size(Hs, spec.kh, spec.stride, spec.pad), out_size(Ws, spec.kw, spec.stride, spec.pad)
    y = empty((B, Hd, Wd, spec.cout), x1)
    d = IgemmDesc()
    d.x1, d.x2 = P(x1), P(x2)
    d.B, d.Hs, d.Ws, d.C1, d.C2, d.up1 = B, Hs, Ws, C1, C2, int(up1)
    d.kh, d.kw, d.Np, d.Kp = spec.kh, spec.kw, spec.np, spec.kp
    d.mode = G_SMALLC if spec.smallc else G_DIRECT
    d.stride, d.sign, d.offset, d.pad_mode = spec.stride, 1, -spec.pad, spec.pad_mode
    d.y, d.Hd, d.Wd, d.Cd, d.n_begin, d.n_count, d.y_choff = P(y), Hd, Wd, spec.cout, 0, spec.cout, 0
    d.bias, d.act = P(spec.bias), act
    d.tile = tile
    d.groups = groups if stats else 1
    _weights_for(spec, d, False)
    if spec.is_stem() and not stats_blocks_stem(spec, stats) and x2 is None and not up1 and not (tile >> 9) & 1:
        d.w_stem = P(spec.packed_stem())               # a 7x7 stride-2 stem: patch-in-LDS kernel (tile bit 9 keeps the general one)
        if spec.mma == MMA_SPLIT_ALL and spec.smallc and spec.cout == 64:
            d.mma = spec.mma                           # the depth stem in the split form (stem7x7s2_split_fwd_kernel splits the packed fp32 slice itself): level with
                                                       # the fp32 stem kernel, so only "every launch on the split kernels" (the parity tests) takes it
    if (up1 and x2 is not None and not stats and spec.kh == 3 and spec.kw == 3 and spec.stride == 1 and spec.pad == 1
            and spec.pad_mode == PAD_REFLECT and C1 % 16 == 0 and not (tile >> 11) & 1 and not d.mma):
        d.w_upmerge = P(spec.packed_upmerge(C1))       # the upsampled part as 4 merged taps on the low-resolution x1 (tile bit 11: off)
    h = L.lib()
    slab = None
    if stats:
        mt = h.mcav_igemm_mtiles(ctypes.byref(d))
        if mt <= 0:
            raise L.MCAVError("mcav_igemm_mtiles: invalid descriptor (%d)" % mt)
        slab = empty((mt, 2, spec.cout), x1)
        d.stats = P(slab)
    flops = 2.0 * B * Hd * Wd * spec.cout * spec.cin * spec.kh * spec.kw
    executed = flops
    if d.w_upmerge or (up1 and x2 is None and spec.kh == 3 and spec.pad_mode == PAD_REFLECT and C1 in (16, 32) and spec.cout <= 32
                       and not (tile >> 10) & 1 and not (tile >> 9) & 1):
        executed = flops * ((4.0 / 9.0) * C1 + C2) / (C1 + C2)          # upsampled source: 4 merged taps instead of 9 (table kernel / halo kernel)
    if d.w_stem:
        executed = flops * (8.0 / 7.0) * (1.0 if spec.cout >= 32 else 2.0)  # the stem's K order pads 7 columns to 8; 16 outputs use half a 32-wide tile
    abytes = 4.0 * (x1.numel() + (x2.numel() if x2 is not None else 0) + y.numel() + spec.weight.numel())
    with _Timed("fwd", flops, "M=%d N=%d K=%dx%d s%d %dx%d%s" % (B * Hd * Wd, spec.cout, spec.cin, spec.kh * spec.kw, spec.stride, Hd, Wd, _pipe_tag(d)),
                executed, _planes(d), abytes):
        L.check(h.mcav_igemm(ctypes.byref(d), L.stream()), "mcav_igemm(fwd)")
    return (y, slab) if stats else y


BN_BWD_FUSE = _os.environ.get("MCAV_BN_BWD_FUSE", "1") != "0"      # BatchNorm-backward column sums in the epilogue of the dgrad that produces dy


def can_fuse_bn_stats(spec):
    """The data gradient of `spec` can carry the BatchNorm-backward statistics of the layer it feeds (mcav_igemm_desc.stats_x): a stride-1
    zero-padded convolution (the DIRECT gather keeps the groups of a stacked pass apart)."""
    return BN_BWD_FUSE and spec.stride == 1 and spec.pad_mode == PAD_ZERO


def conv_dgrad(spec, dy, in_shape, n_begin=0, n_count=None, out=None, dact_aux=None, dact=ACT_NONE, addend=None, pool=False, tile=0, bn_stats=None):
    """Gradient w.r.t. the (logical, concatenated) conv input, channels [n_begin, n_begin + n_count).

    dy: [B, Hd, Wd, Cout] gradient at the conv output (pre-activation).  in_shape = (Hs, Ws) of the logical input.
    Epilogue: 2x2 sum pooling (pool), * act'(dact_aux), + addend.  Returns [B, Hs(/2), Ws(/2), n_count].
    bn_stats = (x_raw, BNState): the result is the gradient arriving at a train-mode BatchNorm whose raw input was x_raw; the epilogue also
    leaves that BatchNorm's backward partial sums (sum g, sum g * xhat per tile and channel) in a slab -> returns (y, slab, tiles per group)."""
    B, Hd, Wd, Cout = dy.shape
    Hs, Ws = in_shape
    if n_count is None:
        n_count = spec.cin
    if (pool and n_begin == 0 and out is None and spec.kh == 3 and spec.kw == 3 and spec.stride == 1 and spec.pad == 1
            and spec.pad_mode == PAD_REFLECT and n_count % 16 == 0 and n_count >= UPMERGE_ADJ_MIN_N and Cout == up16(spec.cout)
            and Cout % 32 == 0 and Hs % 2 == 0 and Ws % 2 == 0 and not (tile >> 12) & 1):
        return _dgrad_upsample_merged(spec, dy, (Hs, Ws), n_count, dact_aux, dact, addend, tile)
    y = out if out is not None else empty((B, Hs // 2 if pool else Hs, Ws // 2 if pool else Ws, n_count), dy)
    d = IgemmDesc()
    d.x1, d.x2 = P(dy), None
    d.B, d.Hs, d.Ws, d.C1, d.C2, d.up1 = B, Hd, Wd, Cout, 0, 0       # Cout = channels physically in dy (may be zero-padded past spec.cout)
    d.kh, d.kw, d.Np, d.Kp = spec.kh, spec.kw, up16(spec.cin), up16(spec.cout)
    if spec.stride == 1:
        if spec.pad_mode == PAD_REFLECT:
            d.mode, d.stride, d.sign, d.offset = G_ADJ_REFLECT, 1, -1, 1
        else:
            d.mode, d.stride, d.sign, d.offset = G_DIRECT, 1, -1, spec.pad
    elif spec.stride == 2 and spec.pad_mode == PAD_ZERO:
        d.mode, d.stride, d.sign, d.offset = G_ADJ_STRIDE2, 2, -1, spec.pad
    else:
        raise L.MCAVError("conv_dgrad: unsupported stride/padding combination")
    d.pad_mode = PAD_ZERO
    d.y, d.Hd, d.Wd, d.Cd, d.n_begin, d.n_count, d.y_choff = P(y), Hs, Ws, y.shape[3], n_begin, n_count, 0
    d.bias, d.act = None, ACT_NONE
    d.dact_aux, d.dact, d.addend, d.pool = P(dact_aux), dact, P(addend), int(pool)
    d.tile = tile
    _weights_for(spec, d, True)
    slab = None
    if bn_stats is not None:
        x_raw, st = bn_stats
        if pool or n_begin != 0 or n_count != y.shape[3] or tuple(x_raw.shape) != tuple(y.shape) or st.mean is None:
            raise L.MCAVError("conv_dgrad: bn_stats needs an un-pooled, whole-width gradient of the BatchNorm input's shape")
        d.groups = st.groups
        d.stats_x, d.stats_mean, d.stats_invstd = P(x_raw), P(st.mean), P(st.invstd)
        d.stats = d.x1                                  # (placeholder: the tile count does not depend on it)
        mt = L.lib().mcav_igemm_mtiles(ctypes.byref(d))
        if mt <= 0 or mt % st.groups:
            raise L.MCAVError("mcav_igemm_mtiles: invalid descriptor (%d)" % mt)
        slab = empty((mt, 2, n_count), dy)
        d.stats = P(slab)
    with _Timed("dgrad", 2.0 * B * Hd * Wd * spec.cout * n_count * spec.kh * spec.kw,
                "M=%d N=%d K=%dx%d s%d mode%d pool%d %dx%d%s%s" % (B * Hs * Ws, n_count, Cout, spec.kh * spec.kw, spec.stride, d.mode, int(pool), Hs, Ws,
                                                                  " +bn-bwd stats" if slab is not None else "", _pipe_tag(d)), None, _planes(d),
                4.0 * (B * Hd * Wd * spec.cout + y.numel() * (1 + (dact_aux is not None) + (addend is not None) + (slab is not None))
                       + spec.cout * n_count * spec.kh * spec.kw)):
        L.check(L.lib().mcav_igemm(ctypes.byref(d), L.stream()), "mcav_igemm(dgrad)")
    return y if slab is None else (y, slab, slab.shape[0] // bn_stats[1].groups)


UPMERGE_ADJ_MIN_N = int(_os.environ.get("MCAV_UPMERGE_ADJ_MIN_N", "32"))       # narrower outputs: the halo-tile adjoint kernel is faster


def _dgrad_upsample_merged(spec, dy, in_shape, c1, dact_aux, dact, addend, tile):
    """d loss / d a for conv(reflect_pad(up2(a)) ...): a 4x4 stride-2 conv of dy on the edge-replicated low-resolution domain (16 taps per
    low-resolution pixel instead of 4 pixels x 9 taps), then the ring folded back onto the border (see mcav_conv.h)."""
    B, Hd, Wd, Cout = dy.shape
    Hl, Wl = in_shape[0] // 2, in_shape[1] // 2
    tmp = empty((B, Hl + 2, Wl + 2, c1), dy)
    d = IgemmDesc()
    d.x1, d.x2 = P(dy), None
    d.B, d.Hs, d.Ws, d.C1, d.C2, d.up1 = B, Hd, Wd, Cout, 0, 0
    d.kh, d.kw, d.Np, d.Kp = 4, 4, up16(c1), up16(spec.cout)
    d.mode, d.stride, d.sign, d.offset, d.pad_mode = G_DIRECT, 2, 1, -3, PAD_ZERO
    d.y, d.Hd, d.Wd, d.Cd, d.n_begin, d.n_count, d.y_choff = P(tmp), Hl + 2, Wl + 2, c1, 0, c1, 0
    d.bias, d.act = None, ACT_NONE
    d.tile = tile & 0xff
    d.w = P(spec.packed_upmerge_adj(c1))
    if spec.mma in (MMA_BF16, MMA_SPLIT, MMA_SPLIT_ALL):
        d.mma, d.w16 = spec.mma, d.w
        if L.lib().mcav_igemm_uses_bf16(ctypes.byref(d)):
            d.w16 = P(spec.packed_upmerge_adj(c1, bf16=3 if spec.mma >= MMA_SPLIT else True))
        else:
            d.mma, d.w16 = 0, None
    with _Timed("dgrad", 2.0 * B * Hd * Wd * spec.cout * c1 * 9,
                "M=%d N=%d K=%dx9 s1 mode2 pool1 (merged 4x4/s2) %dx%d" % (B * Hd * Wd, c1, Cout, Hd, Wd),
                2.0 * B * (Hl + 2) * (Wl + 2) * spec.cout * c1 * 16, _planes(d),          # 16 taps per low-resolution pixel (incl. the ring) instead of 4 x 9
                4.0 * (dy.numel() + B * Hl * Wl * c1 * (1 + (dact_aux is not None) + (addend is not None)) + spec.cout * c1 * 9)):
        L.check(L.lib().mcav_igemm(ctypes.byref(d), L.stream()), "mcav_igemm(dgrad, merged upsample)")
        y = empty((B, Hl, Wl, c1), dy)
        L.check(L.lib().mcav_upsample_adj_fold(P(tmp), B, Hl, Wl, c1, P(dact_aux), dact, P(addend), P(y), L.stream()), "mcav_upsample_adj_fold")
    return y


def conv_wgrad(spec, x1, dy, x2=None, up1=False, tile=0):
    """Accumulates d loss / d weight (OIHW) and d loss / d bias into the parameters' .grad buffers."""
    B = x1.shape[0]
    Hs, Ws = (x1.shape[1] * 2, x1.shape[2] * 2) if up1 else (x1.shape[1], x1.shape[2])
    gw = grad_buffer(spec.weight)
    gb = grad_buffer(spec.bias) if spec.bias is not None else None
    d = WgradDesc()
    d.x1, d.x2 = P(x1), P(x2)
    d.B, d.Hs, d.Ws, d.C1, d.C2, d.up1 = B, Hs, Ws, x1.shape[3], (x2.shape[3] if x2 is not None else 0), int(up1)
    d.kh, d.kw, d.Kp = spec.kh, spec.kw, spec.kp
    d.mode = G_SMALLC if spec.smallc else G_DIRECT
    d.stride, d.sign, d.offset, d.pad_mode = spec.stride, 1, -spec.pad, spec.pad_mode
    d.dy, d.Hd, d.Wd, d.Cdy, d.dy_choff = P(dy), dy.shape[1], dy.shape[2], dy.shape[3], 0
    d.Cout, d.Cin = spec.cout, spec.cin
    d.dw_oihw, d.accumulate, d.dbias = P(gw), 1, P(gb)
    d.tile = tile
    # MMA_SPLIT: the library takes the split form where it is ahead (wgrad3x3_patch_kernel: single-source 3x3 stride-1 layers of 64-channel
    # multiples) and the fp32 MFMA kernels elsewhere; MCAV_SPLIT_WGRAD=1 = the split form on every launch the bf16 kernels cover
    d.mma = MMA_SPLIT_ALL if (spec.mma == MMA_SPLIT and SPLIT_WGRAD) else (spec.mma if spec.mma in (MMA_BF16, MMA_SPLIT, MMA_SPLIT_ALL) else 0)
    bf16 = bool(d.mma and L.lib().mcav_wgrad_uses_bf16(ctypes.byref(d)))
    flops = 2.0 * B * dy.shape[1] * dy.shape[2] * spec.cout * spec.cin * spec.kh * spec.kw
    tag = "pix=%d Cout=%d Ktot=%dx%d s%d" % (B * dy.shape[1] * dy.shape[2], spec.cout, spec.cin, spec.kh * spec.kw, spec.stride)
    c1 = x1.shape[3]
    if bf16 and x2 is not None and c1 % 16 == 0 and x2.shape[3] % 16 == 0:
        # bf16: the two sources as two single-source launches (a two-source row tile would fetch both tensors for every tap: measured 4x slower);
        # each writes its own input-channel range of the OIHW gradient.  The merged-tap form is not used: the bf16 kernels are memory-bound.
        c2 = x2.shape[3]
        d.x1, d.x2, d.C1, d.C2, d.up1, d.Kp = P(x2), None, c2, 0, 0, c2
        d.Cin, d.Cin_total, d.ci_offset, d.upm = c2, spec.cin, c1, 0
        launch_wgrad(d, (x2, dy), flops * c2 / spec.cin, tag + " [second source, bf16]")
        d.x1, d.C1, d.up1, d.Kp = P(x1), c1, int(up1), c1
        d.Cin, d.ci_offset, d.dbias = c1, 0, None
        launch_wgrad(d, (x1, dy), flops * c1 / spec.cin, tag + " [first source, bf16]")
        return
    if (not bf16 and up1 and x2 is not None and spec.kh == 3 and spec.kw == 3 and spec.stride == 1 and spec.pad == 1 and spec.pad_mode == PAD_REFLECT
            and c1 % 16 == 0 and x2.shape[3] % 16 == 0 and Hs % 2 == 0 and Ws % 2 == 0 and not (tile >> 11) & 1
            and (spec.cout >= 64 or (tile >> 12) & 1          # measured: 32 output channels are faster in one launch (bit 12 forces) ...
                 or (d.mma >= MMA_SPLIT and spec.cout == 32 and x2.shape[3] % 64 == 0))):      # ... unless the skip half takes the patch kernel
        # two launches: the skip tensor's channels as an ordinary weight gradient, the upsampled map's in merged-tap form (mcav_conv.h)
        c2 = x2.shape[3]
        d.x1, d.x2, d.C1, d.C2, d.up1, d.Kp = P(x2), None, c2, 0, 0, up16(c2)
        d.Cin, d.Cin_total, d.ci_offset, d.upm = c2, spec.cin, c1, 0
        launch_wgrad(d, (x2, dy), flops * c2 / spec.cin, tag + " [skip half]")
        d.x1, d.C1, d.up1, d.Kp = P(x1), c1, 1, c1
        d.Cin, d.ci_offset, d.upm, d.dbias, d.tile = c1, 0, 1, None, 0
        launch_wgrad(d, (x1, dy), flops * c1 / spec.cin, tag + " [upsampled half, merged taps]", flops * c1 / spec.cin * 4.0 / 9.0)
        return
    executed = flops
    if spec.is_stem() and x2 is None and not up1 and not (tile >> 9) & 1:
        executed = flops * (8.0 / 7.0) * (1.0 if spec.cout >= 32 else 2.0)
    elif up1 and x2 is None and spec.kh == 3 and spec.pad_mode == PAD_REFLECT and c1 == 16 and spec.cout <= 16 and not (tile >> 9) & 1:
        executed = flops * 4.0 / 9.0                                      # level 0: conv3x3_halo_wgrad_up_kernel (merged taps)
    launch_wgrad(d, (x1, x2, dy), flops, tag, executed)


class _WgradSide:
    """Weight-gradient launches go to a second HIP stream.

    Inside a backward pass the wgrad of a layer only feeds the optimiser, while its dgrad feeds the next layer: the wgrad chain
    (GEMM + slab reductions, about a third of the step) runs concurrently with the dgrad chain of the main stream, so that
    the ramp-up, tail and launch gap of one kernel are filled with the other chain's workgroups.  The fork is an event wait
    per launch, the join is queued on the autograd engine as an end-of-backward callback (the mechanism DDP uses), so
    `.grad` is complete on the caller's stream when `backward()` returns.  Outside a backward pass (direct calls, tests,
    micro-benchmarks) launches stay on the caller's stream.

    The same callback also joins every stream a network's backward ran on (autograd replays a backward node on the stream of
    its forward, e.g. the pose branch of mcav/streams.py) with the stream `backward()` was called from: the engine only does
    that for gradients it accumulates itself, and these networks write their parameter gradients straight into the arena."""

    def __init__(self):
        self.enabled = _os.environ.get("MCAV_WGRAD_SIDE", "1") != "0"      # (0: weight gradients in line on the calling stream -- A/B timing)
        self.stream = None
        self.keep = []
        self.mains = []            # streams a network backward ran on during this backward pass
        self.in_backward = False
        self.forked = False
        self.dual = None           # set by mcav/graph.py while TWO graphs are captured at once (main chain / weight-gradient chain): an object
                                   # with .stream (the side stream, capturing its own graph), .fork(main) and .join(main) -- external
                                   # event record / wait node pairs instead of the eager path's cross-stream event waits, which would
                                   # merge the side stream into the main stream's capture

    def _note(self, main):
        if not self.in_backward:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.join)
            except RuntimeError:             # not inside a backward pass
                return False
            self.in_backward = True
        if all(m != main for m in self.mains):
            self.mains.append(main)
        return True

    def run(self, fn, tensors):
        if not tensors[0].is_cuda:
            return fn()
        from . import streams
        main = torch.cuda.current_stream()
        if self.dual is not None and main == self.dual.stream:
            return fn()                          # already on the side chain (the pose network's backward under a two-graph capture): in order as it is
        if self.dual is not None and self.enabled and self._note(main):
            self.stream = self.dual.stream
            self.dual.fork(main)                 # side graph: wait for what the main graph has recorded up to here
            with torch.cuda.stream(self.stream):
                fn()
            self.keep.append(tensors)
            self.forked = True
            return
        if not self._note(main) or not self.enabled or PROFILE is not None or streams.SERIAL:
            return fn()
        if self.stream is None or self.stream.device != main.device:
            self.stream = torch.cuda.Stream(device=main.device)
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            fn()
        self.keep.append(tensors)            # the forking stream's allocator must not recycle these before the join
        self.forked = True

    def join(self):
        cur = torch.cuda.current_stream()
        flush_wgrad_batch()                  # the last gradient bucket's slabs: reduced behind their GEMMs, on the stream those ran on
        if self.forked and self.dual is not None:
            self.dual.join(cur)                  # main graph: wait for the side graph's last node (one stream under capture: mains == [cur])
        elif self.forked:
            for m in self.mains:
                m.wait_stream(self.stream)
            cur.wait_stream(self.stream)
        for m in self.mains:
            if m != cur:
                cur.wait_stream(m)
        self.keep.clear()
        self.mains = []
        self.in_backward = False
        self.forked = False


WGRAD_SIDE = _WgradSide()

# Data parallelism (mcav/dist.py:GradSync): called from a network's backward when every gradient of a group of parameters has
# been ISSUED (weight gradients on the wgrad stream, BatchNorm gradients on the calling stream), so that their slice of the
# gradient arena can be all-reduced while the rest of the backward pass still runs.  None on a single GPU.
GRADS_READY = None


def grads_ready(params):
    """A network's backward calls this when every gradient of a group of parameters has been issued (a bucket boundary): the pending slab
    reductions of the bucket go out as one batch, then the data-parallel hook (if any) may all-reduce the bucket."""
    flush_wgrad_batch()
    if GRADS_READY is not None:
        GRADS_READY(list(params))


def flush_wgrad_batch():
    """Issue the pending slab reductions on the stream their GEMMs were issued on (the weight-gradient side stream once it has forked)."""
    if WGRAD_BATCH.items:
        if WGRAD_SIDE.forked:
            with torch.cuda.stream(WGRAD_SIDE.stream):
                WGRAD_BATCH.flush()
        else:
            WGRAD_BATCH.flush()


class _WgradBatch:
    """Weight-gradient slab reductions of a gradient bucket as ONE presum + ONE reduce launch (mcav_wgrad_deferred / mcav_wgrad_reduce_multi).
    Inside a backward pass every weight-gradient GEMM leaves its slab in its own buffer (the k-th pending launch uses pool[k]: the launch
    sequence of a step is static, so the buffers and therefore the device tables are the same every step -- a table is built once per
    launch sequence).  Outside a backward pass (direct calls, micro-benchmarks) the per-layer launches run."""

    def __init__(self):
        self.enabled = os.environ.get("MCAV_WGRAD_BATCH", "1") != "0"
        self.items = []           # WgradReduceItem of the pending launches
        self.targets = set()      # (gradient pointer, first input channel) of the pending launches
        self.pool = {}            # (device, k) -> byte buffer of the k-th pending launch
        self.tables = {}          # bytes of the planned item array -> (device table, presum blocks, reduce blocks, lds bytes)
        self.retired = []         # outgrown buffers: NEVER freed -- a captured hipGraph may have their addresses baked in (see buffer())
        self.device = None

    def buffer(self, nbytes, device):
        """The k-th pending launch's slab buffer.  A hipGraph captured for a smaller input shape has this buffer's address (and the device
        table that names it) baked into its nodes, so a buffer that a larger shape outgrows is retired, not freed, and the tables built for
        it stay cached: the earlier graph keeps replaying into memory nothing else owns (tests/graph_fresh_worker.py runs
        small-then-large StepGraphs in a fresh process).  The table cache is keyed by the item array's bytes, slab addresses included, so a
        stale table can never be hit by a launch sequence that uses the new buffers.  Growth is monotone per k: at most a handful retire."""
        k = (str(device), len(self.items))
        buf = self.pool.get(k)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                self.retired.append(buf)
            buf = self.pool[k] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.device = device
        return buf

    def flush(self):
        n = len(self.items)
        if n == 0:
            return
        arr = (WgradReduceItem * n)(*self.items)
        self.items = []
        self.targets = set()
        pb, rb, lds = c_i(0), c_i(0), c_sz(0)
        h = L.lib()
        L.check(h.mcav_wgrad_reduce_plan(arr, n, ctypes.byref(pb), ctypes.byref(rb), ctypes.byref(lds)), "mcav_wgrad_reduce_plan")
        key = (str(self.device), bytes(arr))
        hit = self.tables.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                raise L.MCAVError("a batch of weight-gradient slab reductions was cut differently under hipGraph capture than in the warm-up step "
                                  "(its device table does not exist yet, and building it is a host -> device copy): warm-up and capture must "
                                  "announce the same gradient buckets (mcav/graph.py)")
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            hit = self.tables[key] = (table, pb.value, rb.value, lds.value)
        table, npb, nrb, nlds = hit
        with _Timed("wgrad", 0.0, "slab reduction of %d weight gradients (batched)" % n):
            L.check(h.mcav_wgrad_reduce_multi(P(table), n, npb, nrb, nlds, L.stream()), "mcav_wgrad_reduce_multi")


WGRAD_BATCH = _WgradBatch()


LAST_WGRAD_PLANES = 0


def launch_wgrad(d, tensors, flops=0.0, tag="", executed=None):
    """mcav_wgrad for a filled descriptor (workspace handling + stream choice).  tensors: what the launch reads."""
    h = L.lib()
    abytes = 4.0 * (sum(t.numel() for t in tensors if t is not None) + d.Cout * d.Cin * d.kh * d.kw)
    nbytes = h.mcav_wgrad_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise L.MCAVError("mcav_wgrad: invalid descriptor")

    dev = tensors[0].device
    planes = _planes(d) if (d.mma and h.mcav_wgrad_uses_bf16(ctypes.byref(d))) else 0
    global LAST_WGRAD_PLANES
    LAST_WGRAD_PLANES = planes                      # (tests: which pipe the last weight-gradient launch took -- 6: split form, 1: bf16, 0: fp32 MFMA)
    if planes:
        tag += _pipe_tag(d)
    defer = WGRAD_BATCH.enabled and dev.type == "cuda" and WGRAD_SIDE._note(torch.cuda.current_stream())
    if defer:
        # inside a backward pass: GEMM now, the slab reduction with its bucket (grads_ready / the end-of-backward join).  The descriptor is
        # copied: the caller re-uses `d` for its next launch.
        target = (d.dw_oihw, d.ci_offset)
        if target in WGRAD_BATCH.targets:        # a second launch into the same gradient (separate depth passes): its accumulate must see the first
            flush_wgrad_batch()
        WGRAD_BATCH.targets.add(target)
        ws = WGRAD_BATCH.buffer(nbytes, dev)
        item = WgradReduceItem()
        dd = WgradDesc.from_buffer_copy(d)

        def go_deferred():
            with _Timed("wgrad", flops, tag, executed, planes, abytes):
                L.check(h.mcav_wgrad_deferred(ctypes.byref(dd), P(ws), ws.numel(), ctypes.byref(item), L.stream()), "mcav_wgrad_deferred")
        WGRAD_SIDE.run(go_deferred, tensors)
        WGRAD_BATCH.items.append(item)
        return

    def go():
        ws = L.workspace(nbytes, dev, "wgrad")
        with _Timed("wgrad", flops, tag, executed, planes, abytes):
            L.check(h.mcav_wgrad(ctypes.byref(d), P(ws), ws.numel(), L.stream()), "mcav_wgrad")
    WGRAD_SIDE.run(go, tensors)


def grad_buffer(param):
    """The tensor gradients are accumulated into: param.grad (created zeroed on first use; arena-backed when flattened)."""
    if param.grad is None:
        make = getattr(param, "_mcav_grad_view", None)
        param.grad = make() if make is not None else torch.zeros_like(param)
    return param.grad


# ------------------------------------------------------------------------------------------------ BatchNorm
class BNState:
    """Per-call saved tensors of one train-mode BatchNorm."""
    __slots__ = ("scale", "shift", "mean", "invstd", "groups")


def bn_train_coeffs(bn, slab, count, groups=1):
    """bn: holder with weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum.
    count: pixels per group.  groups > 1: per-group statistics, running stats updated once per group in order."""
    C = bn.weight.shape[0]
    st = BNState()
    buf = empty((4, groups, C), bn.weight)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    st.groups = groups
    h = L.lib()
    mtiles = slab.shape[0] // groups
    nbytes = h.mcav_bn_finalize_workspace_bytes(mtiles, C, groups)
    ws = L.workspace(nbytes, slab.device, "bn_fin", zero=True) if nbytes else None      # (holds the finalize kernel's completion tickets: zero at allocation)
    L.check(h.mcav_bn_finalize(P(slab), mtiles, C, float(count), P(bn.weight), P(bn.bias), bn.eps, bn.momentum,
                               P(bn.running_mean), P(bn.running_var), P(st.scale), P(st.shift), P(st.mean), P(st.invstd), groups,
                               P(ws), ws.numel() if ws is not None else 0, L.stream()), "mcav_bn_finalize")
    _NBT_PENDING.append((bn.num_batches_tracked, groups))      # one fused add per forward pass instead of one tiny launch per layer
    if len(_NBT_PENDING) >= 256:
        flush_bn_counters()
    return st


_NBT_PENDING = []


def flush_bn_counters():
    """num_batches_tracked += groups for every train-mode BatchNorm since the last flush (the networks call this at the end of forward)."""
    if _NBT_PENDING:
        by_inc = {}
        for t, inc in _NBT_PENDING:
            by_inc.setdefault(inc, []).append(t)
        for inc, ts in by_inc.items():
            torch._foreach_add_(ts, inc)
        _NBT_PENDING.clear()


def bn_eval_coeffs(bn):
    C = bn.weight.shape[0]
    st = BNState()
    buf = empty((2, C), bn.weight)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], None, None
    st.groups = 1
    L.check(L.lib().mcav_bn_eval_coeffs(P(bn.weight), P(bn.bias), P(bn.running_mean), P(bn.running_var), bn.eps, C, P(st.scale), P(st.shift),
                                        L.stream()), "mcav_bn_eval_coeffs")
    return st


def bn_apply(x, st, relu, residual=None):
    y = torch.empty_like(x)
    C = x.shape[-1]
    n_pix = x.numel() // C
    L.check(L.lib().mcav_bn_apply(P(x), P(st.scale), P(st.shift), P(residual), ACT_RELU if relu else ACT_NONE, n_pix, C, P(y),
                                  n_pix // st.groups, L.stream()), "mcav_bn_apply")
    return y


def bn_backward(bn, st, dy, y_act, x, relu, want_dres=False, dres_out=None, dres_accumulate=False, fused=None):
    """-> dx (and dz, the masked incoming gradient, when want_dres).  Accumulates dgamma/dbeta into .grad.
    fused = (slab, tiles per group): dy comes from a data gradient that already applied the ReLU mask and left this BatchNorm's partial sums
    in `slab` (conv_dgrad(bn_stats=...)): the reduce pass over dy / the activations / x is skipped, and so is the mask in the apply pass."""
    C = x.shape[-1]
    n_pix = x.numel() // C
    h = L.lib()
    G = st.groups
    if fused is not None:
        slab, mtg = fused
        sums = empty((G, 2, C), x)
        gg, gb = grad_buffer(bn.weight), grad_buffer(bn.bias)
        L.check(h.mcav_bn_bwd_finalize(P(slab), mtg, C, P(gg), P(gb), 1, P(sums), G, L.stream()), "mcav_bn_bwd_finalize")
        dx = torch.empty_like(x)
        L.check(h.mcav_bn_bwd_apply(P(dy), None, P(x), P(bn.weight), P(st.mean), P(st.invstd), P(sums), 0, n_pix, C, P(dx), None, 0, G, L.stream()),
                "mcav_bn_bwd_apply")
        if want_dres:
            if dres_out is not None:
                raise L.MCAVError("bn_backward: a fused gradient is its own residual gradient (no dres_out)")
            return dx, dy                       # the masked incoming gradient IS dy
        return dx
    ws = L.workspace(h.mcav_bn_bwd_workspace_bytes(n_pix, C, G), x.device, "bn_bwd")
    sums = empty((G, 2, C), x)
    gg, gb = grad_buffer(bn.weight), grad_buffer(bn.bias)
    L.check(h.mcav_bn_bwd_reduce(P(dy), P(y_act), P(x), P(st.mean), P(st.invstd), int(relu), n_pix, C, P(gg), P(gb), 1, P(sums), G, P(ws),
                                 ws.numel(), L.stream()), "mcav_bn_bwd_reduce")
    dx = torch.empty_like(x)
    dres = None
    if want_dres:
        dres = dres_out if dres_out is not None else torch.empty_like(x)
    L.check(h.mcav_bn_bwd_apply(P(dy), P(y_act), P(x), P(bn.weight), P(st.mean), P(st.invstd), P(sums), int(relu), n_pix, C, P(dx), P(dres),
                                int(dres_accumulate), G, L.stream()), "mcav_bn_bwd_apply")
    return (dx, dres) if want_dres else dx


# ------------------------------------------------------------------------------------------------ misc ops
def nchw_to_nhwc(src, Cp, dst=None, choff=0):
    B, C, H, W = src.shape
    if dst is None:
        dst = torch.zeros((B, H, W, Cp), dtype=torch.float32, device=src.device)
    L.check(L.lib().mcav_nchw_to_nhwc(P(src), B, C, H, W, P(dst), Cp, choff, L.stream()), "mcav_nchw_to_nhwc")
    return dst


def nchw3_to_nhwc(s0, s1, s2, Cp):
    """cat([s0, s1, s2], 1) -> NHWC with Cp channels (zeros past 3 C), one launch."""
    B, C, H, W = s0.shape
    if s1.shape != s0.shape or s2.shape != s0.shape:
        raise L.MCAVError("nchw3_to_nhwc: the three images must have one shape")
    dst = torch.empty((B, H, W, Cp), dtype=torch.float32, device=s0.device)
    L.check(L.lib().mcav_nchw3_to_nhwc(P(s0), P(s1), P(s2), B, C, H, W, P(dst), Cp, L.stream()), "mcav_nchw3_to_nhwc")
    return dst


def nhwc_to_nchw(src, C=None, choff=0):
    B, H, W, Cp = src.shape
    C = Cp if C is None else C
    dst = empty((B, C, H, W), src)
    L.check(L.lib().mcav_nhwc_to_nchw(P(src), B, C, H, W, Cp, choff, P(dst), L.stream()), "mcav_nhwc_to_nchw")
    return dst


def maxpool_fwd(x):
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = empty((B, Ho, Wo, C), x)
    idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    L.check(L.lib().mcav_maxpool3s2_fwd(P(x), B, H, W, C, P(y), P(idx), L.stream()), "mcav_maxpool3s2_fwd")
    return y, idx


def maxpool_bwd(dy, idx, in_shape, dx=None, accumulate=False):
    B, H, W, C = in_shape
    if dx is None:
        dx = empty(in_shape, dy)
        accumulate = False
    L.check(L.lib().mcav_maxpool3s2_bwd(P(dy), P(idx), B, H, W, C, P(dx), int(accumulate), L.stream()), "mcav_maxpool3s2_bwd")
    return dx


def act_bwd(dy, y, act, out=None, accumulate=False):
    if out is None:
        out = torch.empty_like(y)
        accumulate = False
    L.check(L.lib().mcav_act_bwd(P(dy), P(y), act, y.numel(), P(out), int(accumulate), L.stream()), "mcav_act_bwd")
    return out


def act_bwd_padded(dy, y, act, cp):
    """dy * act'(y) for a 1-channel map, written to channel 0 of a zeroed [B,H,W,cp] tensor."""
    B, H, W, C = y.shape
    assert C == 1
    out = torch.zeros((B, H, W, cp), dtype=torch.float32, device=y.device)
    L.check(L.lib().mcav_act_bwd_strided(P(dy), P(y), act, y.numel(), P(out), cp, L.stream()), "mcav_act_bwd_strided")
    return out


# ------------------------------------------------------------------------------------------------ one-channel 3x3 heads
def narrow_ok(spec, x):
    """The disparity-head shape the stencil kernels cover: 3x3, stride 1, reflection pad 1, one output channel."""
    return (spec.cout == 1 and spec.kh == 3 and spec.kw == 3 and spec.stride == 1 and spec.pad == 1 and spec.pad_mode == PAD_REFLECT
            and x.shape[3] == spec.cin and spec.cin in (16, 32, 64, 128) and x.shape[1] >= 2 and x.shape[2] >= 2)


def conv3x3r_c1_fwd(spec, x, act):
    B, H, W, C = x.shape
    y = empty((B, H, W, 1), x)
    with _Timed("fwd", 2.0 * B * H * W * C * 9, "head M=%d N=1 K=%dx9 %dx%d" % (B * H * W, C, H, W), None, 0, 4.0 * (x.numel() + y.numel())):
        L.check(L.lib().mcav_conv3x3r_c1_fwd(P(x), B, H, W, C, P(spec.weight), P(spec.bias), act, P(y), L.stream()), "mcav_conv3x3r_c1_fwd")
    return y


def conv3x3r_c1_bwd(spec, x, dy, y, act, x_act, addend=None):
    """Fused backward of the head: -> dx = adjoint(dy * act'(y)) * x_act'(x) + addend; accumulates weight / bias gradients."""
    B, H, W, C = x.shape
    h = L.lib()
    if act != ACT_NONE and B * H * W >= (1 << 16):      # large maps: one elementwise pass for dy * act'(y), then half the gathers
        dy, y, act = act_bwd(dy, y, act), None, ACT_NONE
    ws = L.workspace(h.mcav_conv3x3r_c1_bwd_workspace_bytes(C), x.device, "narrow")
    dx = torch.empty_like(x)
    gw = grad_buffer(spec.weight)
    gb = grad_buffer(spec.bias) if spec.bias is not None else None
    with _Timed("dgrad", 4.0 * B * H * W * C * 9, "head bwd (dgrad+wgrad) M=%d K=%dx9 %dx%d" % (B * H * W, C, H, W), None, 0,
                4.0 * (2 * x.numel() + dy.numel() + (addend.numel() if addend is not None else 0))):
        L.check(h.mcav_conv3x3r_c1_bwd(P(x), B, H, W, C, P(spec.weight), P(dy), P(y), act, x_act, P(addend), P(dx), P(gw), P(gb), 1,
                                       P(ws), ws.numel(), L.stream()), "mcav_conv3x3r_c1_bwd")
    return dx


def add(a, b):
    out = torch.empty_like(a)
    L.check(L.lib().mcav_add(P(a), P(b), a.numel(), P(out), L.stream()), "mcav_add")
    return out


def spatial_mean(x, scale):
    B, H, W, C = x.shape
    out = empty((B, C), x)
    L.check(L.lib().mcav_spatial_mean(P(x), B, H * W, C, scale, P(out), L.stream()), "mcav_spatial_mean")
    return out


def spatial_mean_bwd(dout, shape, scale):
    B, H, W, C = shape
    dx = empty(shape, dout)
    L.check(L.lib().mcav_spatial_mean_bwd(P(dout), B, H * W, C, scale, P(dx), L.stream()), "mcav_spatial_mean_bwd")
    return dx
