"""Pack a Terran `state_dict` into the op program + weight blob `libterran_amd.so` runs.

Replaces the reference's `load_model()` + `nn.Module` construction
(retinaface/wrapper.py:16-22, arcface/wrapper.py:13-19, openpose/wrapper.py:27-36):
BatchNorms are folded (float64 math, float32 result), weights are laid out for the
implicit-GEMM kernel ([K-slab][cout][32 floats], K = (ky,kx,cin) with cin fastest), and
the graph is flattened into conv / depthwise / pool / copy ops over halo-padded NHWC
tensors.  Concats become channel slices of a shared tensor; the few graph-level fusions
(merged sibling convs, residual/upsample-add and the next unit's BatchNorm in the conv
epilogue) are documented next to each builder.

Blob layout mirrors `terran_amd/csrc/ta_internal.h` (ta_blob_header / ta_tensor_desc /
ta_op_desc).
"""
import os

import numpy as np

from . import arch

MODEL_RETINAFACE, MODEL_ARCFACE, MODEL_OPENPOSE = 1, 2, 3
OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_COPYCH, OP_RFSTEM, OP_DWPW = 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_RELU, ACT_PRELU = 0, 1, 2
MAGIC = 0x314D4154

HEADER_DT = np.dtype({
    'names': ['magic', 'version', 'kind', 'n_tensors', 'n_ops', 'input_tensor', 'n_outputs', 'outputs',
              'tensors_off', 'ops_off', 'weights_off', 'weights_bytes'],
    'formats': ['<u4', '<u4', '<i4', '<i4', '<i4', '<i4', '<i4', ('<i4', 16), '<i8', '<i8', '<i8', '<i8'],
    'offsets': [0, 4, 8, 12, 16, 20, 24, 28, 96, 104, 112, 120],
    'itemsize': 128,
})
TENSOR_DT = np.dtype([('channels', '<i4'), ('halo', '<i4'), ('alias_of', '<i4'), ('fmt', '<i4'), ('unscale_off', '<i4')])
FMT_F32, FMT_SPLIT, FMT_SPLIT16, FMT_F16 = 0, 1, 2, 3
_OP_I32 = ['type', 'in', 'out', 'in_ch_off', 'cin', 'out_ch_off', 'cout', 'coutp', 'kh', 'kw', 'stride', 'pad',
           'act', 'res', 'res_ch_off', 'res_up2', 'out2', 'out2_ch_off', 'n_slabs', 'prec', 'groups', 'variant', 'pool', 'wscale_log2']
_OP_I64 = ['w_off', 'bias_off', 'prelu_off', 'scale2_off', 'shift2_off', 'wus_off']
OP_DT = np.dtype([(n, '<i4') for n in _OP_I32] + [(n, '<i8') for n in _OP_I64] + [('macs_per_pixel', '<f8')])
assert OP_DT.itemsize == 152 and TENSOR_DT.itemsize == 20
BLOB_VERSION = 9            # 9: OP_RFSTEM with cout == 32 = the front kernel fused with the next depthwise + 1x1 block; 8: per-channel activation scales (ta_tensor_desc.unscale_off) and per-output-channel un-scale vectors (ta_op_desc.wus_off) replace the per-layer wscale_log2; 7: arithmetic mode 4 ('f16') and the 2-byte tensor format; 6: op lanes (variant bits 17..18); 2: ta_op_desc grew `groups` (grouped convs); 3: fused RetinaFace ops (OP_RFSTEM, OP_DWPW), `variant`; 4: `pool`; 5: `wscale_log2` (f16x3)


# every environment switch a packer reads: a program packed with one of them set must never be served to a default run
# (runtime.packed_program bypasses the on-disk repack cache then, and keys its in-process memo on them)
PACK_SWITCHES = ('TERRAN_AMD_NO_FUSED_POOL', 'TERRAN_AMD_NO_GROUPED', 'TERRAN_AMD_NO_MERGED_OUTPUTS', 'TERRAN_AMD_ARCFACE_SECOND_OUTPUT',
                 'TERRAN_AMD_NO_STAGE4_KSPLIT', 'TERRAN_AMD_DETECTOR_F32', 'TERRAN_AMD_DETECTOR_BASE_F32', 'TERRAN_AMD_NO_FUSED_DETECTOR', 'TERRAN_AMD_NO_FUSED_FRONT',
                 'TERRAN_AMD_NO_DETECTOR_LANES', 'TERRAN_AMD_NO_ACT_SCALES')


def active_switches():
    return tuple((k, os.environ[k]) for k in PACK_SWITCHES if os.environ.get(k))


_SOURCE_TAG = []


def source_tag():
    """Short hash of the packer's own sources: a repack cache written by another version of pack.py / arch.py is not read.
    Computed once per process; an install without the .py files (pyc-only, zip) has nothing to hash: it returns None and
    runtime.packed_program then runs WITHOUT the on-disk cache (two such installs of different packers would share one key)."""
    if not _SOURCE_TAG:
        import hashlib
        h = hashlib.sha256()
        here = os.path.dirname(os.path.abspath(__file__))
        try:
            for f in ('pack.py', 'arch.py'):
                with open(os.path.join(here, f), 'rb') as fh:
                    h.update(fh.read())
            _SOURCE_TAG.append(h.hexdigest()[:10])
        except OSError:
            _SOURCE_TAG.append(None)
    return _SOURCE_TAG[0]


PRECISIONS = {'f32': 0, 'bf16x3': 1, 'bf16': 2, 'f16x3': 3, 'f16': 4, 'f16x2': 5}     # 'f16x2': f16x3's tensors and weights, two of its three MFMAs per product ((w_hi + w_lo) * x_hi): embedder only
SPLIT_FMT = {0: FMT_F32, 1: FMT_SPLIT, 2: FMT_SPLIT, 3: FMT_SPLIT16, 4: FMT_F16, 5: FMT_SPLIT16}     # pre-split activation format per arithmetic mode


def _rup(x, m):
    return (x + m - 1) // m * m


def _bf16_bits(x):
    """float32 -> bfloat16 bit pattern (uint16), round to nearest even."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def _bf16_to_f32(b):
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def split_bf16_rows(packed):
    """[slab][cout][32] float32 -> the same 128-byte rows as [hi x32 | lo x32] bfloat16, returned as a
    float32-typed view of the bytes (x = hi + lo to ~2^-17 relative)."""
    hi = _bf16_bits(packed)
    lo = _bf16_bits(packed - _bf16_to_f32(hi))
    rows = np.concatenate([hi, lo], axis=-1)                # (..., 64) uint16
    return np.ascontiguousarray(rows).view(np.float32)      # (..., 32)


def row_exponents(packed):
    """[slab][cout][32] float32 -> per-output-channel exponents s[cout]: max |W[co, :]| 2^s[co] lies in [2^13, 2^14)
    (0 for an all-zero row)."""
    m = np.abs(np.asarray(packed, np.float32)).max(axis=(0, 2)).astype(np.float64)
    ok = np.isfinite(m) & (m > 0)
    s = np.zeros(m.shape, np.int64)
    s[ok] = np.clip(13 - np.floor(np.log2(m[ok])), -60, 60).astype(np.int64)
    return s


def split_f16_rows(packed, exps=None):
    """[slab][cout][32] float32 -> ([hi x32 | lo x32] IEEE half rows viewed as float32, exponents s[cout]).

    Half floats carry 11 significant bits down to 2^-14 only; a lo half below that loses bits.  Every OUTPUT CHANNEL's
    row is therefore packed times 2^s[co], s[co] chosen so that max |W[co, :]| 2^s[co] lies in [2^13, 2^14): every weight
    within 2^-16 of its row's largest keeps a normal lo half (22 significant bits in hi + lo), nothing gets near 65504,
    and a channel whose weights are all small (a BatchNorm with a small gamma / sigma folded in) keeps its bits instead of
    inheriting the scale of the layer's largest channel.  The conv epilogue multiplies the sums by 2^-s[co] (folded into
    its per-channel un-scale vector, Program.blob), which is exact."""
    packed = np.ascontiguousarray(packed, dtype=np.float32)
    s = row_exponents(packed) if exps is None else np.asarray(exps, np.int64)
    scaled = np.ldexp(packed, s[None, :, None].astype(np.int32)).astype(np.float32)               # exact (powers of two)
    hi = scaled.astype(np.float16)
    lo = (scaled - hi.astype(np.float32)).astype(np.float16)
    rows = np.concatenate([hi.view(np.uint16), lo.view(np.uint16)], axis=-1)
    return np.ascontiguousarray(rows).view(np.float32), s


# ---- moment propagation (plan-time activation scales) -------------------------------------------------------------------
# The half-float formats of the f16x3 / f16 modes keep all their bits for |x| in [2^-3, 65504] only (TA_FMT_SPLIT16:
# the lo half goes subnormal below; TA_FMT_F16: 2^-14).  Every CHANNEL of every tensor of a program with half-float convs is
# therefore STORED times a power of two 2^a[c] chosen at pack time so that the channel's largest expected |x| 2^a[c] lands
# near 2^10 -- 64 x of headroom to the end of the range, 13 binades of full precision below -- whatever the other channels
# of the tensor do (trained networks spread their channels over orders of magnitude).  It costs the kernels nothing: a
# consumer's weights absorb 2^-a[c] per INPUT channel (exact: powers of two, before the hi | lo split), a producer's
# per-channel epilogue vectors absorb 2^a[co] (bias, un-scale; ReLU and PReLU are positively homogeneous), tensors that are
# added (shortcuts), pooled, copied or aliased share their exponents.  The expectation comes from propagating per-channel
# (mean, variance) through the folded weights: Gaussian moments through ReLU / PReLU, independent channels, half-correlated
# filter taps.  It only has to be right to within a few binades: a channel that still overflows raises the range flag
# (TA_E_RANGE -> the wrappers' exact-f32 re-run).
_SQRT2, _SQRT2PI = np.sqrt(2.0), np.sqrt(2.0 * np.pi)
_TAP_CORR = 0.5          # share of the variance that adds coherently over the taps of a k x k filter (smooth images)
_ACT_TARGET_LOG2 = 10    # estimated max |x| 2^a in (2^9, 2^10]
_ACT_SIGMAS = 6.0
_CH_SPREAD = 8           # channel exponents of one tensor differ by at most this much


def _erf(x):
    from math import erf
    return np.vectorize(erf, otypes=[np.float64])(x)


def act_moments(mu, var, act, slope=None):
    """(mean, variance) of relu(z) / prelu(z, slope) for z ~ N(mu, var), element-wise."""
    if act == ACT_NONE:
        return mu, var
    mu, var = np.asarray(mu, np.float64), np.maximum(np.asarray(var, np.float64), 1e-60)
    sd = np.sqrt(var)
    t = mu / sd
    Phi = 0.5 * (1.0 + _erf(t / _SQRT2))
    phi = np.exp(-0.5 * t * t) / _SQRT2PI
    m_pos = mu * Phi + sd * phi
    s_pos = (mu * mu + var) * Phi + mu * sd * phi
    m_neg = -mu * (1.0 - Phi) + sd * phi
    s_neg = (mu * mu + var) * (1.0 - Phi) - mu * sd * phi
    a = np.zeros_like(mu) if (act == ACT_RELU or slope is None) else np.asarray(slope, np.float64)
    m = m_pos - a * m_neg
    s2 = s_pos + a * a * s_neg
    return m, np.maximum(s2 - m * m, 0.0)


def fold_input_affine(W, bias, scale, shift):
    """conv3x3(pad0(scale * x + shift)) + bias  ==  conv3x3(pad0(x); W') + bias16[class of the output pixel].

    The scale folds into the weights; the shift reaches an output only through the taps that are not padding, so it becomes
    one bias per border class: per axis the pixel is the first / a middle / the last / the ONLY row (column), class =
    4 * cy + cx (csrc/conv_igemm.hip: ta_border_class), 5 = interior.  W (cout, cin, 3, 3) float64 -> (W', bias16 (16, cout))."""
    W = np.asarray(W, np.float64)
    a_s, a_t = np.asarray(scale, np.float64), np.asarray(shift, np.float64)
    T = np.einsum('ocyx,c->oyx', W, a_t)                                  # what the shift adds through tap (ky, kx)
    b0 = np.zeros(W.shape[0]) if bias is None else np.asarray(bias, np.float64)
    valid = ([1, 2], [0, 1, 2], [0, 1], [1])                              # in-bounds taps of a first / middle / last / only row or column
    bias16 = np.stack([b0 + T[:, valid[cy]][:, :, valid[cx]].sum((1, 2)) for cy in range(4) for cx in range(4)])
    return W * a_s[None, :, None, None], bias16


class Program:
    """Accumulates tensors, ops and the weight region of one model."""

    def __init__(self, kind, precision='f32'):
        self.kind = kind
        self.precision = precision
        self.prec = PRECISIONS[precision]
        self.tensors = []      # (channels, halo, alias_of)
        self.ops = []
        self.wchunks = []
        self.wbytes = 0
        self.names = {}        # debug taps: name -> (tensor, ch_off, ch)
        self.extra = {}        # JSON-able notes that travel with the repack cache (save_cache / from_cache)
        self.input_tensor = None
        self.outputs = []
        self.f32_only = set()
        self.allow_split = True
        self._raw = {}         # op index -> (K x coutp float32, taps, cin_p, coutp) of the 'f16' mode's convs (see blob())
        self._chunk_at = {}    # weight-region offset -> index into wchunks
        self.lane = 0          # convs emitted while this is 1 / 2 run on that side stream (ta_op_desc.variant bits 17..18)
        # activation scales (module text above `act_moments`): per tensor and channel the expected (mean, variance) of what the
        # ops write, in program order; `_fold[op]` keeps the un-scaled epilogue vectors until blob() knows every tensor's scale
        self.stats = {}        # tensor -> [mean (C,), var (C,), written (C,) bool, amax (C,): bound on |x| the channel is expected to reach]
        self._chunk_size = {}  # weight-region offset -> bytes of a chunk reserved for blob() to fill
        self._fold = {}
        self.input_stats = None                # (mean, var) per input channel; default N(0, 1)
        self.forced_scale = {}                 # tensor (or ('mid', op index): a dw+pw block's depthwise intermediate) -> exponent
                                               # (tests: provoke / avoid the half-float range)
        self.scales_enabled = not os.environ.get('TERRAN_AMD_NO_ACT_SCALES')     # A/B switch: every tensor stored unscaled

    def tensor(self, channels, halo, alias_of=-1, name=None, f32=False):
        """f32=True pins the tensor to plain float32 (outputs read by post-processing kernels / the host).
        alias_of=-2: shape only, never materialised (the input of a program whose first op reads the frames itself)."""
        assert channels % 4 == 0
        self.tensors.append((channels, halo, alias_of))
        tid = len(self.tensors) - 1
        if f32:
            self.f32_only.add(tid)
        if name:
            self.names[name] = (tid, 0, channels)
        return tid

    def tap(self, name, tid, ch_off, ch):
        self.names[name] = (tid, ch_off, ch)

    def _w(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        off = self.wbytes
        self._chunk_at[off] = len(self.wchunks)
        self.wchunks.append(arr.tobytes())
        pad = (-len(self.wchunks[-1])) % 256
        if pad:
            self.wchunks.append(b'\0' * pad)
        self.wbytes += arr.nbytes + pad
        return off

    def _w_reserve(self, nbytes):
        """Space for a chunk blob() fills once the activation scales are known (the packed weights of a conv)."""
        off = self.wbytes
        self._chunk_at[off] = len(self.wchunks)
        self._chunk_size[off] = int(nbytes)
        self.wchunks.append(None)
        pad = (-int(nbytes)) % 256
        if pad:
            self.wchunks.append(b'\0' * pad)
        self.wbytes += int(nbytes) + pad
        return off

    def _rewrite(self, off, arr, raw=False):
        """Replace the chunk at `off` (same size)."""
        k = self._chunk_at[off]
        data = arr if raw else np.ascontiguousarray(arr, dtype=np.float32).tobytes()
        size = self._chunk_size.get(off, None if self.wchunks[k] is None else len(self.wchunks[k]))
        assert len(data) == size, (len(data), size)
        self.wchunks[k] = data

    # ---- expected moments per tensor channel -------------------------------------------------------------------------
    def _stats_of(self, tid):
        c = self.tensors[tid][0]
        a = self.tensors[tid][2]
        if a >= 0:                                    # (N,1,1,H*W*C) view of tensor `a`: position-major, channel fastest
            mu, var, wr, am = self._stats_of(a)
            rep = c // len(mu)
            return [np.tile(mu, rep), np.tile(var, rep), np.tile(wr, rep), np.tile(am, rep)]
        if tid not in self.stats:
            mu, var = np.zeros(c), np.ones(c)
            if tid == self.input_tensor and self.input_stats is not None:
                m_, v_ = self.input_stats
                mu[:len(m_)], var[:len(v_)] = m_, v_
            self.stats[tid] = [mu, var, np.zeros(c, bool), np.abs(mu) + _ACT_SIGMAS * np.sqrt(var)]
            if tid == self.input_tensor:
                self.stats[tid][2][:] = True
        return self.stats[tid]

    def _write_stats(self, tid, ch_off, mu, var, amax=None):
        """amax: bound on |x| per channel (default |mean| + 6 sigma); a rectified channel passes the bound of its positive tail,
        NOT the moments of the rectified variable -- a channel that is almost always zero still reaches that tail somewhere
        in a batch of 10^7 pixels."""
        st = self._stats_of(tid)
        n = len(mu)
        if amax is None:
            amax = np.abs(mu) + _ACT_SIGMAS * np.sqrt(var)
        new = ~st[2][ch_off:ch_off + n]
        # a slice written by several ops (ping-pong stage tensors): moments of the writer with the larger bound, largest bound
        take = new | (amax > st[3][ch_off:ch_off + n])
        st[0][ch_off:ch_off + n][take] = mu[take]
        st[1][ch_off:ch_off + n][take] = var[take]
        st[3][ch_off:ch_off + n] = np.where(new, amax, np.maximum(amax, st[3][ch_off:ch_off + n]))
        st[2][ch_off:ch_off + n] = True

    @staticmethod
    def _act_bound(mu, var, act, slope=None):
        """Bound on |act(z)| for z ~ N(mu, var): the 6-sigma tails of z pushed through the activation."""
        sd = np.sqrt(np.maximum(var, 0.0))
        hi, lo = mu + _ACT_SIGMAS * sd, _ACT_SIGMAS * sd - mu            # reach of the positive / negative tail
        if act == ACT_NONE:
            return np.maximum(hi, lo)
        # a channel the moments call (almost) always off may still fire: the mean of a deep channel is the least certain number
        # here, so a rectified channel is never given less than 4 sigma of reach
        if act == ACT_RELU or slope is None:
            return np.maximum(hi, 4.0 * sd)
        a = np.abs(np.asarray(slope, np.float64))
        return np.maximum(np.maximum(hi, a * lo), 4.0 * sd * np.maximum(a, 1.0))

    @staticmethod
    def _conv_moments(full, bias, mu_in, var_in, groups, cout):
        """full: (taps, cin_p, coutp) weights at their physical input positions; -> (mean, var) of the `cout` sums."""
        taps, cin_p, coutp = full.shape
        f64 = full.astype(np.float64)
        wsum = f64.sum(0)                                          # (cin_p, coutp)
        wsq = (f64 * f64).sum(0)
        rho = _TAP_CORR if taps > 1 else 0.0
        wvar = (1.0 - rho) * wsq + rho * wsum * wsum
        mu = np.zeros(coutp)
        var = np.zeros(coutp)
        if groups > 1:
            cg = cout // groups
            for g in range(groups):
                sl = slice(g * cg, (g + 1) * cg)
                mu[sl] = mu_in[g * cin_p:(g + 1) * cin_p] @ wsum[:, sl]
                var[sl] = var_in[g * cin_p:(g + 1) * cin_p] @ wvar[:, sl]
        else:
            mu = mu_in[:cin_p] @ wsum
            var = var_in[:cin_p] @ wvar
        b = np.zeros(coutp)
        if bias is not None:
            b[:cout] = np.asarray(bias, np.float64)
        return (mu + b)[:cout], np.maximum(var[:cout], 0.0)

    def conv(self, tin, tout, W, bias, *, stride=1, pad=None, act=ACT_NONE, in_ch_off=0, ch_pos=None, cin_p=None,
             out_ch_off=0, cout_p=None, prelu=None, res=-1, res_ch_off=0, res_up2=0, out2=-1, out2_ch_off=0,
             scale2=None, shift2=None, groups=1, variant=0, pool=False, k_split=0, precision=None, in_affine=None):
        """W: (cout, cin, kh, kw) float (BN already folded), bias: (cout,).
        ch_pos[ci] = position of true input channel ci inside the slice [in_ch_off, in_ch_off+cin_p).
        groups > 1: W is (cout, cin / groups, kh, kw) as in torch; group g reads input channels
        [in_ch_off + g cin_g, + cin_g) and writes output channels [out_ch_off + g cout_g, + cout_g); cin_g a multiple
        of 32 and cout_g of 128 (a 128-channel output tile never straddles two groups).
        variant != 0 pins the conv to one kernel variant (lib.CONV_VARIANTS; parity tests): loading fails when that
        kernel cannot run the layer.
        in_affine=(scale, shift) per input channel: the conv computes conv(pad0(scale * x + shift)) -- ArcFace's BatchNorm in
        front of a zero-padded conv (arcface/model.py:12-14) -- with the scale folded into the weights and the shift into
        one bias per border class of the output pixel (nine on ordinary maps) (the shift reaches the sum only through taps that are not padding;
        3x3, stride 1, pad 1 only).  The input tensor is then read raw: no BatchNorm'd copy of it has to exist.
        k_split > 1: the layer's K is cut in that many fixed ranges (one workgroup each, ordered reduction): for layers whose
        output is too small to fill the chip at any batch in use.  A property of the LAYER, never of the batch.
        pool=True fuses the 2x2 / 2 max-pool that follows the conv (+ activation) into its epilogue: `tout` is the POOLED
        tensor (split-role kernel only: cin % 32 == 0, cout % 64 == 0, plain epilogue)."""
        W = np.asarray(W, dtype=np.float64)
        cout, cin, kh, kw = W.shape
        bias9 = None
        if in_affine is not None:
            assert (kh, kw, stride) == (3, 3, 1) and pad in (None, 1) and groups == 1 and out2 < 0 and scale2 is None and not pool
            W, bias9 = fold_input_affine(W, bias, *in_affine)
            bias = bias9[5]                                               # (middle, middle): the interior
        if groups > 1:
            assert cout % groups == 0 and cin % 32 == 0 and (cout // groups) % 128 == 0 and ch_pos is None
        if pad is None:
            pad = kh // 2
        if pool:
            assert groups == 1 and res < 0 and out2 < 0 and stride == 1 and cin % 32 == 0 and cout % 64 == 0 and ch_pos is None
        if cin_p is None:
            cin_p = _rup(cin, 4)
        if ch_pos is None:
            ch_pos = np.arange(cin)
        if cout_p is None:
            cout_p = _rup(cout, 4)
        coutp = _rup(cout_p, 32)
        K = kh * kw * cin_p
        n_slabs = _rup(K, 32) // 32
        full = np.zeros((kh * kw, cin_p, coutp), np.float32)
        full[:, np.asarray(ch_pos), :cout] = W.transpose(2, 3, 1, 0).reshape(kh * kw, cin, cout)
        flat = np.zeros((n_slabs * 32, coutp), np.float32)
        flat[:K] = full.reshape(K, coutp)
        prec = self.prec if precision is None else PRECISIONS[precision]      # a single op may run in another arithmetic mode
        # the weights are packed by blob(): their columns absorb the input channels' activation exponents first

        def vec(v, fill=0.0):
            if v is None:
                return -1, None
            out = np.full(coutp, fill, np.float64)
            out[:cout] = np.asarray(v, dtype=np.float64)
            return self._w(out), out
        fold = dict(flat=flat, taps=kh * kw, cin_p=cin_p, K=K)
        scale2_off = -1
        if bias9 is not None:                                             # [16][coutp] in the place of the (absent) second output's scale
            t9 = np.zeros((16, coutp), np.float64)
            t9[:, :cout] = bias9
            scale2_off = self._w(t9)
            fold['bias9'] = t9
            variant |= 1 << 16
        else:
            scale2_off, fold['scale2'] = vec(scale2)
        bvec = np.zeros(coutp, np.float64)
        if bias is not None:
            bvec[:cout] = np.asarray(bias, np.float64)
        fold['bias'] = bvec
        bias_off = self._w(np.concatenate([bvec, np.ones(coutp)]))       # [bias | per-channel un-scale]: one pointer for the kernels
        wus_off = bias_off + 4 * coutp
        prelu_off, _ = vec(prelu)
        shift2_off, fold['shift2'] = vec(shift2)
        op = dict(type=OP_CONV, out=tout, in_ch_off=in_ch_off, cin=cin_p, out_ch_off=out_ch_off, cout=cout_p,
                  coutp=coutp, kh=kh, kw=kw, stride=stride, pad=pad, act=act, res=res, res_ch_off=res_ch_off,
                  res_up2=res_up2, out2=out2, out2_ch_off=out2_ch_off, n_slabs=n_slabs, prec=prec,
                  groups=groups, variant=variant | (int(k_split) << 8) | (self.lane << 17), pool=int(bool(pool)), wscale_log2=0,
                  w_off=self._w_reserve(n_slabs * coutp * 128), bias_off=bias_off, prelu_off=prelu_off, scale2_off=scale2_off,
                  shift2_off=shift2_off, wus_off=wus_off, macs_per_pixel=float(cout * cin * kh * kw))
        op['in'] = tin
        self._fold[len(self.ops)] = fold
        self.ops.append(op)
        # ---- expected moments of what this op writes
        st_in = self._stats_of(tin)
        span = cin_p * max(groups, 1)
        mu, var = self._conv_moments(full, bias, st_in[0][in_ch_off:in_ch_off + span], st_in[1][in_ch_off:in_ch_off + span],
                                     groups, cout)
        slope = None if prelu is None else np.asarray(prelu, np.float64)
        bound = self._act_bound(mu, var, act, slope)
        mu, var = act_moments(mu, var, act, slope)
        if res >= 0:
            st_r = self._stats_of(res)
            mu = mu + st_r[0][res_ch_off:res_ch_off + cout]
            var = var + st_r[1][res_ch_off:res_ch_off + cout]
            bound = bound + st_r[3][res_ch_off:res_ch_off + cout]
        if pool:                                                          # max of four ~ independent values
            mu, var = mu + 1.03 * np.sqrt(var), 0.49 * var
        self._write_stats(tout, out_ch_off, mu, var, bound)
        if out2 >= 0:
            s2, h2 = np.asarray(scale2, np.float64), np.asarray(shift2, np.float64)
            self._write_stats(out2, out2_ch_off, mu * s2 + h2, var * s2 * s2, np.abs(s2) * bound + np.abs(h2))

    def dwconv(self, tin, tout, W, bias, *, stride=1, relu=True):
        """W: (C,1,3,3) folded, bias (C,).  (The layer-by-layer detector program: float32 tensors, stored unscaled.)"""
        C = W.shape[0]
        w9 = np.asarray(W, dtype=np.float64).reshape(C, 9).T            # [9][C]
        op = dict(type=OP_DWCONV, out=tout, in_ch_off=0, cin=C, out_ch_off=0, cout=C, coutp=_rup(C, 32), kh=3,
                  kw=3, stride=stride, pad=1, act=ACT_RELU if relu else ACT_NONE, res=-1, res_ch_off=0, res_up2=0,
                  out2=-1, out2_ch_off=0, n_slabs=0, prec=0, groups=1, variant=0, pool=0, wscale_log2=0, w_off=self._w(w9), bias_off=self._w(bias),
                  prelu_off=-1, scale2_off=-1, shift2_off=-1, wus_off=-1, macs_per_pixel=float(C * 9))
        op['in'] = tin
        self.ops.append(op)
        st = self._stats_of(tin)
        mu, var = self._dw_moments(w9, bias, st[0][:C], st[1][:C])
        act = ACT_RELU if relu else ACT_NONE
        self._write_stats(tout, 0, *act_moments(mu, var, act), self._act_bound(mu, var, act))

    @staticmethod
    def _dw_moments(w9, bias, mu_in, var_in):
        w9 = np.asarray(w9, np.float64)
        wsum, wsq = w9.sum(0), (w9 * w9).sum(0)
        return (np.asarray(bias, np.float64) + mu_in * wsum,
                var_in * ((1.0 - _TAP_CORR) * wsq + _TAP_CORR * wsum * wsum))

    def rfstem(self, tin, tout, Ws, bs, Wd, bd, Wp, bp, Wd2=None, bd2=None, Wp2=None, bp2=None):
        """RetinaFace front as ONE op: conv3x3 s2 (3 -> 8) -> depthwise 3x3 (8) -> 1x1 (8 -> 16), each + folded BN + ReLU.
        Ws (8,3,3,3) / bs (8,), Wd (8,1,3,3) / bd (8,), Wp (16,8,1,1) / bp (16,), all already folded.
        With Wd2 (16,1,3,3) / bd2 (16,), Wp2 (32,16,1,1) / bp2 (32,): the NEXT block of the base -- depthwise 3x3 stride 2 (16) ->
        1x1 (16 -> 32), retinaface/model.py:26-39 -- in the same kernel: `tout` is the 32-channel quarter-resolution map and
        the 16-channel half-resolution map (the largest tensor of the network) never reaches HBM."""
        parts = [np.asarray(Ws, np.float64).reshape(8, 27).ravel(), np.asarray(bs, np.float64),
                 np.asarray(Wd, np.float64).reshape(8, 9).T.ravel(), np.asarray(bd, np.float64),
                 np.asarray(Wp, np.float64).reshape(16, 8).ravel(), np.asarray(bp, np.float64)]
        fuse = Wd2 is not None
        if fuse:                        # [9][16] depthwise taps, [16] bias, [16][32] 1x1 as (c, oc), [32] bias: what rf_stem_kernel<true> reads
            parts += [np.asarray(Wd2, np.float64).reshape(16, 9).T.ravel(), np.asarray(bd2, np.float64),
                      np.asarray(Wp2, np.float64).reshape(32, 16).T.ravel(), np.asarray(bp2, np.float64)]
        blob = np.concatenate(parts)
        assert blob.size == (448 + 704 if fuse else 448)
        op = dict(type=OP_RFSTEM, out=tout, in_ch_off=0, cin=4, out_ch_off=0, cout=32 if fuse else 16, coutp=32, kh=3, kw=3, stride=2,
                  pad=1, act=ACT_RELU, res=-1, res_ch_off=0, res_up2=0, out2=-1, out2_ch_off=0, n_slabs=0, prec=0, groups=1,
                  variant=0, pool=0, wscale_log2=0, w_off=self._w(blob), bias_off=-1, prelu_off=-1, scale2_off=-1, shift2_off=-1,
                  wus_off=-1, macs_per_pixel=float(8 * 27 + 16 * 8) + (32 * 16 / 4.0 if fuse else 0.0))
        op['in'] = tin
        self._fold[len(self.ops)] = dict(rfstem=parts)
        self.ops.append(op)
        # moments: the frames are raw 0..255 BGR pixels (the kernel reads them itself: no float copy exists)
        st = self._stats_of(tin)
        ws = np.asarray(Ws, np.float64)                                   # (8, 3, 3, 3)
        wsum, wsq = ws.sum((2, 3)), (ws * ws).sum((2, 3))                 # (8, 3)
        mu = np.asarray(bs, np.float64) + wsum @ st[0][:3]
        var = ((1.0 - _TAP_CORR) * wsq + _TAP_CORR * wsum * wsum) @ st[1][:3]
        mu, var = act_moments(mu, var, ACT_RELU)
        mu, var = self._dw_moments(np.asarray(Wd, np.float64).reshape(8, 9).T, bd, mu, var)
        mu, var = act_moments(mu, var, ACT_RELU)
        wp = np.asarray(Wp, np.float64).reshape(16, 8)
        mu, var = np.asarray(bp, np.float64) + wp @ mu, (wp * wp) @ var
        if fuse:
            mu, var = act_moments(mu, var, ACT_RELU)
            mu, var = self._dw_moments(np.asarray(Wd2, np.float64).reshape(16, 9).T, bd2, mu, var)
            mu, var = act_moments(mu, var, ACT_RELU)
            wp2 = np.asarray(Wp2, np.float64).reshape(32, 16)
            mu, var = np.asarray(bp2, np.float64) + wp2 @ mu, (wp2 * wp2) @ var
        self._write_stats(tout, 0, *act_moments(mu, var, ACT_RELU), self._act_bound(mu, var, ACT_RELU))

    def dwpw(self, tin, tout, Wd, bd, Wp, bp, *, stride=1, precision=None):
        """Depthwise 3x3 (stride 1 / 2, pad 1) + ReLU fused into the following 1x1 conv + ReLU (both BN-folded):
        Wd (C,1,3,3) / bd (C,), Wp (cout, C, 1, 1) / bp (cout,).  The depthwise part is float32 FMAs; the 1x1 runs on the
        exact-f32 MFMA or (precision='f16x3') on the split-half MFMA."""
        prec = self.prec if precision is None else PRECISIONS[precision]
        assert prec in (0, 3), 'the fused depthwise + pointwise block exists in the f32 and f16x3 modes'
        C = Wd.shape[0]
        Wp = np.asarray(Wp, np.float64)
        cout = Wp.shape[0]
        assert Wp.shape[1] == C and C % 4 == 0 and cout % 4 == 0
        coutp = _rup(cout, 32)
        n_slabs = _rup(C, 32) // 32
        flat = np.zeros((n_slabs * 32, coutp), np.float32)
        flat[:C, :cout] = Wp.reshape(cout, C).T
        bias = np.zeros(coutp, np.float64)
        bias[:cout] = np.asarray(bp, np.float64)
        w9 = np.asarray(Wd, np.float64).reshape(C, 9).T                                          # [9][C]
        bd = np.asarray(bd, np.float64)
        op = dict(type=OP_DWPW, out=tout, in_ch_off=0, cin=C, out_ch_off=0, cout=cout, coutp=coutp, kh=1, kw=1,
                  stride=stride, pad=0, act=ACT_RELU, res=-1, res_ch_off=0, res_up2=0, out2=-1, out2_ch_off=0,
                  n_slabs=n_slabs, prec=prec, groups=1, variant=0, pool=0, wscale_log2=0, w_off=self._w_reserve(n_slabs * coutp * 128),
                  bias_off=self._w(np.concatenate([bias, np.ones(coutp)])), prelu_off=-1,
                  scale2_off=self._w(w9), shift2_off=self._w(bd), wus_off=-1,
                  macs_per_pixel=float(cout * C))
        op['wus_off'] = op['bias_off'] + 4 * coutp
        op['in'] = tin
        # moments: depthwise (+ ReLU) -> the intermediate that is split into half floats in registers -> 1x1 (+ ReLU)
        st = self._stats_of(tin)
        mu0, var0 = self._dw_moments(w9, bd, st[0][:C], st[1][:C])
        mid_bound = self._act_bound(mu0, var0, ACT_RELU)                  # per channel: the depthwise result is per channel
        mu, var = act_moments(mu0, var0, ACT_RELU)
        self._fold[len(self.ops)] = dict(flat=flat, bias=bias, dw_w=w9, dw_b=bd, mid_bound=mid_bound)
        self.ops.append(op)
        wp = Wp.reshape(cout, C)
        mu2, var2 = bias[:cout] + wp @ mu, (wp * wp) @ var
        self._write_stats(tout, 0, *act_moments(mu2, var2, ACT_RELU), self._act_bound(mu2, var2, ACT_RELU))

    def simple(self, typ, tin, tout, in_ch_off=0, out_ch_off=0, ch=0):
        op = dict(type=typ, out=tout, in_ch_off=in_ch_off, cin=ch, out_ch_off=out_ch_off, cout=ch, coutp=0, kh=2,
                  kw=2, stride=2, pad=0, act=0, res=-1, res_ch_off=0, res_up2=0, out2=-1, out2_ch_off=0, n_slabs=0,
                  prec=0, groups=1, variant=0, pool=0, wscale_log2=0, w_off=-1, bias_off=-1, prelu_off=-1, scale2_off=-1, shift2_off=-1,
                  wus_off=-1, macs_per_pixel=0.0)
        op['in'] = tin
        self.ops.append(op)
        st = self._stats_of(tin)
        if typ == OP_MAXPOOL:
            self._write_stats(tout, 0, st[0] + 1.03 * np.sqrt(st[1]), 0.49 * st[1], st[3].copy())
        else:
            self._write_stats(tout, out_ch_off, st[0][in_ch_off:in_ch_off + ch].copy(), st[1][in_ch_off:in_ch_off + ch].copy(),
                              st[3][in_ch_off:in_ch_off + ch].copy())

    @classmethod
    def from_cache(cls, path):
        """Load a program written by `save_cache` (packed blob + debug-tap names)."""
        import json
        with open(path, 'rb') as f:
            head = f.read(16)
            if head[:8] != b'TAMCACHE':
                raise ValueError('not a pack cache file')
            n = int.from_bytes(head[8:16], 'little')
            meta = json.loads(f.read(n).decode())
            blob = f.read()
        self = cls(meta['kind'], meta['precision'])
        self.names = {k: tuple(v) for k, v in meta['names'].items()}
        self.outputs = meta['outputs']
        self.extra = dict(meta.get('extra') or {})
        self._blob = blob
        return self

    def save_cache(self, path):
        import json
        import os
        # `extra`: decisions taken about this program after packing (arcface.guard_f16x2's calibration result) -- kept with the blob
        meta = json.dumps({'kind': self.kind, 'precision': self.precision, 'names': self.names,
                           'outputs': [int(o) for o in self.outputs], 'extra': getattr(self, 'extra', {})}).encode()
        tmp = path + '.tmp.%d' % os.getpid()
        with open(tmp, 'wb') as f:
            f.write(b'TAMCACHE' + len(meta).to_bytes(8, 'little') + meta + self.blob())
        os.replace(tmp, path)                      # atomic: concurrent ranks may race to write the same file

    def tensor_formats(self):
        """Storage format per tensor.  In the bf16 modes a tensor is kept PRE-SPLIT -- per pixel and 32-channel
        block, 32 bf16 `hi` then 32 bf16 `lo` (x = hi + lo; same 4 bytes per element as float32) -- when all of its
        conv consumers are whole-block reads by the pipelined kernel, whose MFMA operand fragments then come
        straight out of LDS with no conversion VALU.  Everything else stays float32."""
        n = len(self.tensors)
        # the arithmetic mode of a tensor's conv consumers decides its split format (a program may mix modes per op: the
        # detector's base runs exact f32, its refiner f16x3); consumers of different modes -> float32
        cprec = {}
        for op in self.ops:
            if op['type'] == OP_CONV:
                cprec.setdefault(op['in'], set()).add(op['prec'])
        fmt = []
        for t, (c, _, _) in enumerate(self.tensors):
            precs = cprec.get(t, {self.prec})
            p = next(iter(precs)) if len(precs) == 1 else 0
            fmt.append(SPLIT_FMT[p] if (self.allow_split and c % (64 if p == 4 else 32) == 0) else FMT_F32)
        for t in self.f32_only | {self.input_tensor}:
            fmt[t] = FMT_F32
        for op in self.ops:
            if op['type'] == OP_CONV:
                taps = op['kh'] * op['kw']
                pipe = (op['cin'] % 32 == 0 and op['in_ch_off'] % 32 == 0 and op['coutp'] % 64 == 0
                        and op['n_slabs'] >= 2 and op['n_slabs'] == taps * (op['cin'] // 32))
                if op['prec'] == 4:                          # half-float tensors: whole-tensor reads in slabs of 64 channels
                    k64 = op['cin'] % 64 == 0 and op['n_slabs'] == taps * (op['cin'] // 64)      # already re-packed by blob()
                    pipe = (pipe or (k64 and op['coutp'] % 64 == 0)) and \
                        op['cin'] % 64 == 0 and op['in_ch_off'] == 0 and op['groups'] == 1
                # the kernels that read pre-split tensors drain through LDS only: every channel slice of the op on an 8-channel
                # boundary (conv_igemm.hip: variant_eligible `staged`); anything else runs on the generic kernel, float32 in
                if op['out_ch_off'] % 8 or op['res_ch_off'] % 8 or op['out2_ch_off'] % 8:
                    pipe = False
                if not pipe:
                    fmt[op['in']] = FMT_F32
            elif op['type'] in (OP_DWPW, OP_RFSTEM):            # float32 in, float32 out
                fmt[op['in']] = FMT_F32
                if op['type'] == OP_RFSTEM:
                    fmt[op['out']] = FMT_F32
            elif op['type'] == OP_COPYCH:
                if op['cin'] % 32 or op['in_ch_off'] % 32 or op['out_ch_off'] % 32:
                    fmt[op['in']] = fmt[op['out']] = FMT_F32
            if op['type'] != OP_CONV:                       # half-float tensors are the conv kernels' alone
                for t in (op['in'], op['out']):
                    if fmt[t] == FMT_F16:
                        fmt[t] = FMT_F32
        changed = True
        while changed:                                  # aliases share memory; copies are raw
            changed = False
            for t, (_, _, a) in enumerate(self.tensors):
                if a >= 0 and fmt[t] != fmt[a]:
                    fmt[t] = fmt[a] = FMT_F32
                    changed = True
            for op in self.ops:
                if op['type'] == OP_COPYCH and fmt[op['in']] != fmt[op['out']]:
                    fmt[op['in']] = fmt[op['out']] = FMT_F32
                    changed = True
        return fmt

    def expected_amax(self, tid, per_channel=False):
        """Largest |x| the packer expects in tensor `tid` (per channel: the bound of every channel some op writes, else 0)."""
        st = self._stats_of(tid)
        am = np.where(st[2], st[3], 0.0)
        return am if per_channel else float(am.max()) if len(am) else 0.0

    def tensor_scales(self):
        """Exponents a[c] per tensor and channel: channel c is STORED times 2^a[c] (module text above `act_moments`).  All 0
        unless the program has half-float convs; 0 for the input, for tensors the host / post-processing kernels read
        unscaled (f32_only) and for anything a plain depthwise op touches.  Channels that are added (shortcuts), pooled,
        copied or seen through an alias share one exponent (union-find over (tensor, channel) nodes); a channel's exponent puts
        the largest bound of its group in (2^9, 2^10], but no channel sits more than 2^_CH_SPREAD below its tensor's largest."""
        n = len(self.tensors)
        sizes = [t[0] for t in self.tensors]
        scales = [np.zeros(c, np.int64) for c in sizes]
        if not self.scales_enabled or not any(op['type'] in (OP_CONV, OP_DWPW) and op['prec'] in (3, 4, 5) for op in self.ops):
            return scales
        base = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        parent = np.arange(base[-1], dtype=np.int64)

        def find(x):
            r = x
            while parent[r] != r:
                r = parent[r]
            while parent[x] != r:
                parent[x], x = r, parent[x]
            return r

        def union_run(ta, ca, tb, cb, cnt):
            for i in range(cnt):
                ra, rb = find(base[ta] + ca + i), find(base[tb] + cb + i)
                if ra != rb:
                    parent[ra] = rb
        for t, (c, _, a) in enumerate(self.tensors):
            if a >= 0:
                ca = sizes[a]
                for p in range(c):
                    ra, rb = find(base[t] + p), find(base[a] + p % ca)
                    if ra != rb:
                        parent[ra] = rb
        fixed_t = set(self.f32_only) | {self.input_tensor}
        for op in self.ops:
            if op['type'] == OP_MAXPOOL:
                union_run(op['in'], 0, op['out'], 0, sizes[op['in']])
            elif op['type'] == OP_COPYCH:
                union_run(op['in'], op['in_ch_off'], op['out'], op['out_ch_off'], op['cin'])
            elif op['type'] == OP_DWCONV:
                fixed_t |= {op['in'], op['out']}
            elif op['type'] == OP_CONV and op['res'] >= 0:
                union_run(op['res'], op['res_ch_off'], op['out'], op['out_ch_off'], op['cout'])
        amax = np.concatenate([self.expected_amax(t, per_channel=True) for t in range(n)])
        # no channel more than 2^_CH_SPREAD below its tensor's largest: a channel's own bound is a noisier number than the tensor's
        # (a mis-predicted weak channel must not be blown up into the end of the range), and 2^8 of relief already keeps a
        # channel 2^14 below the tensor's largest inside the window where hi + lo carries all its bits
        for t in range(n):
            seg = amax[base[t]:base[t + 1]]
            if len(seg) and seg.max() > 0:
                np.maximum(seg, np.where(seg > 0, seg.max() * 2.0 ** -_CH_SPREAD, 0.0), out=seg)
        roots = np.array([find(i) for i in range(base[-1])], np.int64)
        gmax = np.zeros(base[-1])
        np.maximum.at(gmax, roots, amax)
        fixed = np.zeros(base[-1], bool)
        for t in fixed_t:
            fixed[roots[base[t]:base[t + 1]]] = True
        forced = {}
        for t, e in self.forced_scale.items():
            if isinstance(t, (int, np.integer)):
                for r in roots[base[t]:base[t + 1]]:
                    forced[int(r)] = int(e)
        g = gmax[roots]
        ok = np.isfinite(g) & (g > 0) & ~fixed[roots]
        expo = np.zeros(base[-1], np.int64)
        expo[ok] = np.clip(_ACT_TARGET_LOG2 - np.ceil(np.log2(g[ok])), -40, 40).astype(np.int64)
        for r, e in forced.items():
            if not fixed[r]:
                expo[roots == r] = e
        return [expo[base[t]:base[t + 1]].copy() for t in range(n)]

    def _pack_conv_weights(self, op, f, a_in):
        """The [slab][cout][32] weight image of conv `op` with 2^-a_in[c] folded into the columns of input channel c (exact),
        split into half floats / bf16 where the op's mode wants it.  -> (bytes, row exponents s[coutp])."""
        flat, taps, cin_p, K = f['flat'], f['taps'], f['cin_p'], f['K']
        coutp, groups = op['coutp'], max(op['groups'], 1)
        cols = a_in[op['in_ch_off']:op['in_ch_off'] + cin_p * groups]
        fs = flat
        if np.any(cols != 0):
            fs = flat.copy()
            if groups > 1:
                cg = op['cout'] // groups
                for g_ in range(groups):
                    e = -np.tile(cols[g_ * cin_p:(g_ + 1) * cin_p], taps).astype(np.int32)
                    fs[:K, g_ * cg:(g_ + 1) * cg] = np.ldexp(flat[:K, g_ * cg:(g_ + 1) * cg], e[:, None])
            else:
                fs[:K] = np.ldexp(flat[:K], -np.tile(cols[:cin_p], taps).astype(np.int32)[:, None])
        n_slabs = flat.shape[0] // 32
        packed = fs.reshape(n_slabs, 32, coutp).transpose(0, 2, 1)          # [slab][cout][32]
        wexp = np.zeros(coutp, np.int64)
        if op['prec'] in (3, 4, 5):
            packed, wexp = split_f16_rows(packed)
        elif op['prec'] != 0:
            packed = split_bf16_rows(np.ascontiguousarray(packed))
        return np.ascontiguousarray(packed, dtype=np.float32).tobytes(), wexp, fs

    def blob(self):
        if getattr(self, '_blob', None) is not None:
            return self._blob
        hdr = np.zeros(1, HEADER_DT)
        tens = np.zeros(len(self.tensors), TENSOR_DT)
        fmts = self.tensor_formats()
        scales = self.tensor_scales()
        self.scales = scales
        # per tensor with a non-zero exponent anywhere: [C] floats 2^-a[c] in the weights region (debug taps and the pose
        # post-processing multiply what they read with it)
        unscale_off = []
        for i, (c, h, a) in enumerate(self.tensors):
            unscale_off.append(self._w(np.ldexp(np.ones(c), -scales[i].astype(np.int32))) if np.any(scales[i] != 0) else -1)
            assert unscale_off[-1] < 2 ** 31
            tens[i] = (c, h, a, fmts[i], unscale_off[-1])
        # epilogue vectors with every power of two folded in: the sums of channel co arrive times 2^s[co] (the activation
        # exponents of the input channels are inside the weights), the results leave times 2^a_out[co]
        self.mid_scales = {}
        for i, f in self._fold.items():
            op = self.ops[i]
            a_in = scales[op['in']]
            a_out = np.zeros(op['coutp'], np.int64)
            if op['type'] in (OP_CONV, OP_DWPW, OP_RFSTEM):
                seg = scales[op['out']][op['out_ch_off']:op['out_ch_off'] + op['cout']]
                a_out[:len(seg)] = seg
            up = np.ldexp(1.0, a_out.astype(np.int32))
            if op['type'] == OP_CONV:
                data, wexp, fs = self._pack_conv_weights(op, f, a_in)
                # 'f16' mode: a conv whose input tensor is stored as plain half floats (TA_FMT_F16) walks K in slabs of 64
                # channels -- a slab row is 64 halfs of ONE operand, not [hi x32 | lo x32] -- half the bytes of the image reserved
                if op['prec'] == 4 and fmts[op['in']] == FMT_F16:
                    cin_p, K, coutp = f['cin_p'], f['K'], op['coutp']
                    assert cin_p % 64 == 0 and op['in_ch_off'] == 0 and op['groups'] == 1
                    rows = np.ldexp(fs[:K].reshape(K // 64, 64, coutp).transpose(0, 2, 1), wexp[None, :, None].astype(np.int32)).astype(np.float16)   # [slab][cout][64]
                    assert rows.nbytes * 2 == len(data)
                    data = rows.tobytes() + b'\0' * rows.nbytes
                    op['n_slabs'] = K // 64
                    f['rows64'] = rows
                f['wexp'] = wexp
                self._rewrite(op['w_off'], data, raw=True)
                self._rewrite(op['bias_off'], np.concatenate([f['bias'] * up, np.ldexp(np.ones(op['coutp']), (a_out - wexp).astype(np.int32))]))
                if 'bias9' in f:
                    self._rewrite(op['scale2_off'], f['bias9'] * up[None, :])
                elif op['out2'] >= 0:
                    a2 = np.zeros(op['coutp'], np.int64)
                    seg = scales[op['out2']][op['out2_ch_off']:op['out2_ch_off'] + op['cout']]
                    a2[:len(seg)] = seg
                    self._rewrite(op['scale2_off'], f['scale2'] * np.ldexp(1.0, (a2 - a_out).astype(np.int32)))
                    self._rewrite(op['shift2_off'], f['shift2'] * np.ldexp(1.0, a2.astype(np.int32)))
            elif op['type'] == OP_DWPW:
                # the depthwise result is split into half floats in registers (f16x3): per channel an exponent of its own
                C = op['cin']
                a_mid = np.zeros(C, np.int64)
                if op['prec'] == 3 and self.scales_enabled:
                    mb = f['mid_bound']
                    okc = np.isfinite(mb) & (mb > 0)
                    mbf = np.maximum(mb, mb[okc].max() * 2.0 ** -_CH_SPREAD) if okc.any() else mb
                    a_mid[okc] = np.clip(_ACT_TARGET_LOG2 - np.ceil(np.log2(mbf[okc])), -40, 40).astype(np.int64)
                elif op['prec'] == 0:
                    a_mid = a_in[:C].copy()                             # exact f32: any power of two gives the same bits
                if ('mid', i) in self.forced_scale:
                    a_mid[:] = int(self.forced_scale[('mid', i)])
                self.mid_scales[i] = a_mid
                self._rewrite(op['scale2_off'], f['dw_w'] * np.ldexp(1.0, (a_mid - a_in[:C]).astype(np.int32))[None, :])
                self._rewrite(op['shift2_off'], f['dw_b'] * np.ldexp(1.0, a_mid.astype(np.int32)))
                fl = f['flat'].copy()
                fl[:C] = np.ldexp(fl[:C], -a_mid.astype(np.int32)[:, None])
                n_slabs = fl.shape[0] // 32
                packed = np.ascontiguousarray(fl.reshape(n_slabs, 32, op['coutp']).transpose(0, 2, 1))
                wexp = np.zeros(op['coutp'], np.int64)
                if op['prec'] == 3:
                    packed, wexp = split_f16_rows(packed)
                self._rewrite(op['w_off'], np.ascontiguousarray(packed, dtype=np.float32).tobytes(), raw=True)
                self._rewrite(op['bias_off'], np.concatenate([f['bias'] * up, np.ldexp(np.ones(op['coutp']), (a_out - wexp).astype(np.int32))]))
            elif op['type'] == OP_RFSTEM:
                parts = list(f['rfstem'])
                if len(parts) == 10:        # fused second block: the stored tensor is ITS output; the 16-channel map stays inside the kernel
                    parts[8] = (parts[8].reshape(16, 32) * up[None, :32]).ravel()
                    parts[9] = parts[9] * up[:32]
                else:
                    parts[4] = (parts[4].reshape(16, 8) * up[:16, None]).ravel()
                    parts[5] = parts[5] * up[:16]
                self._rewrite(op['w_off'], np.concatenate(parts))
        ops = np.zeros(len(self.ops), OP_DT)
        for i, op in enumerate(self.ops):
            for k, v in op.items():
                ops[i][k] = v
        t_off = HEADER_DT.itemsize
        o_off = t_off + tens.nbytes
        w_off = _rup(o_off + ops.nbytes, 256)
        outs = np.full(16, -1, np.int32)
        outs[:len(self.outputs)] = self.outputs
        hdr[0] = (MAGIC, BLOB_VERSION, self.kind, len(self.tensors), len(self.ops), self.input_tensor, len(self.outputs), outs,
                  t_off, o_off, w_off, self.wbytes)
        head = hdr.tobytes() + tens.tobytes() + ops.tobytes()
        for f in self._fold.values():                 # the float32 weight matrices are not needed any more
            f.pop('flat', None)
        self._blob = head + b'\0' * (w_off - len(head)) + b''.join(self.wchunks)     # a program is packed once
        return self._blob


# ---- BatchNorm folding -------------------------------------------------------------------
def _bn_affine(sd, key, eps):
    g = np.asarray(sd[key + '.weight'], np.float64)
    b = np.asarray(sd[key + '.bias'], np.float64)
    m = np.asarray(sd[key + '.running_mean'], np.float64)
    v = np.asarray(sd[key + '.running_var'], np.float64)
    s = g / np.sqrt(v + eps)
    return s, b - m * s


def _fold(W, bias, scale, shift):
    """BN(conv(x)+bias) -> conv'(x)+bias'."""
    W = np.asarray(W, np.float64) * scale[:, None, None, None]
    b0 = np.zeros(W.shape[0]) if bias is None else np.asarray(bias, np.float64)
    return W, b0 * scale + shift


# ---- OpenPose ------------------------------------------------------------------------------
# Concat tensor channel layout (192): feat 0..127 | PAF 128..165 (+2 zero) | HM 168..186 (+5 zero).
OP_FEAT, OP_PAF, OP_HM, OP_XCH = 0, 128, 168, 192


def pack_openpose(sd, precision='f32'):
    """openpose/model.py:27-141.  Stage inputs cat[PAF, HM, feat] live in two ping-pong
    192-channel tensors; every stage-output conv writes its slice directly."""
    if precision in ('f16', 'f16x2'):   # the embedder's tolerance modes are for networks without discrete decisions: pose keeps 22 bits
        precision = 'f16x3'
    P = Program(MODEL_OPENPOSE, precision)
    t = P.tensor(4, 1, name='input')
    P.input_tensor = t
    P.input_stats = (np.array([-0.05, -0.05, -0.05, 0.0]), np.array([0.08, 0.08, 0.08, 0.0]))   # RGB / 255 - 0.5 of natural images
    X0 = P.tensor(OP_XCH, 3, name='X0')
    X1 = P.tensor(OP_XCH, 3, name='X1')
    items = arch.OPENPOSE_MODEL0
    fuse_pool = not os.environ.get('TERRAN_AMD_NO_FUSED_POOL')     # A/B switch: separate max-pool launches
    skip_pool = False
    for i, item in enumerate(items):
        nxt = items[i + 1] if i + 1 < len(items) else None
        if item[0] == 'pool':
            if skip_pool:                                           # already done in the previous conv's epilogue
                skip_pool = False
                continue
            o = P.tensor(P.tensors[t][0], 1)
            P.simple(OP_MAXPOOL, t, o)
            t = o
            continue
        name, cin, cout, k = item
        W, b = sd['model0.%s.weight' % name], sd['model0.%s.bias' % name]
        if nxt is None:       # conv4_4_CPM -> feature slice of X0
            P.conv(t, X0, W, b, act=ACT_RELU, out_ch_off=OP_FEAT)
            P.tap('feat', X0, OP_FEAT, 128)
        elif nxt[0] == 'pool' and fuse_pool and cin % 32 == 0 and cout % 64 == 0:
            # conv + ReLU + 2x2 max-pool in one launch: the full-resolution map (493 MB per 32 frames after conv1_2 at
            # 184 x 327) is never written; max commutes with nothing here -- it is applied to the very values the
            # separate pool would have read
            o = P.tensor(cout, 1, name=name + '_pooled')
            P.conv(t, o, W, b, act=ACT_RELU, pool=True)
            t = o
            skip_pool = True
        else:
            o = P.tensor(cout, 0 if nxt[0] == 'pool' else 1, name=name)
            P.conv(t, o, W, b, act=ACT_RELU)
            t = o
    P.simple(OP_COPYCH, X0, X1, OP_FEAT, OP_FEAT, 128)

    # true input channel (cat order PAF38, HM19, feat128) -> position in the 192-channel tensor
    cat_pos = np.concatenate([OP_PAF + np.arange(38), OP_HM + np.arange(19), OP_FEAT + np.arange(128)])
    for st in range(1, 7):
        xin, xout = (X0, X1) if st % 2 == 1 else (X1, X0)
        # The first conv of the PAF branch and of the heat-map branch read the same stage input with the same
        # kernel size: one launch with their output channels side by side (128 | 128); each branch then goes on
        # from its channel slice.  Per output channel the arithmetic is unchanged.
        l1, l2 = arch.openpose_stage_layers(st, 1), arch.openpose_stage_layers(st, 2)
        (n1, cin, c1, k, r1), (n2, cin2, c2, k2, r2) = l1[0], l2[0]
        assert (cin, k, r1) == (cin2, k2, r2) and c1 == c2 == 128
        W = np.concatenate([np.asarray(sd['model%d_%d.%s.weight' % (st, br, n)]) for br, n in ((1, n1), (2, n2))])
        b = np.concatenate([np.asarray(sd['model%d_%d.%s.bias' % (st, br, n)]) for br, n in ((1, n1), (2, n2))])
        kw = dict(ch_pos=cat_pos, cin_p=OP_XCH) if cin == 185 else dict(in_ch_off=OP_FEAT)   # stage 1: feature slice only
        cur = P.tensor(c1 + c2, l1[1][3] // 2)
        P.conv(xin, cur, W, b, act=ACT_RELU if r1 else ACT_NONE, **kw)
        # The middle layers of the two branches have identical shapes (model.py:56-86): each pair runs as ONE grouped
        # conv (groups=2) over the side-by-side 128 | 128 (or 512 | 512) channels -- twice the tiles per launch, which
        # is what lets the 1080p-sized maps use the 128 x 256 tile, and half the launches.
        grouped = not os.environ.get('TERRAN_AMD_NO_GROUPED')        # A/B switch: one conv per branch instead
        for li in range(1, len(l1) - 1):
            (na, cin, cout, k, relu), (nb, cin2, cout2, k2, relu2) = l1[li], l2[li]
            if not grouped:
                o = P.tensor(2 * cout, l1[li + 1][3] // 2)
                for br, n in ((1, na), (2, nb)):
                    key = 'model%d_%d.%s' % (st, br, n)
                    P.conv(cur, o, sd[key + '.weight'], sd[key + '.bias'], act=ACT_RELU if relu else ACT_NONE,
                           in_ch_off=(br - 1) * cin, out_ch_off=(br - 1) * cout)
                cur = o
                continue
            assert (cin, cout, k, relu) == (cin2, cout2, k2, relu2) and cin % 32 == 0 and cout % 128 == 0
            W = np.concatenate([np.asarray(sd['model%d_%d.%s.weight' % (st, br, n)]) for br, n in ((1, na), (2, nb))])
            b = np.concatenate([np.asarray(sd['model%d_%d.%s.bias' % (st, br, n)]) for br, n in ((1, na), (2, nb))])
            o = P.tensor(2 * cout, l1[li + 1][3] // 2)
            P.conv(cur, o, W, b, act=ACT_RELU if relu else ACT_NONE, groups=2)
            cur = o
        # Output convs (1x1 -> 38 PAF / 19 heat-map channels, adjacent slices 128..167 | 168..187 of xout).  Where both are
        # linear and narrow (stages 2-5) they run as ONE launch: rows [PAF 38 + 2 zero | HM 19 + 1 zero] over all 2 cin input channels
        # with zero weights on the other branch's half -- exact zeros, so every output keeps its bits.  Stage 6 keeps two
        # launches: its heat-map conv is followed by a ReLU (the reference's `no_relu_layers` typo), its PAF conv is not.
        (np_, cin, co1, k, r1), (nh, cin2, co2, k2, r2) = l1[-1], l2[-1]
        if r1 == r2 and cin == cin2 <= 128 and not os.environ.get('TERRAN_AMD_NO_MERGED_OUTPUTS'):   # (stage 1: 2 x 512 inputs, no gain)
            Wp, bp = np.asarray(sd['model%d_1.%s.weight' % (st, np_)]), np.asarray(sd['model%d_1.%s.bias' % (st, np_)])
            Wh, bh = np.asarray(sd['model%d_2.%s.weight' % (st, nh)]), np.asarray(sd['model%d_2.%s.bias' % (st, nh)])
            Wm = np.zeros((60, 2 * cin, 1, 1), np.float32)
            bm = np.zeros(60, np.float32)
            Wm[0:co1, :cin], bm[0:co1] = Wp, bp
            Wm[40:40 + co2, cin:], bm[40:40 + co2] = Wh, bh
            P.conv(cur, xout, Wm, bm, act=ACT_RELU if r1 else ACT_NONE, out_ch_off=OP_PAF, cout_p=60)
            P.ops[-1]['macs_per_pixel'] = float((co1 + co2) * cin)          # algorithmic work: the two real convs
        else:
            for br, layers in ((1, l1), (2, l2)):
                name, cin, cout, k, relu = layers[-1]
                key = 'model%d_%d.%s' % (st, br, name)
                off, cp = (OP_PAF, 40) if br == 1 else (OP_HM, 20)
                P.conv(cur, xout, sd[key + '.weight'], sd[key + '.bias'], act=ACT_RELU if relu else ACT_NONE,
                       in_ch_off=(br - 1) * cin, out_ch_off=off, cout_p=cp)
        P.tap('stage%d_paf' % st, xout, OP_PAF, 38)
        P.tap('stage%d_hm' % st, xout, OP_HM, 19)
    P.outputs = [X0]
    P.tap('pafs', X0, OP_PAF, 38)
    P.tap('heatmaps', X0, OP_HM, 19)
    return P


# ---- ArcFace -------------------------------------------------------------------------------
def pack_arcface(sd, precision='f32'):
    """arcface/model.py:4-97.  The residual stream R is the only tensor between units (halo 1).  A unit opens with a
    BatchNorm in front of a zero-padded 3x3 conv (model.py:12-14): folding it into that conv is exact away from the border
    only -- padded taps are true zeros, not BN(0) -- so the conv gets NINE biases, one per border class of the output pixel
    (Program.conv(in_affine=...)): the BatchNorm'd copy of R the unit-closing conv used to write as a second output
    (TERRAN_AMD_ARCFACE_SECOND_OUTPUT=1 keeps that program) is gone, and with it one of the two store streams of every
    unit-closing epilogue."""
    eps = arch.ARC_BN_EPS
    P = Program(MODEL_ARCFACE, precision)
    tin = P.tensor(4, 1, name='input')
    P.input_tensor = tin
    P.input_stats = (np.array([-0.1, -0.1, -0.1, 0.0]), np.array([0.25, 0.25, 0.25, 0.0]))      # (BGR - 127.5) / 128 of face crops
    units = list(arch.arcface_units())
    second = bool(os.environ.get('TERRAN_AMD_ARCFACE_SECOND_OUTPUT'))

    def next_bn(i):
        if i < len(units):
            st, u = units[i][0], units[i][1]
            return _bn_affine(sd, 'stages.%d.%d.body.0' % (st, u), eps)
        return _bn_affine(sd, 'final_layer.0', eps)

    s, sh = _bn_affine(sd, 'initial_layer.1', eps)
    W, b = _fold(sd['initial_layer.0.weight'], None, s, sh)
    R = P.tensor(64, 0 if second else 1, name='stem')
    Z = None
    if second:
        s2, sh2 = next_bn(0)
        Z = P.tensor(64, 1)
        P.conv(tin, R, W, b, act=ACT_PRELU, prelu=sd['initial_layer.2.weight'], out2=Z, scale2=s2, shift2=sh2)
    else:
        P.conv(tin, R, W, b, act=ACT_PRELU, prelu=sd['initial_layer.2.weight'])
    for i, (st, u, cin, cout, stride, sc) in enumerate(units):
        p = 'stages.%d.%d' % (st, u)
        s, sh = _bn_affine(sd, p + '.body.2', eps)
        W1, b1 = _fold(sd[p + '.body.1.weight'], None, s, sh)
        Y = P.tensor(cout, 1)
        # stage 4 (7 x 7 maps, 512 channels): 100 output tiles at 64 crops on 256 CUs, 144 K slabs -> K in two fixed halves
        ks = 2 if (cout == 512 and not os.environ.get('TERRAN_AMD_NO_STAGE4_KSPLIT')) else 0
        if second:
            P.conv(Z, Y, W1, b1, act=ACT_PRELU, prelu=sd[p + '.body.3.weight'], k_split=ks if cin == 512 else 0)
        else:
            P.conv(R, Y, W1, b1, act=ACT_PRELU, prelu=sd[p + '.body.3.weight'], k_split=ks if cin == 512 else 0,
                   in_affine=next_bn(i))
        if sc:
            s, sh = _bn_affine(sd, p + '.shortcut.1', eps)
            Ws, bs = _fold(sd[p + '.shortcut.0.weight'], None, s, sh)
            S = P.tensor(cout, 0)
            P.conv(R, S, Ws, bs, stride=stride, pad=0)
            res = S
        else:
            res = R
        s, sh = _bn_affine(sd, p + '.body.5', eps)
        W2, b2 = _fold(sd[p + '.body.4.weight'], None, s, sh)
        last = i == len(units) - 1
        Rn = P.tensor(cout, 0 if (second or last) else 1)
        if second:
            Zn = P.tensor(cout, 0 if last else 1)
            s2, sh2 = next_bn(i + 1)
            P.conv(Y, Rn, W2, b2, stride=stride, res=res, out2=Zn, scale2=s2, shift2=sh2, k_split=ks)
            Z = Zn
        else:
            P.conv(Y, Rn, W2, b2, stride=stride, res=res, k_split=ks)
        if u == arch.ARC_UNITS[st] - 1:
            P.tap('stage%d' % (st + 1), Rn, 0, cout)
        R = Rn
    # head: BN2d -> Flatten(C,H,W) -> Linear -> BN1d, as a 1x1 conv over the (N,1,1,25088) NHWC view
    s, sh = _bn_affine(sd, 'final_layer.4', eps)
    Wl = np.asarray(sd['final_layer.3.weight'], np.float64) * s[:, None]
    bl = np.asarray(sd['final_layer.3.bias'], np.float64) * s + sh
    f = np.arange(7 * 7 * 512)
    ch_pos = (f % 49) * 512 + f // 49
    if second:
        A = P.tensor(7 * 7 * 512, 0, alias_of=Z)
    else:
        # no padding between final_layer.0 and the Linear: that BatchNorm folds into the Linear exactly
        s0, t0 = next_bn(len(units))
        chan = f // 49                                          # flatten order (C, H, W): feature f belongs to channel f // 49
        bl = bl + Wl @ t0[chan]
        Wl = Wl * s0[chan][None, :]
        A = P.tensor(7 * 7 * 512, 0, alias_of=R)
    E = P.tensor(512, 0, name='embedding', f32=True)
    # f16 mode: 392 slabs of 64 channels -- below the library's own K-split rule (>= 512 slabs), same 32 ranges asked for here
    P.conv(A, E, Wl.reshape(512, 7 * 7 * 512, 1, 1), bl, ch_pos=ch_pos, pad=0, k_split=32 if P.prec == 4 else 0)
    P.outputs = [E]
    return P


# ---- RetinaFace ----------------------------------------------------------------------------
# Context tensor channel layout (96): ctx3x3 0..31 | reducer 32..47 | ctx5x5 48..63 | 7x7-mid 64..79 |
# ctx7x7 80..95; the merged head conv reads all 96 with zero weights on reducer / 7x7-mid.
def pack_retinaface(sd, precision='f32', fused=None):
    """retinaface/model.py:53-316.  Sibling convs that share an input are merged (ctx3x3+reducer,
    ctx5x5+ctx7x7.0, cls+bbox+landmark heads); the FPN nearest-x2 upsample + add is the
    residual of the lateral 1x1 conv's epilogue.  fused (default in the parity modes): the MobileNet base runs as
    one front kernel (frames -> conv3x3 s2 -> dw3x3 -> 1x1) followed by 12 [depthwise 3x3 -> 1x1] blocks whose depthwise
    output never leaves the CU (27 launches and a float copy of the frames become 13 launches);
    fused=False keeps the layer-by-layer program (the `stem` debug tap; the `bf16` throughput mode)."""
    # The detector's outputs are decisions (score >= 0.5, IoU > 0.4, descending-score ORDER among ~10^2 near-equal
    # scores per image): measured over 208 frames, bf16x3 convs (2^-16 per product) kept every detection but swapped the
    # order of near-tied scores in 3 % of the images.  The graph is HBM-bound (35 FLOP/B), so the exact-f32 MFMA costs
    # next to nothing here: in the `bf16x3` mode the detector runs on it, and its results ARE the `f32` mode's, bit for
    # bit.  (`bf16`, the throughput mode outside the parity bar, stays bf16.)  All activations are float32.
    if precision in ('f16', 'f16x2'):   # the embedder's tolerance modes are for networks without discrete decisions: the detector keeps 22 bits
        precision = 'f16x3'
    det_prec = 'f32' if precision in ('bf16x3', 'f16x3') else precision
    P = Program(MODEL_RETINAFACE, det_prec)
    # f16x3: the REFINER (FPN laterals, 3x3 aggregations, context modules, heads: 18 dense convs, 0.75 of the detector's
    # 1.5 ms at C2 and f32-MFMA-bound) runs on the split-half MFMA -- 22-bit operands, measured 0 decision flips / order
    # swaps against the oracle over 224 frames, the same as exact f32 (bf16x3's 16 bits swapped near-tied scores in 3 % of
    # the images: that mode keeps the whole detector exact f32).  The MobileNet base stays exact f32: its input is raw
    # 0..255 pixels and it is bound by its depthwise taps, not by the matrix pipe.
    refiner_prec = 'f16x3' if (precision == 'f16x3' and not os.environ.get('TERRAN_AMD_DETECTOR_F32')) else None
    P.allow_split = refiner_prec is not None
    if fused is None:
        fused = P.prec == 0 and not os.environ.get('TERRAN_AMD_NO_FUSED_DETECTOR')       # A/B switch
    tin = P.tensor(4, 1, alias_of=-2 if fused else -1, name='input')
    P.input_tensor = tin
    P.input_stats = (np.array([110.0, 110.0, 110.0, 0.0]), np.array([4900.0, 4900.0, 4900.0, 0.0]))   # raw 0..255 BGR pixels
    eps = arch.RETINA_BASE_BN_EPS

    def cbr(key_conv, key_bn, e=eps, bias=False):
        s, sh = _bn_affine(sd, key_bn, e)
        return _fold(sd[key_conv + '.weight'], sd[key_conv + '.bias'] if bias else None, s, sh)

    # the base network as a chain: stem conv, then alternating depthwise / pointwise layers
    pw_keys = [('base.scales.%d.%d.conv_block.0' % (si, bi), 'base.scales.%d.%d.conv_block.1' % (si, bi), cout, both)
               for si, scale in enumerate(arch.RETINA_SCALES) for bi, (cin, cout, stride, both) in enumerate(scale)]
    pw_keys += [('base.final_conv.0.conv_block.0', 'base.final_conv.0.conv_block.1', 256, False),
                ('base.final_conv.1', 'base.final_conv.2', 256, True)]
    dw_keys = [('base.first_conv_block.3', 'base.first_conv_block.4', 1)]
    dw_keys += [('base.scales.%d.%d.sep_block.0' % (si, bi), 'base.scales.%d.%d.sep_block.1' % (si, bi), stride)
                for si, scale in enumerate(arch.RETINA_SCALES) for bi, (cin, cout, stride, both) in enumerate(scale)]
    dw_keys += [('base.final_conv.0.sep_block.0', 'base.final_conv.0.sep_block.1', 1)]
    assert len(pw_keys) == len(dw_keys) == 13
    feats = []
    if fused:
        Ws, bs = cbr('base.first_conv_block.0', 'base.first_conv_block.1')
        t = None
        # the front kernel also runs the next block (depthwise stride 2 -> 1x1 16 -> 32): the 16-channel half-resolution map stays
        # on the CU (TERRAN_AMD_NO_FUSED_FRONT: the two as separate launches, A/B)
        fuse_front = not os.environ.get('TERRAN_AMD_NO_FUSED_FRONT')
        front = None
        for i, ((dk, dbn, stride), (pk, pbn, cout, both)) in enumerate(zip(dw_keys, pw_keys)):
            Wd, bd = cbr(dk, dbn)
            Wp, bp = cbr(pk, pbn)
            if i == 0 and fuse_front:
                front = (Wd, bd, Wp, bp)
                continue
            c = P.tensor(cout, 1 if i < 12 else 0)
            if i == 0:
                P.rfstem(tin, c, Ws, bs, Wd, bd, Wp, bp)
            elif i == 1 and front is not None:
                assert stride == 2 and cout == 32 and Wd.shape[0] == 16 and not both
                P.rfstem(tin, c, Ws, bs, *front, Wd, bd, Wp, bp)
            else:
                # the 1x1 of a [depthwise -> pointwise] block follows the refiner's mode from the stride-8 maps on (cin >= 64):
                # the two blocks on the 104 x 185 maps are bound by their depthwise taps and stay exact f32
                P.dwpw(t, c, Wd, bd, Wp, bp, stride=stride,
                       precision=refiner_prec if (Wd.shape[0] >= 64 and not os.environ.get('TERRAN_AMD_DETECTOR_BASE_F32')) else None)
            if both:
                feats.append(c)
            t = c
    else:
        W, b = cbr('base.first_conv_block.0', 'base.first_conv_block.1')
        a0 = P.tensor(8, 1)
        P.conv(tin, a0, W, b, stride=2, act=ACT_RELU)
        t = a0
        for i, ((dk, dbn, stride), (pk, pbn, cout, both)) in enumerate(zip(dw_keys, pw_keys)):
            W, b = cbr(dk, dbn)
            d = P.tensor(P.tensors[t][0], 0, name='stem' if i == 0 else None)
            P.dwconv(t, d, W, b, stride=stride)
            W, b = cbr(pk, pbn)
            c = P.tensor(cout, 1 if i < 12 else 0)
            P.conv(d, c, W, b, act=ACT_RELU)
            if both:
                feats.append(c)
            t = c
    f8, f16, f32 = feats
    P.tap('feat8', f8, 0, 64)
    P.tap('feat16', f16, 0, 128)
    P.tap('feat32', f32, 0, 256)

    e2 = arch.RETINA_REFINER_BN_EPS
    rp = refiner_prec

    def rcbr(p):
        return cbr(p + '.0', p + '.1', e2, bias=True)

    A = arch.RETINA_NUM_ANCHORS
    heads = {}
    # The context module + heads of a pyramid level depend on that level's map only.  The stride-32 and stride-16 levels are
    # 4 launches of 9-22 us each (a few dozen tiles: launch latency, not work): they go to side streams ("lanes") right
    # behind the op that finishes their map and run beside the rest of the refiner instead of in front of it.
    use_lanes = not os.environ.get('TERRAN_AMD_NO_DETECTOR_LANES')

    def context_and_heads(s, x, lane):
        P.lane = lane if use_lanes else 0
        p = 'refiner.context_stride%d' % s
        ctx = P.tensor(96, 1)
        W3, b3 = rcbr(p + '.context_3x3')
        Wr, br_ = rcbr(p + '.dimension_reducer')
        P.conv(x, ctx, np.concatenate([W3, Wr]), np.concatenate([b3, br_]), act=ACT_RELU, out_ch_off=0, precision=rp)
        W5, b5 = rcbr(p + '.context_5x5')
        W7, b7 = cbr(p + '.context_7x7.0', p + '.context_7x7.1', e2, bias=True)
        P.conv(ctx, ctx, np.concatenate([W5, W7]), np.concatenate([b5, b7]), act=ACT_RELU, in_ch_off=32,
               out_ch_off=48, precision=rp)
        W7b, b7b = cbr(p + '.context_7x7.3', p + '.context_7x7.4', e2, bias=True)
        P.conv(ctx, ctx, W7b, b7b, act=ACT_RELU, in_ch_off=64, out_ch_off=80, precision=rp)
        P.tap('ctx%d_3x3' % s, ctx, 0, 32)
        P.tap('ctx%d_5x5' % s, ctx, 48, 16)
        P.tap('ctx%d_7x7' % s, ctx, 80, 16)
        # merged heads: rows [cls 2A | bbox 4A | landmark 10A]; input order cat[3x3, 5x5, 7x7]
        Wh = np.concatenate([sd['outputs.%s_stride%d.weight' % (h, s)] for h in ('cls', 'bbox', 'landmark')])
        bh = np.concatenate([sd['outputs.%s_stride%d.bias' % (h, s)] for h in ('cls', 'bbox', 'landmark')])
        pos = np.concatenate([np.arange(32), 48 + np.arange(16), 80 + np.arange(16)])
        hd = P.tensor(16 * A, 0, name='head%d' % s, f32=True)
        P.conv(ctx, hd, Wh, bh, ch_pos=pos, cin_p=96, precision=rp)
        heads[s] = hd
        P.lane = 0

    W, b = rcbr('refiner.conv_stride32')
    p32 = P.tensor(64, 1, name='p32')
    P.conv(f32, p32, W, b, act=ACT_RELU, precision=rp)
    context_and_heads(32, p32, 1)
    W, b = rcbr('refiner.conv_stride16')
    s16 = P.tensor(64, 1)
    P.conv(f16, s16, W, b, act=ACT_RELU, res=p32, res_up2=1, precision=rp)
    W, b = rcbr('refiner.aggr_stride16')
    p16 = P.tensor(64, 1, name='p16')
    P.conv(s16, p16, W, b, act=ACT_RELU, precision=rp)
    context_and_heads(16, p16, 2)
    W, b = rcbr('refiner.conv_stride8')
    s8 = P.tensor(64, 1)
    P.conv(f8, s8, W, b, act=ACT_RELU, res=p16, res_up2=1, precision=rp)
    W, b = rcbr('refiner.aggr_stride8')
    p8 = P.tensor(64, 1, name='p8')
    P.conv(s8, p8, W, b, act=ACT_RELU, precision=rp)
    context_and_heads(8, p8, 0)
    P.outputs = [heads[32], heads[16], heads[8]]
    return P


PACKERS = {MODEL_RETINAFACE: pack_retinaface, MODEL_ARCFACE: pack_arcface, MODEL_OPENPOSE: pack_openpose}
