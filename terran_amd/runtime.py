"""Process-wide device contexts and weight resolution for the wrapper classes."""
import collections
import os
import threading

import numpy as np

from . import lib, weights

_contexts = {}
_pack_memo = collections.OrderedDict()
_memo_lock = threading.Lock()


def clear_pack_memo():
    """Drop the packed programs kept for dict states (they pin the state dicts and a few hundred MB of blobs)."""
    with _memo_lock:
        _pack_memo.clear()


def device_index(device):
    """Accepts None, an int, 'cuda', 'cuda:1', 'hip:1' or a torch.device."""
    if device is None:
        return int(os.environ.get('LOCAL_RANK', 0)) if os.environ.get('TERRAN_AMD_USE_LOCAL_RANK') else 0
    if isinstance(device, (int, np.integer)):
        return int(device)
    s = str(device)
    if s in ('cuda', 'hip', 'gpu'):
        return 0
    if ':' in s:
        kind, idx = s.split(':', 1)
        if kind in ('cuda', 'hip', 'gpu'):
            return int(idx)
    raise ValueError('terran_amd runs on MI355X only; cannot place a model on device %r' % (device,))


def get_context(device=None):
    idx = device_index(device)
    ctx = _contexts.get(idx)
    if ctx is None or ctx.h is None:
        ctx = lib.Context(idx)
        _contexts[idx] = ctx
    return ctx


def new_context(device=None):
    """An additional context (own HIP stream + scratch) on the same GPU, for a second host thread."""
    return lib.Context(device_index(device))


def resolve_state(kind, state):
    """state: None (look up the Terran checkpoint file), a path, or a {key: array} dict."""
    if isinstance(state, dict):
        return state
    if isinstance(state, (str, os.PathLike)):
        return weights.load_state(state)
    from . import checkpoint
    path = checkpoint.find_checkpoint_file(kind)
    if path is not None:
        return weights.load_state(path)
    if os.environ.get('TERRAN_AMD_SYNTHETIC_WEIGHTS'):
        return getattr(weights, 'make_%s_state' % kind)()
    raise ValueError('Checkpoint not found.')     # same error as terran/checkpoint.py:242,310


DEFAULT_PRECISION = 'f16x3'


def resolve_precision(precision=None):
    """A precision names the arithmetic of all three networks:
      'f32'    every conv on the exact-f32 MFMA (the like-for-like arithmetic);
      'f16x3'  THE DEFAULT ($TERRAN_AMD_PRECISION overrides): split-half MFMA everywhere: operands x = hi + lo (two IEEE halfs, 22
               bits), three MFMAs per product, float32-grade results (embeddings to ~1e-6 of the oracle, decisions at float32's own
               rate; the detector's raw-pixel front stays exact f32, its deep base and refiner run split-half); half-float range:
               TA_E_RANGE -> the wrappers re-run the batch on an exact-f32 twin (RangeFallback below);
      'f16x2'  opt-in tolerance mode for the EMBEDDER: the detector and the pose network -- everything that takes a discrete
               decision -- exactly as in 'f16x3' (the same packed programs, bit for bit); the embedder, whose only bar is 1e-3 on
               the unit-norm embedding, on two of the three products, (w_hi + w_lo) * x_hi: weights and the shortcut trunk keep 22
               bits, an activation enters a contraction as its hi half.  Measured 1.8e-4 (seeded weights) / 8.2e-4 (wild-statistics
               weights) worst component against the oracle (tests/probe_embedder_modes.py) -- which is why the mode is GUARDED:
               arcface.guard_f16x2 embeds 32 fixed calibration crops in both modes when the model is loaded and keeps 'f16x3'
               (one warning) when they differ by more than 5e-4; the decision is cached with the repack cache;
      'f16'    opt-in: the embedder on ONE MFMA per product and 2-byte activations (3.3e-4 / 1.8e-3: outside the bar on the wild weights);
      'bf16x3' split-bf16 (16-bit operands, float32 range); 'bf16' throughput mode outside the 1e-3 parity bar."""
    p = precision or os.environ.get('TERRAN_AMD_PRECISION', DEFAULT_PRECISION)
    if p not in ('f32', 'f16x3', 'bf16x3', 'bf16', 'f16', 'f16x2'):
        raise ValueError('unknown precision %r' % (p,))
    return p


class RangeFallback:
    """The `f16x3` mode keeps activations as two half floats: |x| > 65504 does not fit, and the library then fails the
    call with TA_E_RANGE instead of returning numbers.  Wrappers route such a batch to a second model packed for the
    exact-f32 MFMA (same weights, float32 range), built on first use; `fallbacks` counts how often that happened."""

    def _init_fallback(self, kind, state):
        self._fb_kind, self._fb_state, self._fb_model, self.fallbacks, self._fb_calls = kind, state, None, 0, 0

    def _with_fallback(self, fn):
        """fn(model) -> result; re-run on the f32 model when the f16x3 one reports TA_E_RANGE.  Weights whose activations
        leave the half-float range on (nearly) every batch would pay for two runs per call: once the first three calls have
        all fallen back, the exact-f32 model simply takes over."""
        self._fb_calls += 1
        if not (self.fallbacks >= 3 and self.fallbacks == self._fb_calls - 1):
            try:
                return fn(self.model)
            except lib.TerranAmdError as e:
                if e.code != lib.E_RANGE:
                    raise
        if self._fb_model is None:
            import warnings
            warnings.warn('terran_amd: %s activations left the half-float range of the split-half mode (TA_E_RANGE); this batch '
                          'is re-run on an exact-f32 copy of the model (about 3x slower; after three such batches in a row it '
                          'takes over).  precision="f32" avoids the first attempt; tools/amax_debug.py shows the layer.'
                          % self._fb_kind, RuntimeWarning, stacklevel=3)
            self._fb_model = lib.Model(self.ctx, packed_program(self._fb_kind, self._fb_state, 'f32'))
        self.fallbacks += 1
        return fn(self._fb_model)


def packed_program(kind, state, precision):
    """Pack `state` for `kind`, going through the repack cache when the weights come from a checkpoint file.

    Cache key = checkpoint id + precision + size/mtime of the .pth (terran/checkpoint.py:118-150 layout) + blob version + a
    hash of the packer's sources: `$TERRAN_HOME/checkpoints/<id>.<precision>.<size>.<mtime>.v<blob version>.<pack hash>.tam`.
    With any pack-time switch set (pack.PACK_SWITCHES: A/B variants of the programs) the disk cache is bypassed in both
    directions.  Dict states (tests, bench) are packed directly and memoised per (dict object, precision, switches): a state
    dict must not be mutated in place between two models built from it (copy it: `dict(sd)`)."""
    from . import checkpoint, pack
    packer = getattr(pack, 'pack_%s' % kind)
    if kind != 'arcface' and precision in ('f16', 'f16x2'):     # the embedder's tolerance modes: detector and pose ARE the f16x3 programs
        precision = 'f16x3'                                     # (one pack, one memo entry, one cache file for all three names)
    path = None
    if state is None:
        path = checkpoint.find_checkpoint_file(kind)
    elif isinstance(state, (str, os.PathLike)):
        path = state
    switches = pack.active_switches()
    if path is None or os.environ.get('TERRAN_AMD_NO_PACK_CACHE') or switches or pack.source_tag() is None:
        # dict states (tests, bench, StreamPipeline's lanes): many models of one process are built from the SAME dict --
        # pack it once (a pack is seconds of numpy work; 16 lanes x 3 networks would spend a minute on it)
        sd = resolve_state(kind, state)
        key = (kind, id(sd), precision, switches)
        with _memo_lock:
            hit = _pack_memo.get(key)
            if hit is not None and hit[0] is sd:
                _pack_memo.move_to_end(key)
                return hit[1]
        prog = packer(sd, precision)
        prog._blob = prog.blob()                    # built once, shared by every model loaded from it
        with _memo_lock:
            _pack_memo[key] = (sd, prog)            # holds `sd`: its id cannot be recycled while the entry lives
            while len(_pack_memo) > 6:
                _pack_memo.popitem(last=False)
        return prog
    st = os.stat(path)
    cache = '%s.%s.%d.%d.v%d.%s.tam' % (os.path.splitext(str(path))[0], precision, st.st_size, int(st.st_mtime),
                                        pack.BLOB_VERSION, pack.source_tag())
    if os.path.exists(cache):
        try:
            prog = pack.Program.from_cache(cache)
            prog._cache_path = cache
            return prog
        except Exception:
            pass                                    # unreadable cache: repack below
    prog = packer(weights.load_state(path), precision)
    prog._cache_path = cache
    try:
        prog.save_cache(cache)
        # repack caches of this checkpoint and precision written for ANOTHER STATE OF THE FILE (size / mtime differ): orphans
        # now.  Caches of the same file state under another blob version / pack hash belong to another install sharing
        # $TERRAN_HOME and stay (two versions would otherwise delete each other's caches on every cold start).
        import glob
        stem = os.path.splitext(str(path))[0]
        mine = '%d.%d.' % (st.st_size, int(st.st_mtime))
        for old in glob.glob('%s.%s.*.tam' % (glob.escape(stem), precision)):
            if old != cache and not old[len(stem) + len(precision) + 2:].startswith(mine):
                try:
                    os.unlink(old)
                except OSError:
                    pass
    except OSError:
        pass                                        # read-only checkpoint dir: run uncached
    return prog
