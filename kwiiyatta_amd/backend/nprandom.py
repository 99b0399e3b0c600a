"""numpy's legacy generator continued on the GPU (kwy_nprandom.hip): the `np.random.normal` behind the reference's
silence padding (/root/reference/kwiiyatta/vocoder/world.py:158-161), draw for draw, without the 4.6 ms of host
time per aligned pair.

    rs = DeviceRandomState.from_global()          # picks up np.random's state (np.random.seed(...) before)
    pads = rs.abs_normal(EPS / fs, (100, 1025))   # device tensor, == np.abs(np.random.normal(0, EPS / fs, (100, 1025)))
    rs.to_global()                                # np.random continues where the device stopped

The state stays in device memory between calls; calls are enqueued on the context's stream in order, which IS the
order of the draws.
"""
import numpy as np
import torch

from .. import _lib
from .._lib import lib, c_vp

_N = 624
_BYTES = int(lib.kwy_np_state_bytes())


def _pack(state):
    """numpy's legacy state tuple -> the library's struct as a uint8 array"""
    name, key, pos = state[0], state[1], state[2]
    has_gauss = state[3] if len(state) > 3 else 0
    cached = state[4] if len(state) > 4 else 0.0
    if name != 'MT19937':
        raise ValueError('only the MT19937 (legacy RandomState) state can be continued')
    buf = np.zeros(_BYTES, dtype=np.uint8)
    buf[:4 * _N] = np.ascontiguousarray(key, dtype=np.uint32).view(np.uint8)
    buf[4 * _N:4 * _N + 8] = np.array([pos, has_gauss], dtype=np.int32).view(np.uint8)
    buf[4 * _N + 8:4 * _N + 16] = np.array([cached], dtype=np.float64).view(np.uint8)
    return buf


def _unpack(buf):
    key = buf[:4 * _N].view(np.uint32).copy()
    pos, has_gauss = (int(v) for v in buf[4 * _N:4 * _N + 8].view(np.int32))
    cached = float(buf[4 * _N + 8:4 * _N + 16].view(np.float64)[0])
    return ('MT19937', key, pos, has_gauss, cached)


class DeviceRandomState:
    def __init__(self, state, device_index=0, ctx=None, stream=None):
        self.dev = torch.device('cuda', device_index)
        if ctx is None:
            self.stream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
            ctx = _lib.Context(device_index, stream=self.stream.cuda_stream)
        else:
            self.stream = torch.cuda.ExternalStream(lib.kwy_ctx_stream(ctx.handle), device=self.dev)
        self.ctx = ctx
        self.state = torch.empty(_BYTES, dtype=torch.uint8, device=self.dev)
        self._foreign = False      # a draw was enqueued on another context's stream (or inside a graph that replays it)
        self.set_state(state)

    def _join_foreign(self):
        """Draws enqueued through another context (`abs_normal_blocks(ctx=...)`: a driver's stream, possibly captured
        into a HIP graph that is replayed on yet another stream) read and write `self.state` outside the generator's
        own stream.  Before the generator's stream touches the state again the device is drained: the only order
        that holds for every such stream, replayed graphs included."""
        if self._foreign:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('DeviceRandomState: the state was last used on another context\'s stream; it cannot '
                                   'be joined from inside a stream capture')
            torch.cuda.synchronize(self.dev)
            self._foreign = False

    @classmethod
    def from_global(cls, **kwargs):
        return cls(np.random.get_state(), **kwargs)

    @classmethod
    def from_seed(cls, seed, **kwargs):
        return cls(np.random.RandomState(seed).get_state(), **kwargs)

    def set_state(self, state):
        self._join_foreign()
        with torch.cuda.stream(self.stream):
            self.state.copy_(torch.from_numpy(_pack(state)))
        self.stream.synchronize()

    def get_state(self):
        """numpy's state tuple after everything enqueued so far -- on the generator's own stream and, when draws went
        through other contexts (`abs_normal_blocks(ctx=...)`), on the whole device (synchronises)"""
        self._join_foreign()
        with torch.cuda.stream(self.stream):
            host = self.state.cpu()
        self.stream.synchronize()
        state = _unpack(host.numpy())
        if state[2] < 0:
            # k_np_advance poisons the position when a request ran out of laid-out attempts (a ~14 sigma event of
            # the acceptance rate): some outputs of that call were never written
            raise RuntimeError('DeviceRandomState: a draw ran out of attempts; its outputs are incomplete and the '
                               'stream no longer follows numpy\'s')
        return state

    def to_global(self):
        np.random.set_state(self.get_state())

    def normal(self, loc, scale, size=None, out=None, absolute=False):
        """np.random.normal(loc, scale, size) as a float64 device tensor (into `out` if given), asynchronously on the
        generator's stream; absolute=True: np.abs of it"""
        if out is None:
            out = torch.empty(size, dtype=torch.float64, device=self.dev)
        elif out.dtype != torch.float64 or not out.is_contiguous() or out.device != self.dev:
            raise ValueError('out must be a contiguous float64 tensor on the generator\'s device')
        self._join_foreign()
        _lib.check(self.ctx, lib.kwy_np_normal_dev(self.ctx.handle, c_vp(self.state.data_ptr()), float(loc), float(scale),
                                                   int(bool(absolute)), out.numel(), c_vp(out.data_ptr())))
        return out

    def abs_normal(self, scale, size=None, out=None):
        return self.normal(0.0, scale, size=size, out=out, absolute=True)

    def abs_normal_blocks(self, scale, outs, ctx=None):
        """np.abs(np.random.normal(0, scale, shape)) for equally sized tensors in a row, one library call (the four pad
        blocks of an aligned pair: source head, source tail, target head, target tail -- or those of all pairs of a
        batch in pair order: one layout of the generator's words for all of them).
        ctx: enqueue on this library context's stream instead of the generator's own (a driver that wants the draw in
        line with its other work; it then also orders the draw against other users of the generator)."""
        if ctx is None or ctx is self.ctx:
            ctx = self.ctx
            self._join_foreign()
        else:
            self._foreign = True
        n_each = outs[0].numel()
        for t in outs:
            if t.dtype != torch.float64 or not t.is_contiguous() or t.device != self.dev or t.numel() != n_each:
                raise ValueError('outs must be equally sized contiguous float64 tensors on the generator\'s device')
        ptrs = (c_vp * len(outs))(*[t.data_ptr() for t in outs])
        _lib.check(ctx, lib.kwy_np_normal_blocks_dev(ctx.handle, c_vp(self.state.data_ptr()), 0.0, float(scale),
                                                     1, len(outs), n_each, ptrs))
        return outs

    def sync(self):
        """wait for everything enqueued on the generator's stream (results are written asynchronously on it)"""
        self.stream.synchronize()

    def record_event(self):
        self._join_foreign()
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev
