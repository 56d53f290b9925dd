"""A minimal single-threaded stand-in for the slice of the GNU Radio runtime the hot-path blocks
rely on, so that flowgraphs in the style of the reference's QA tests
(`vector_source -> block under test -> vector_sink; tb.run()`, python/qa_MUSIC_lin_array.py:73-91)
can run where GNU Radio 3.7 is not installed.

Semantics kept (they are observable in the outputs):
  * `gr::block` history: a block with `history() = h` sees `h-1` samples before the first new
    one, zero-filled at stream start (lib/autocorrelate_impl.cc:57);
  * `forecast` / `consume_each` of a `gr::block` with a non-1:1 rate (:75-80,114);
  * `sync_block` 1:1 item flow; multi-output blocks (find_local_max has two ports);
  * scheduler-sized calls: `work` is invoked repeatedly with at most `max_noutput_items`.
Nothing else of GNU Radio (threads, tags, message ports) is modelled.
"""
from __future__ import annotations

import numpy as np


class vector_source:
    def __init__(self, data, dtype, vlen=1):
        self.data = np.ascontiguousarray(np.asarray(data, dtype=dtype).reshape(-1, vlen) if vlen > 1
                                         else np.asarray(data, dtype=dtype).reshape(-1))
        self.vlen = vlen


def vector_source_c(data, repeat=False, vlen=1):
    return vector_source(data, np.complex64, vlen)


def vector_source_f(data, repeat=False, vlen=1, tags=None):
    return vector_source(data, np.float32, vlen)


class vector_sink:
    def __init__(self, dtype, vlen=1):
        self.dtype, self.vlen = dtype, vlen
        self._chunks = []

    def _push(self, arr):
        self._chunks.append(np.array(arr, copy=True))

    def data(self):
        if not self._chunks:
            return np.zeros((0,), dtype=self.dtype)
        return np.concatenate([c.reshape(-1) for c in self._chunks])


def vector_sink_c(vlen=1):
    return vector_sink(np.complex64, vlen)


def vector_sink_f(vlen=1):
    return vector_sink(np.float32, vlen)


class null_sink(vector_sink):
    def __init__(self, itemsize=0):
        super().__init__(np.float32, 1)

    def _push(self, arr):
        pass


class top_block:
    """connect((src, port), (dst, port)) edges, then run(): evaluates the graph block by block in
    topological order (every block in this module's scope is rate-deterministic, so running each
    block to completion over its whole input is equivalent to GNU Radio's interleaved schedule)."""

    def __init__(self, max_noutput_items=8):
        self.edges = []
        self.max_noutput_items = max_noutput_items

    def connect(self, src, dst):
        self.edges.append((src, dst))

    def run(self):
        outputs = {}          # (block id, port) -> ndarray of items
        blocks = {}
        for (s, _sp), (d, _dp) in self.edges:
            blocks[id(s)] = s
            blocks[id(d)] = d
        for b in blocks.values():
            if isinstance(b, vector_source):
                outputs[(id(b), 0)] = b.data
        pending = [b for b in blocks.values() if not isinstance(b, vector_source)]
        progress = True
        while pending and progress:
            progress = False
            for b in list(pending):
                ins = sorted([(dp, (id(s), sp)) for (s, sp), (d, dp) in self.edges if d is b])
                if not all(key in outputs for _, key in ins):
                    continue
                in_arrays = [outputs[key] for _, key in ins]
                if isinstance(b, vector_sink):
                    b._push(in_arrays[0])
                else:
                    outs = self._run_block(b, in_arrays)
                    for port, arr in enumerate(outs):
                        outputs[(id(b), port)] = arr
                pending.remove(b)
                progress = True
        if pending:
            raise RuntimeError("flowgraph has unconnected inputs")

    def _run_block(self, b, in_arrays):
        mno = self.max_noutput_items
        if hasattr(b, "general_work"):        # gr::block with history + forecast
            hist = b.history() - 1
            in_arrays = [a.reshape(-1) for a in in_arrays]           # stream ports carry scalar items
            streams = [np.concatenate([np.zeros(hist, dtype=a.dtype), a]) for a in in_arrays]
            n_new = min(a.shape[0] for a in in_arrays)
            chunks, consumed = [[] for _ in b.out_sig], 0
            while True:
                avail_new = n_new - consumed
                n = min(mno, avail_new // b.forecast(1))
                if n <= 0:
                    break
                outs = [np.empty((n, vl), dtype=dt) for dt, vl in b.out_sig]
                produced, used = b.general_work(n, [s[consumed:] for s in streams], outs)
                for c, o in zip(chunks, outs):
                    c.append(o[:produced])
                consumed += used                      # consume_each
            return [np.concatenate(c) if c else np.zeros((0, vl), dtype=dt) for c, (dt, vl) in zip(chunks, b.out_sig)]
        # gr::sync_block (one or several input ports, all at the same rate)
        ins = []
        for a, (dt, vl) in zip(in_arrays, b.in_sig):
            ins.append(a.reshape(-1, vl) if vl > 1 else a.reshape(-1, 1))
        total = min(i.shape[0] for i in ins)
        outs = [[] for _ in b.out_sig]
        pos = 0
        while pos < total:
            n = min(mno, total - pos)
            bufs = [np.empty((n, vl), dtype=dt) for dt, vl in b.out_sig]
            produced = b.work(n, [np.ascontiguousarray(i[pos:pos + n]) for i in ins], bufs)
            for o, buf in zip(outs, bufs):
                o.append(buf[:produced])
            pos += produced
        return [np.concatenate(o) if o else np.zeros((0, vl), dtype=dt) for o, (dt, vl) in zip(outs, b.out_sig)]
