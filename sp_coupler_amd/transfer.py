"""Transfer plumbing of the drop-in path: groups of named arrays that cross PCIe together (``Arena``: ONE pinned host
buffer mirrored by ONE device buffer; ``ShardedArena``: the same host buffer mirrored row-block-wise on several GPUs,
sp_coupler_amd.multi) and the optional per-step HIP-event trace of every copy and launch (``StepTrace``).
PyTorch is plumbing here: pinned memory, device memory, streams, events."""
import numpy
import torch


class Arena:
    """Named arrays packed into ONE pinned host buffer and ONE device buffer, so that a whole group of inputs
    (or results) crosses PCIe in a single copy.  ``h[name]`` / ``hn[name]`` are the host views (torch / NumPy),
    ``d[name]`` the device views; every array starts 256-B aligned (the 16-B accesses of the compile-time-geometry
    kernels need aligned bases).  With a CPU "device" (the test-only oracle engine) host and device are one buffer.

    ``stream``: the COMPUTE stream the kernels reading / writing this buffer are launched on (an engine's own stream,
    ``Engine(stream=...)``); None = torch's current stream of the device at the time of each call.  Every ordering
    below (single copies, fences of the copy stream, events) is taken against that stream, never against whatever
    stream happens to be current when an engine launches elsewhere."""

    ALIGN = 256
    PIECE_MIN = 4 << 20          # arrays smaller than this travel together in one copy, larger ones each on their own

    def __init__(self, device, specs, stream=None):
        self.device = torch.device(device)
        self.stream = stream
        off, lay = 0, []
        for name, shape, dtype in specs:
            nbytes = int(numpy.prod(shape, dtype=numpy.int64)) * torch.empty((), dtype=dtype).element_size()
            lay.append((name, tuple(shape), dtype, off, nbytes))
            off += -(-nbytes // self.ALIGN) * self.ALIGN
        self.nbytes = off
        on_gpu = self.device.type == "cuda"
        self.host = torch.empty(max(off, 1), dtype=torch.uint8, pin_memory=on_gpu)
        self.dev = torch.empty(max(off, 1), dtype=torch.uint8, device=self.device) if on_gpu else self.host
        self.h, self.d, self.hn, self.begin, self.end = {}, {}, {}, {}, {}
        for name, shape, dtype, o, nb in lay:
            self.h[name] = self.host[o:o + nb].view(dtype).view(shape)
            self.d[name] = self.dev[o:o + nb].view(dtype).view(shape)
            self.hn[name] = self.h[name].numpy()
            self.begin[name], self.end[name] = o, o + nb
        self.done = None        # event of the last download (created on first use)
        self.side, self.pushed, self.landed, self.pending, self.deferred, self.deferred_what = None, False, {}, {}, None, "h2d"
        # name -> {stream handle: event recorded behind the last upload of that array ON THAT STREAM} (writable()); an event
        # object is shared by the names one copy carried, counted in _ev_refs and re-recorded only when no name refers to it
        self.sent, self._ev_refs, self._ev_free = {}, {}, []

    def _cur(self):
        """the compute stream every copy of this buffer is ordered against"""
        return self.stream if self.stream is not None else torch.cuda.current_stream(self.device)

    def _copy(self, dst, src, what, nbytes, stream, current=False):
        """one async copy issued on ``stream`` (torch issues copies on ITS current stream: make ``stream`` that, unless the
        caller knows it is the current one already -- entering the stream context costs ~10 us of host time per copy)"""
        if current:
            if trace is not None:
                with trace.region(what, nbytes, self.device):
                    dst.copy_(src, non_blocking=True)
            else:
                dst.copy_(src, non_blocking=True)
            return
        with torch.cuda.stream(stream):
            if trace is not None:
                with trace.region(what, nbytes, self.device):
                    dst.copy_(src, non_blocking=True)
            else:
                dst.copy_(src, non_blocking=True)

    def _mark_sent(self, names, stream):
        """one event behind the copy that carried ``names`` up: the host may refill them once it has fired.  A later copy of
        the same array on the SAME stream supersedes the earlier one (stream order); copies on different streams (upload()
        / fence() on the compute stream, push() on the copy stream) are both kept, and writable() waits for all of them.
        (Round-4 advisor: the event used to be keyed by the first name of the copy and shared by reference -- a later copy
        starting with the same name re-recorded the object other names were still waiting on, on another stream.)"""
        if not names:
            return
        ev = self._ev_free.pop() if self._ev_free else torch.cuda.Event()
        ev.record(stream)
        key = stream.cuda_stream
        self._ev_refs[ev] = len(names)
        for nm in names:
            old = self.sent.setdefault(nm, {}).get(key)
            if old is not None:
                self._release(old)
            self.sent[nm][key] = ev

    def _release(self, ev):
        left = self._ev_refs.get(ev, 1) - 1
        if left <= 0:
            self._ev_refs.pop(ev, None)
            self._ev_free.append(ev)          # no name refers to it any more: free to be recorded again
        else:
            self._ev_refs[ev] = left

    def writable(self, name):
        """wait until every outstanding upload of ``name`` has left the pinned host buffer (call before refilling it on the
        host).  Free in the normal call order: a step waits for its results, hence for the kernel behind the uploads."""
        evs = self.sent.pop(name, None)        # (an array collected for the copy at the next fence() is not on the wire)
        if evs:
            for ev in evs.values():
                ev.synchronize()
                self._release(ev)

    def _names_in(self, lo, hi):
        return [nm for nm in self.begin if lo <= self.begin[nm] and self.end[nm] <= hi]

    def upload(self, upto=None, what="h2d"):
        """host -> device (one async copy on the compute stream; later kernels on that stream are ordered after it)"""
        if self.dev is not self.host:
            n = self.nbytes if upto is None else self.end[upto]
            cur = self._cur()
            self._copy(self.dev[:n], self.host[:n], what, n, cur, current=self.stream is None)
            self._mark_sent(self._names_in(0, n), cur)

    def download(self, upto=None, what="d2h", start=None):
        """device -> host of the arrays from ``start`` (default the first) up to and including ``upto`` (default the
        last) on the compute stream, then wait for THAT copy (an event of this arena, not a synchronisation of the
        whole stream)"""
        if self.dev is not self.host:
            n = self.nbytes if upto is None else self.end[upto]
            o = 0 if start is None else self.begin[start]
            cur = self._cur()
            self._copy(self.host[o:n], self.dev[o:n], what, n - o, cur, current=self.stream is None)
            if self.done is None:
                self.done = torch.cuda.Event()
            self.done.record(cur)
            self.done.synchronize()

    # ---- piecewise transfers on a copy stream of the arena's own: overlapped with the host's model calls ----------------
    def _side(self):
        if self.side is None:
            self.side = torch.cuda.Stream(self.device)
        return self.side

    def push(self, name, what="h2d"):
        """host -> device of ONE array, asynchronously on the arena's copy stream: the caller goes on filling the next
        array (a model getter) while this one is on the wire.  The compute stream picks the data up with ``fence()``."""
        if self.dev is self.host:
            return
        o, n = self.begin[name], self.end[name]
        if n - o < self.PIECE_MIN:                  # small arrays: collected, sent as ONE copy at fence() (a copy of ~1 MB
            lo, hi = self.deferred or (o, n)        # reaches half the PCIe rate of a large one: overlap would not pay)
            self.deferred, self.deferred_what = (min(lo, o), max(hi, n)), what
            return
        side = self._side()
        if not self.pushed:      # first push since the last fence(): a kernel launched earlier on the compute stream may still
            side.wait_stream(self._cur())                                # be reading this buffer (write-after-read)
        self._copy(self.dev[o:n], self.host[o:n], what, n - o, side)
        self._mark_sent([name], side)
        self.pushed = True

    def fence(self):
        """the COMPUTE stream waits for every ``push()`` issued so far (call before launching the kernel that reads them)"""
        if self.dev is self.host:
            return
        cur = self._cur()
        if self.deferred is not None:
            lo, hi = self.deferred
            self.deferred = None
            self._copy(self.dev[lo:hi], self.host[lo:hi], self.deferred_what, hi - lo, cur, current=self.stream is None)
            self._mark_sent(self._names_in(lo, hi), cur)
        if self.pushed:
            cur.wait_stream(self.side)
            self.pushed = False

    def settle(self):
        """the COMPUTE stream waits for pulls nobody has waited for yet (call before launching a kernel that overwrites
        this buffer on the device); free when every pull has been taken with ``ready()``, the normal case"""
        if self.dev is not self.host and self.pending:
            self._cur().wait_stream(self.side)

    def pull(self, names, what="d2h"):
        """device -> host of the named arrays, one after the other on the copy stream, each followed by an event of its
        own (``ready(name)`` waits for it): the host hands the first array to a model setter while the next ones are still
        on the wire.  The copy stream first waits for what the compute stream has been given so far (the producing kernel)."""
        if self.dev is self.host:
            return
        side = self._side()
        side.wait_stream(self._cur())
        if all(self.end[nm] - self.begin[nm] < self.PIECE_MIN for nm in names):     # small: ONE copy of the covering range
            lo, hi = min(self.begin[nm] for nm in names), max(self.end[nm] for nm in names)
            self._copy(self.host[lo:hi], self.dev[lo:hi], what, hi - lo, side)
            ev = self.landed.get("*")
            if ev is None:
                ev = self.landed["*"] = torch.cuda.Event()
            ev.record(side)
            for nm in names:
                self.pending[nm] = ev
            return
        for name in names:
            o, n = self.begin[name], self.end[name]
            self._copy(self.host[o:n], self.dev[o:n], what, n - o, side)
            ev = self.landed.get(name)
            if ev is None:
                ev = self.landed[name] = torch.cuda.Event()
            ev.record(side)
            self.pending[name] = ev

    def ready(self, name=None):
        """wait until the pulled array ``name`` (default: every pulled array) is in host memory"""
        if name is None:
            for ev in self.pending.values():
                ev.synchronize()
            self.pending.clear()
        else:
            ev = self.pending.pop(name, None)
            if ev is not None:
                ev.synchronize()


class StepTrace:
    """Optional per-step breakdown of the drop-in path (bench.py `dropin`, tools/dropin_breakdown.py): every PCIe copy
    and kernel launch of spcpl is bracketed by HIP events on the stream it is issued on.  Install with
    ``transfer.trace = StepTrace()``; ``summary()`` synchronises and returns name -> {calls, ms, bytes, GBs}."""

    def __init__(self):
        self.items = []

    class _Region:
        __slots__ = ("tr", "name", "nbytes", "device", "e0")

        def __init__(self, tr, name, nbytes, device):
            self.tr, self.name, self.nbytes, self.device = tr, name, nbytes, device

        def __enter__(self):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.device))

        def __exit__(self, *exc):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.tr.items.append((self.name, self.nbytes, self.e0, e1))

    def region(self, name, nbytes, device):
        return StepTrace._Region(self, name, nbytes, device)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, nbytes, e0, e1 in self.items:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
        for d in out.values():
            d["GBs"] = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 and d["bytes"] else None
        return out


trace = None       # a StepTrace while a breakdown is being taken, else None (no events, no overhead)


class Sharded:
    """One [n x ...] array living as contiguous ROW BLOCKS on several devices (``parts[d]`` holds rows
    ``bounds[d]:bounds[d+1]`` on device d), or -- ``bounds`` None -- a small array replicated on every device (the
    shared LES grid).  Only what the coupling path needs: shape, per-device parts, gathering onto one device."""

    def __init__(self, parts, bounds=None):
        self.parts, self.bounds = list(parts), bounds

    @property
    def shape(self):
        p = self.parts[0]
        return tuple(p.shape) if self.bounds is None else (self.bounds[-1],) + tuple(p.shape[1:])

    def dim(self):
        return self.parts[0].dim()

    def __getitem__(self, key):
        """``t[:n]`` with n = all rows (what ColumnBatch asks of the GCM arrays): the sharded rows are the SP rows already"""
        if isinstance(key, slice) and key.start in (None, 0) and key.step in (None, 1) and (
                key.stop is None or self.bounds is None or key.stop == self.bounds[-1]):
            return self
        raise IndexError("a Sharded array can only be taken whole")

    #: how often any Sharded array was gathered onto ONE device (tests assert that the hot path, the slow paths and the
    #: nudge never do: results leave the devices block by block, straight to the host)
    gather_calls = 0

    def map(self, fn, *others):
        """``fn(part, other_part, ...)`` on every device's block (``others``: Sharded arrays with the same partition, or
        plain values handed to every call); a Sharded of the results"""
        parts = [fn(p, *[(o.parts[d] if isinstance(o, Sharded) else o) for o in others]) for d, p in enumerate(self.parts)]
        return Sharded(parts, self.bounds)

    def to_host(self):
        """NumPy array of the whole thing: every block copied to the host from ITS device and concatenated there -- the
        host-side gather of north_star (spifs output, setter fan-out); no device ever holds the whole batch"""
        import numpy
        for p in self.parts:                      # produced on the engines' own streams
            if p.device.type == "cuda":
                torch.cuda.synchronize(p.device)
        if self.bounds is None:
            return self.parts[0].cpu().numpy()
        blocks = [p.cpu().numpy() for p in self.parts if p.shape[0]]
        return numpy.concatenate(blocks, axis=0) if blocks else self.parts[0].cpu().numpy()

    def gather(self, device=None):
        """the whole array on ONE device.  Nothing in the package calls it any more (round 5: the slow paths, the sputils
        helpers and the nudge run per device on that device's rows); kept for callers that want a device-resident copy."""
        Sharded.gather_calls += 1
        device = self.parts[0].device if device is None else device
        for p in self.parts:                      # the parts may have been produced on other streams / devices
            if p.device.type == "cuda":
                torch.cuda.synchronize(p.device)
        if self.bounds is None:
            return self.parts[0].to(device)
        return torch.cat([p.to(device) for p in self.parts if p.shape[0]], dim=0)


class ShardedArena:
    """``Arena`` for several devices: the SAME single pinned host buffer with full-length host views (so everything on
    the host side -- model getters writing into it, setters reading from it -- is unchanged), mirrored on device d by a
    buffer that holds rows ``bounds[d]:bounds[d+1]`` of every array.  Copies go per (device, array) on that device's
    current stream -- the devices' DMA engines run concurrently -- and a download waits for one event per device.
    ``rows``: the number of leading rows that are sharded (arrays may be longer on the host: the extra output columns of
    the GCM state stay host-only)."""

    ALIGN = Arena.ALIGN

    def __init__(self, devices, bounds, specs, rows, streams=None):
        self.devices = [torch.device(d) for d in devices]
        self.device = self.devices[0]
        self.streams = list(streams) if streams is not None else [None] * len(self.devices)   # per "device": its own stream or None
        self.bounds, self.rows = list(bounds), rows
        off, lay = 0, []
        for name, shape, dtype in specs:
            nbytes = int(numpy.prod(shape, dtype=numpy.int64)) * torch.empty((), dtype=dtype).element_size()
            lay.append((name, tuple(shape), dtype, off, nbytes))
            off += -(-nbytes // self.ALIGN) * self.ALIGN
        self.nbytes = off
        on_gpu = self.device.type == "cuda"
        self.host = torch.empty(max(off, 1), dtype=torch.uint8, pin_memory=on_gpu)
        self.h, self.d, self.hn, self.begin, self.end, self.order = {}, {}, {}, {}, {}, []
        parts = {name: [] for name, *_ in lay}
        for di, dev in enumerate(self.devices):
            lo, hi = self.bounds[di], self.bounds[di + 1]
            o_d, lay_d = 0, []
            for name, shape, dtype, _, _ in lay:
                if not shape or shape[0] < rows:
                    raise ValueError("ShardedArena: %s has %s rows, fewer than the %d sharded rows" % (name, shape[:1], rows))
                shp = (hi - lo,) + shape[1:]
                nb = int(numpy.prod(shp, dtype=numpy.int64)) * torch.empty((), dtype=dtype).element_size()
                lay_d.append((name, shp, dtype, o_d, nb))
                o_d += -(-nb // self.ALIGN) * self.ALIGN
            buf = torch.empty(max(o_d, 1), dtype=torch.uint8, device=dev)
            for name, shp, dtype, o, nb in lay_d:
                parts[name].append(buf[o:o + nb].view(dtype).view(shp))
        for name, shape, dtype, o, nb in lay:
            self.h[name] = self.host[o:o + nb].view(dtype).view(shape)
            self.hn[name] = self.h[name].numpy()
            self.d[name] = Sharded(parts[name], self.bounds)
            self.begin[name], self.end[name] = o, o + nb
            self.order.append(name)
        self.done = [None] * len(self.devices)
        self.pending = {}
        self.sent = {}          # name -> events behind the last upload of that array, one per device (writable())

    def _on(self, di):
        """context in which device di's copies are issued: its device, and its own stream if it has one"""
        dev = self.devices[di]
        if dev.type != "cuda":
            return _Nothing()
        st = self.streams[di]
        return torch.cuda.stream(st) if st is not None else torch.cuda.device(dev)

    def _names(self, upto, start):
        i0 = 0 if start is None else self.order.index(start)
        i1 = len(self.order) if upto is None else self.order.index(upto) + 1
        return self.order[i0:i1]

    def upload(self, upto=None, what="h2d", start=None):
        names = self._names(upto, start)
        for di, dev in enumerate(self.devices):
            lo, hi = self.bounds[di], self.bounds[di + 1]
            if hi == lo:
                continue
            with self._on(di):
                for name in names:
                    self.d[name].parts[di].copy_(self.h[name][lo:hi], non_blocking=True)
                if dev.type == "cuda":
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    for name in names:
                        self.sent.setdefault(name, {})[di] = ev

    def writable(self, name):
        """wait until every device's last upload of ``name`` has left the pinned host buffer (Arena.writable)"""
        for ev in self.sent.pop(name, {}).values():
            ev.synchronize()

    def download(self, upto=None, what="d2h", start=None):
        names = self._names(upto, start)
        for di, dev in enumerate(self.devices):
            lo, hi = self.bounds[di], self.bounds[di + 1]
            if hi == lo:
                continue
            with self._on(di):
                for name in names:
                    self.h[name][lo:hi].copy_(self.d[name].parts[di], non_blocking=True)
                if dev.type == "cuda":
                    if self.done[di] is None:
                        self.done[di] = torch.cuda.Event()
                    self.done[di].record(torch.cuda.current_stream(dev))
        for ev in self.done:
            if ev is not None:
                ev.synchronize()


    # ---- piecewise transfers (as Arena's): every device's copies go on that device's own stream, so there is nothing to
    #      fence; an array is ready when every device's piece of it has landed
    def push(self, name, what="h2d"):
        self.upload(upto=name, start=name, what=what)

    def fence(self):
        pass

    def settle(self):
        pass

    def pull(self, names, what="d2h"):
        for di, dev in enumerate(self.devices):
            lo, hi = self.bounds[di], self.bounds[di + 1]
            if hi == lo:
                continue
            with self._on(di):
                for name in names:
                    self.h[name][lo:hi].copy_(self.d[name].parts[di], non_blocking=True)
                    if dev.type == "cuda":
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(dev))
                        self.pending.setdefault(name, []).append(ev)

    def ready(self, name=None):
        names = list(self.pending) if name is None else [name]
        for nm in names:
            for ev in self.pending.pop(nm, ()):
                ev.synchronize()


class _Nothing:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
