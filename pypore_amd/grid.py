"""Currents that remember the ADC counts they came from.

An .abf file holds int16 counts; the reader returns `counts * scale + offset` as float64 (read_abf.py:202-210) and
the header scale is almost never a power of two (10 V / 0.0005 V/pA / 20 / 32768 = 0.0305...).  The device segmenter
works on exact integer sums of the counts, so the float64 array has to find its way back to them:

* GridArray -- a float64 ndarray (drop-in for what the reference reader returns) that carries `counts`, `quantum`
  and `offset`; basic slices keep the triple, every other operation yields a plain array.  pypore_amd.abf.read_abf
  returns one, so File(filename).parse(), Event.parse(SpeedyStatSplit(...)) and parser.parse(read_abf(f)[1]) take
  the int16 route (2 B/sample over PCIe and HBM) without the caller doing anything.
* affine_grid() -- recovers (quantum, offset, counts) from a bare float64 array `k * q + o` of any origin.
"""
import threading

import numpy as np


class GridArray(np.ndarray):
    """float64 current `counts * quantum + offset` together with its int16/int32 counts."""

    def __new__(cls, current, counts, quantum, offset=0.0):
        obj = np.asarray(current).view(cls)
        obj.counts, obj.quantum, obj.offset = counts, float(quantum), float(offset)
        # The float64 values and the counts must stay in step: in-place edits (a -= baseline, a[i:j] = x, a.sort())
        # return the same object without passing __array_finalize__ and would leave the counts stale, so the array is
        # read-only -- such an edit raises ValueError; `np.array(a)` / `a.copy()` give a writable plain array.
        obj.setflags(write=False)
        return obj

    def __array_finalize__(self, obj):
        self.counts = None                          # only basic slicing below keeps the counts
        self.quantum = getattr(obj, 'quantum', None)
        self.offset = getattr(obj, 'offset', 0.0)

    def __getitem__(self, index):
        out = np.ndarray.__getitem__(self, index)
        if isinstance(out, GridArray) and isinstance(index, slice) and self.counts is not None:
            out.counts = self.counts[index]
        return out

    def copy(self, order='C'):
        """A writable plain float64 array (the counts do not follow edits)."""
        return np.array(self, dtype=np.float64, order=order, subok=False)

    @classmethod
    def from_counts(cls, counts, quantum, offset=0.0):
        """What the reference reader computes (read_abf.py:210): float64(counts) * scale + offset."""
        current = np.array(counts, dtype=np.float64)        # (same two roundings, without the two temporaries)
        current *= quantum
        if offset:
            current += offset
        return cls(current, counts, quantum, offset)


def grid_of(current):
    """(counts, quantum, offset) if `current` still knows its counts, else None."""
    if isinstance(current, GridArray) and current.counts is not None and current.counts.shape == current.shape:
        return current.counts, current.quantum, current.offset
    if isinstance(current, Deferred):
        if current.counts is not None:
            return current.counts, current.quantum, current.offset
        return grid_of(current.value())
    return None


def affine_grid(x, sample=65536, tol=1e-6):
    """(quantum, offset, counts int64) with x == counts * quantum + offset to within float64 rounding, for an array
    that came from integer ADC counts through any scale and offset; ValueError if there is no such grid.  The spacing
    is the smallest gap between distinct values of a subset (noisy traces visit neighbouring counts), confirmed on
    every sample."""
    x = np.asarray(x, dtype=np.float64)
    if x.size == 0:
        return 1.0, 0.0, np.zeros(0, dtype=np.int64)
    u = np.unique(x[::max(1, x.size // sample)])
    if u.size < 2:
        u = np.unique(x)
    if u.size < 2:
        return 1.0, float(u[0]), np.zeros(x.size, dtype=np.int64)
    gaps = np.diff(u)
    q = float(gaps.min())
    # the smallest gap may be a multiple of the true spacing only if no two neighbouring counts occur: the check on the
    # whole array below then fails for the samples in between and a finer candidate is tried
    for _ in range(4):
        k = (x - u[0]) / q
        kr = np.rint(k)
        if np.max(np.abs(k - kr)) <= tol:
            if np.max(np.abs(kr)) >= 2 ** 31:
                break
            # re-centre: the counts are anchored at the subset minimum (0 .. max - min), and a trace that touches both
            # int16 rails would leave the int16 range although it is exactly int16 -- shift by the mid-range count
            ki = kr.astype(np.int64)
            c = (int(ki.min()) + int(ki.max()) + 1) // 2
            return q, float(u[0]) + c * q, ki - c
        frac = np.abs(k - kr)
        q = float(np.min(frac[frac > tol])) * q              # a remainder that is itself on the grid, or garbage
    raise ValueError("samples are not on an ADC grid (counts * quantum + offset); pass quantum= and offset=")


class Deferred(object):
    """A float64 current that has not been written out yet.

    `File(filename)` and the device filter produce currents of 10^8 samples that most callers never look at sample by
    sample: the event detector and the segmenter take the int16 counts (or the device tensor) they came from, and the
    segments carry their own statistics.  A Deferred stands in for such an array until somebody asks for its values:
    `value()` (also `numpy.asarray(d)`) builds the float64 ndarray once and keeps it; `d[a:b]` of an unbuilt current is
    another Deferred over the same source, so events and segments cut from it stay unbuilt too.  The classes of
    core.py / DataTypes.py hold it behind their `current` attribute, which builds it on first access -- user code only
    ever sees the ndarray (for a file: the GridArray the reference-style reader returns).

    Sources: int16 counts with scale and offset (`from_counts`; `grid_of` hands the triple to the device route), a
    float64 device tensor (`from_tensor`), or a stretch of another Deferred (slices share the parent's array once it
    exists)."""

    live_device_bytes = 0                                # bytes parked on GPUs, all devices (see from_tensor, device_counts)
    DEVICE_BYTES_MAX = 16 << 30                          # per device, and never more than a quarter of its memory
    _lock = threading.RLock()                            # the accounting is shared by the host threads of a process.  Re-entrant: a
                                                         # Deferred sits in event <-> segment cycles, so the cyclic collector may run its
                                                         # __del__ -> _unpark on THIS thread at any allocation inside _park
    _by_device = {}                                      # device index -> bytes parked there
    _cap_cache = {}                                      # device index -> [calls until the next look at the device, cap, bytes parked at that look]
    CAP_REFRESH = 64                                     # _park asks the driver for the free memory once per this many calls ...
    CAP_REFRESH_BYTES = 1 << 30                          # ... and whenever the parked bytes have grown by this much since the last look

    @classmethod
    def _device_cap(cls, dev_key, refresh=False):
        """DEVICE_BYTES_MAX, at most a quarter of the device's memory, and nothing while less than a tenth of the device is
        free.  Memory that torch's caching allocator holds but has not handed out counts as free (it is: the next tensor
        comes out of it).  The driver is asked once per CAP_REFRESH calls, and again whenever a request was refused."""
        ent = cls._cap_cache.get(dev_key)
        parked = cls._by_device.get(dev_key, 0)
        # (several ranks or threads share a GPU: "less than a tenth free" must not be 64 parks old once a GiB has gone in)
        if ent is not None and ent[0] > 0 and not refresh and parked - ent[2] < cls.CAP_REFRESH_BYTES:
            ent[0] -= 1
            return min(cls.DEVICE_BYTES_MAX, ent[1])
        limit = cls.DEVICE_BYTES_MAX
        try:
            import torch
            free, total = torch.cuda.mem_get_info(dev_key)
            free += max(0, torch.cuda.memory_reserved(dev_key) - torch.cuda.memory_allocated(dev_key))
            limit = 0 if free < total // 10 else total // 4
        except Exception:
            pass
        cls._cap_cache[dev_key] = [cls.CAP_REFRESH, limit, parked]
        return min(cls.DEVICE_BYTES_MAX, limit)

    @classmethod
    def _park(cls, dev_key, nbytes):
        """Reserve `nbytes` of the budget of parked tensors on device `dev_key` (None: no device, the accounting only).
        The budget is _device_cap -- an Experiment over many files, or several ranks on one GPU, must not run the
        allocator dry where the eager route (copy to the host at once) would have worked."""
        with cls._lock:
            cap = cls.DEVICE_BYTES_MAX if dev_key is None else cls._device_cap(dev_key)
            if cls._by_device.get(dev_key, 0) + nbytes > cap and dev_key is not None:
                cap = cls._device_cap(dev_key, refresh=True)     # a refusal is decided on fresh numbers
            used = cls._by_device.get(dev_key, 0)        # (read after the driver calls: a collection inside them may have unparked)
            if used + nbytes > cap:
                return False
            cls._by_device[dev_key] = used + nbytes
            cls.live_device_bytes += nbytes
            return True

    @classmethod
    def _unpark(cls, dev_key, nbytes):
        with cls._lock:
            cls._by_device[dev_key] = cls._by_device.get(dev_key, 0) - nbytes
            cls.live_device_bytes -= nbytes

    @staticmethod
    def _dev_key(tensor):
        dev = getattr(tensor, "device", None)
        return dev.index if dev is not None and getattr(dev, "type", "") == "cuda" else None

    def __init__(self, n, make, counts=None, quantum=None, offset=0.0, parent=None, start=0):
        self._n, self._make, self._value = int(n), make, None
        self.counts, self.quantum, self.offset = counts, quantum, float(offset)
        self._parent, self._start = parent, int(start)
        self.tensor = None

    # ---- constructors ---------------------------------------------------------------------------------------
    @classmethod
    def from_counts(cls, counts, quantum, offset=0.0):
        """counts * quantum + offset, as abf.read_abf computes it (a read-only GridArray when built)."""
        return cls(len(counts), lambda: GridArray.from_counts(counts, quantum, offset), counts, float(quantum), offset)

    @classmethod
    def from_tensor(cls, tensor, offset=0.0):
        """A float64 result that lives on the GPU (the filtered current of an event).  The tensors parked this way are
        capped at DEVICE_BYTES_MAX in total; beyond that the current is copied to the host right away."""
        nbytes = tensor.numel() * tensor.element_size()
        d = cls(tensor.numel(), None, offset=offset)
        d.tensor = tensor                                # (built in value(): a closure over `d` would keep it alive)
        d._park_key = cls._dev_key(tensor)
        if cls._park(d._park_key, nbytes):
            d._parked = nbytes
        else:
            d.value()                                    # over the budget: to the host right away, the tensor is let go
        return d

    def device_counts(self, dev, upload):
        """int16 CUDA tensor of this current's counts.  The root of a family of slices uploads its counts once
        (`upload(counts, dev)`) and keeps the tensor while it fits the budget of parked device bytes; stretches are
        views of it."""
        root = self._parent if self._parent is not None else self
        if root.counts is None or (self._parent is not None and self._parent.counts is None):
            return upload(self.counts, dev)
        cached = root.__dict__.get('_dev_counts')
        if cached is None or cached.device != dev:
            if self is not root and 4 * self._n < root._n:
                return upload(self.counts, dev)          # a short stretch of a file nobody has sent up: only the stretch
            if cached is not None:                       # parked on another device: that copy is given up first
                root._release_counts()
            cached = upload(root.counts, dev)
            nbytes = cached.numel() * cached.element_size()
            if Deferred._park(Deferred._dev_key(cached), nbytes):
                root._dev_counts, root._parked_counts, root._counts_key = cached, nbytes, Deferred._dev_key(cached)
        if self is root:
            return cached
        # (a view wherever it starts: 16-byte loads from 2-byte-aligned addresses are correct on this part -- round 5,
        #  tools/probes/unaligned_probe.hip; until then a stretch off an allocation boundary was cloned)
        return cached[self._start:self._start + self._n]

    def device_stretch(self, dev, upload):
        """(int16 CUDA tensor of the whole family's counts, first sample, length) of this stretch -- no copy, whatever its
        alignment -- or None when the family's counts are not parked on `dev` (callers that hand the library a base pointer
        and ranges: FastStatSplit.parse_filtered_batch)."""
        root = self._parent if self._parent is not None else self
        if root.counts is None or self.counts is None:
            return None
        self.device_counts(dev, upload)                  # (uploads and parks the family's counts if nobody has yet)
        cached = root.__dict__.get('_dev_counts')
        if cached is None or cached.device != dev:
            return None
        return (cached, 0, self._n) if self is root else (cached, self._start, self._n)

    def _release_counts(self):
        if getattr(self, "_parked_counts", 0):
            Deferred._unpark(getattr(self, "_counts_key", None), self._parked_counts)
            self._parked_counts = 0
        self.__dict__.pop('_dev_counts', None)

    def _release(self):
        self._release_counts()
        if getattr(self, "_parked", 0):
            Deferred._unpark(getattr(self, "_park_key", None), self._parked)
            self._parked = 0
        self.tensor = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- array-likeness -------------------------------------------------------------------------------------
    dtype = np.dtype(np.float64)
    ndim = 1

    def __len__(self):
        return self._n

    @property
    def shape(self):
        return (self._n,)

    @property
    def size(self):
        return self._n

    @property
    def built(self):
        return self._value is not None

    def value(self):
        if self._value is None:
            if self._parent is not None:
                self._value = self._parent.value()[self._start:self._start + self._n]
            elif self.tensor is not None:
                out = self.tensor.cpu().numpy()
                self._value = out + self.offset if self.offset else out
                self._release()
            else:
                self._value = self._make()
            self._make = None
        return self._value

    def __array__(self, dtype=None, copy=None):
        v = self.value()
        return v if dtype is None else np.asarray(v, dtype=dtype)

    def __getitem__(self, index):
        if self._value is None and isinstance(index, slice):
            a, b, step = index.indices(self._n)
            if step == 1:
                counts = self.counts[a:b] if self.counts is not None else None
                root, base = (self._parent, self._start) if self._parent is not None else (self, 0)
                if root._value is not None:              # (the family's array exists by now: a plain view of it)
                    return root._value[base + a:base + max(a, b)]
                return Deferred(max(0, b - a), None, counts, self.quantum, self.offset, parent=root, start=base + a)
        return self.value()[index]


def stretches(current, edges):
    """[current[a:z]] for consecutive edges (ascending ints inside the array).  For a current that has not been written
    out the stand-ins are filled in directly -- one per segment of a file's events, 10^4 .. 10^5 of them."""
    if not isinstance(current, Deferred) or current._value is not None:
        cur = built(current)
        return [cur[a:z] for a, z in zip(edges, edges[1:])]
    root, base = (current._parent, current._start) if current._parent is not None else (current, 0)
    if root._value is not None:
        v = root._value
        return [v[base + a:base + z] for a, z in zip(edges, edges[1:])]
    counts, q, o = current.counts, current.quantum, current.offset
    out, new = [], Deferred.__new__
    for a, z in zip(edges, edges[1:]):
        d = new(Deferred)
        d.__dict__.update(_n=z - a, _make=None, _value=None, counts=None if counts is None else counts[a:z], quantum=q,
                          offset=o, _parent=root, _start=base + a, tensor=None)
        out.append(d)
    return out


def built(current):
    """The ndarray behind `current` (a Deferred is built, anything else is returned as it is)."""
    return current.value() if isinstance(current, Deferred) else current
