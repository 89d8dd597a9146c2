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
