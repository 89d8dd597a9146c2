"""ABF2 reader and writer (the format PyPore/read_abf.py:22-212 consumes).

read_abf(path) -> (time_step_msec, float64 current) reproduces the reference's arithmetic:
channel 0 of the little-endian int16 data section, `* scale + offset` in float64 with
scale = fADCRange / fInstrumentScaleFactor / fSignalGain / fADCProgrammableGain / lADCResolution
(/ fTelegraphAdditGain when telegraphed, read_abf.py:202-203) and offset = fInstrumentOffset -
fSignalOffset (:205).  read_abf_counts() hands back the raw int16 counts plus (scale, offset) so the
GPU can consume 2 B/sample (PS_DTYPE_I16).  write_abf() produces the minimal ABF2 file both readers
(this one and the reference's) accept; the reference ships a reader only, so synthetic .abf inputs
for configs 3/4 come from here.

Layout facts (from the struct formats of the reference reader): 512-byte blocks; header = block 0
with the signature 0x32464241 at byte 0, uFileInfoSize at byte 8 and a table of 18 sections
(uint32 block index, uint32 bytes per entry, int64 entries) from byte 76 -- Protocol 0, ADC 1,
Data 10; protocol block: fADCSequenceInterval (us) at field 1, fADCRange field 33, lADCResolution
field 35; ADC block (128 B per channel): nTelegraphEnable 1, fTelegraphAdditGain 3,
fADCProgrammableGain 10, fInstrumentScaleFactor 13, fInstrumentOffset 14, fSignalGain 15,
fSignalOffset 16.
"""
import struct

import numpy as np

BLOCK = 512
SIGNATURE = 0x32464241                    # 'ABF2', little endian
_HEADER = struct.Struct("<7I4hI16s5I" + 18 * "IIq" + "148x")
_PROTOCOL = struct.Struct("<hf?3xIff5l3hf3h3flfhfhlllhflhffll3hl2h6h2hhlhhf5h3h3f5h304x")
_ADC = struct.Struct("<h2h3fhf2h9f2cfc?h2l46x")
_SECTION0 = 18                            # index of the first section triple in the unpacked header
_SEC_PROTOCOL, _SEC_ADC, _SEC_DATA = 0, 1, 10
assert _HEADER.size == BLOCK and _PROTOCOL.size == BLOCK and _ADC.size == 128


def _section(h, k):
    i = _SECTION0 + 3 * k
    return h[i], h[i + 1], h[i + 2]


def _read_meta(f):
    head = f.read(BLOCK)
    if len(head) < BLOCK:
        raise ValueError("not an ABF2 file (short header)")
    h = _HEADER.unpack(head)
    if h[0] != SIGNATURE:
        raise ValueError("not an ABF2 file (bad signature)")
    if h[2] != BLOCK:
        raise ValueError("unexpected uFileInfoSize %d" % h[2])
    pblk, pbytes, pnum = _section(h, _SEC_PROTOCOL)
    f.seek(pblk * BLOCK)
    p = _PROTOCOL.unpack(f.read(pbytes * pnum)[:BLOCK])
    time_step_msec = p[1] * 1e-3                          # fADCSequenceInterval [us] (read_abf.py:155)
    adc_range, adc_resolution = p[33], p[35]
    ablk, abytes, anum = _section(h, _SEC_ADC)
    f.seek(ablk * BLOCK)
    raw = f.read(abytes * anum)
    scales, offsets = [], []
    for ch in range(anum):
        a = _ADC.unpack(raw[ch * abytes:ch * abytes + _ADC.size])
        scale = adc_range / a[13] / a[15] / a[10] / adc_resolution     # read_abf.py:202
        if a[1]:
            scale /= a[3]                                                # read_abf.py:203
        scales.append(scale)
        offsets.append(a[14] - a[16])                                    # read_abf.py:205
    dblk, _, dnum = _section(h, _SEC_DATA)
    return time_step_msec, scales, offsets, anum, dblk * BLOCK, dnum


def read_abf_counts(path):
    """(time_step_msec, int16 counts of channel 0 (memmap view), scale, offset)."""
    with open(path, "rb") as f:
        dt, scales, offsets, nch, data_off, nentries = _read_meta(f)
    mm = np.memmap(path, mode="r", dtype=np.dtype("<i2"), offset=data_off)
    return dt, mm[:nentries:nch], scales[0], offsets[0]


def read_abf(path):
    """(time_step_msec, float64 current) -- the reference's read_abf (read_abf.py:208-212).  The array is a
    grid.GridArray: an ordinary float64 ndarray for every consumer, which also remembers the int16 counts, scale and
    offset it was computed from, so that the parsers can send 2 B/sample to the GPU instead of reverse-engineering
    the grid (real headers give scales like 10 / 0.0005 / 20 / 32768, never a power of two)."""
    from .grid import GridArray
    dt, counts, scale, offset = read_abf_counts(path)
    return dt, GridArray.from_counts(counts, scale, offset)


def write_abf(path, counts, sampling_interval_us=10.0, adc_range=1.0, adc_resolution=32,
              instrument_scale=1.0, signal_gain=1.0, programmable_gain=1.0,
              instrument_offset=0.0, signal_offset=0.0, telegraph_gain=None, other_channels=()):
    """Writes an ABF2 file whose channel 0 holds int16 `counts`.  The defaults give a scale of
    exactly 2**-5 pA per count and 100 kHz (header floats are fp32, so use
    power-of-two settings for an exactly representable scale).

    telegraph_gain: sets nTelegraphEnable and fTelegraphAdditGain of channel 0 (the reader divides the scale by it,
    read_abf.py:203).  other_channels: int16 arrays of the same length, interleaved behind channel 0 sample by sample
    (ADCNumEntries = 1 + len(other_channels); the readers take every ADCNumEntries-th entry, read_abf.py:210); their
    ADC blocks carry unit settings."""
    counts = np.ascontiguousarray(counts, dtype="<i2")
    n = counts.size
    nch = 1 + len(other_channels)
    if nch > 1:
        cols = [counts] + [np.ascontiguousarray(c, dtype="<i2") for c in other_channels]
        if any(c.size != n for c in cols):
            raise ValueError("all channels must hold the same number of samples")
        data = np.stack(cols, axis=1).reshape(-1)
    else:
        data = counts
    adc_blocks = (nch * _ADC.size + BLOCK - 1) // BLOCK
    sections = [(0, 0, 0)] * 18
    sections[_SEC_PROTOCOL] = (1, BLOCK, 1)
    sections[_SEC_ADC] = (2, _ADC.size, nch)
    sections[_SEC_DATA] = (2 + adc_blocks, 2, n * nch)
    flat = [v for s in sections for v in s]
    head = _HEADER.pack(SIGNATURE, 0x02000000, BLOCK, 1, 0, 0, 0, 1, 0, 1, 0, 0, b"\0" * 16, 0, 0, 0, 0, 0, *flat)
    p = list(_PROTOCOL.unpack(b"\0" * BLOCK))
    p[0] = 3                                            # nOperationMode: gap-free
    p[1] = float(sampling_interval_us)
    p[33] = float(adc_range)
    p[35] = int(adc_resolution)
    proto = _PROTOCOL.pack(*p)
    adc = b""
    for ch in range(nch):
        a = list(_ADC.unpack(b"\0" * _ADC.size))
        a[0] = ch                                       # nADCNum
        a[3] = 1.0                                      # fTelegraphAdditGain
        a[10] = a[11] = a[13] = a[15] = 1.0
        if ch == 0:
            if telegraph_gain is not None:
                a[1] = 1                                # nTelegraphEnable
                a[3] = float(telegraph_gain)
            a[10] = float(programmable_gain)
            a[13] = float(instrument_scale)
            a[14] = float(instrument_offset)
            a[15] = float(signal_gain)
            a[16] = float(signal_offset)
        adc += _ADC.pack(*a)
    with open(path, "wb") as f:
        f.write(head)
        f.write(proto)
        f.write(adc + b"\0" * (adc_blocks * BLOCK - len(adc)))
        f.write(data.tobytes())
    return path
