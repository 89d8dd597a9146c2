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
    """(time_step_msec, float64 current) -- the reference's read_abf (read_abf.py:208-212)."""
    dt, counts, scale, offset = read_abf_counts(path)
    return dt, np.array(counts, dtype=np.float64) * scale + offset


def write_abf(path, counts, sampling_interval_us=10.0, adc_range=1.0, adc_resolution=32,
              instrument_scale=1.0, signal_gain=1.0, programmable_gain=1.0,
              instrument_offset=0.0, signal_offset=0.0):
    """Writes a single-channel ABF2 file holding int16 `counts`.  The defaults give a scale of
    exactly 2**-5 pA per count and 100 kHz (header floats are fp32, so use
    power-of-two settings for an exactly representable scale)."""
    counts = np.ascontiguousarray(counts, dtype="<i2")
    n = counts.size
    sections = [(0, 0, 0)] * 18
    sections[_SEC_PROTOCOL] = (1, BLOCK, 1)
    sections[_SEC_ADC] = (2, _ADC.size, 1)
    sections[_SEC_DATA] = (3, 2, n)
    flat = [v for s in sections for v in s]
    head = _HEADER.pack(SIGNATURE, 0x02000000, BLOCK, 1, 0, 0, 0, 1, 0, 1, 0, 0, b"\0" * 16, 0, 0, 0, 0, 0, *flat)
    p = [0] * 79                                        # number of fields in the protocol block
    fmt_fields = _PROTOCOL.unpack(b"\0" * BLOCK)
    p = list(fmt_fields)
    p[0] = 3                                            # nOperationMode: gap-free
    p[1] = float(sampling_interval_us)
    p[33] = float(adc_range)
    p[35] = int(adc_resolution)
    proto = _PROTOCOL.pack(*p)
    a = list(_ADC.unpack(b"\0" * _ADC.size))
    a[3] = 1.0                                          # fTelegraphAdditGain (unused: nTelegraphEnable = 0)
    a[10] = float(programmable_gain)
    a[11] = 1.0
    a[13] = float(instrument_scale)
    a[14] = float(instrument_offset)
    a[15] = float(signal_gain)
    a[16] = float(signal_offset)
    adc = _ADC.pack(*a)
    with open(path, "wb") as f:
        f.write(head)
        f.write(proto)
        f.write(adc + b"\0" * (BLOCK - len(adc)))
        f.write(counts.tobytes())
    return path
