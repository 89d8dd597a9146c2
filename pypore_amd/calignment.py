"""`cSegmentAligner` with the surface of PyPore/calignment.pyx:20-100, running on the MI355X
(ps_align_batch, csrc/seg_align.hpp).  Next-row component f-5 of SURVEY.md section 8.

    aligner = cSegmentAligner(model_means, model_stds, model_durs, skip_penalty, backslip_penalty)
    score, order = aligner.align(seq_means, seq_stds, seq_durs)        # the reference call
    results = aligner.align_batch([(means, stds, durs), ...])          # many sequences in one launch

Results are bit-exact with the compiled reference (tests/test_align.py); where the reference raises
(ValueError for an empty sequence, ZeroDivisionError for a zero std product, IndexError when the
traceback runs into model index 0 before the first segment or the model has one segment) the same
exception class is raised here.  Where the reference is undefined -- no final score above -1, its
double_argmax then returns an uninitialised int (calignment.pyx:11-18) -- IndexError is raised.
There is no CPU fallback: without the GPU library the call fails.
"""
import numpy as np
import torch

from . import _lib, engine

_ERRORS = {
    _lib.PS_ALIGN_VALUE_ERROR: (ValueError, "Invalid shape in axis 0: 0."),
    _lib.PS_ALIGN_INDEX_ERROR: (IndexError, "Out of bounds on buffer access (axis 1)"),
    _lib.PS_ALIGN_ZERO_DIVISION: (ZeroDivisionError, "float division"),
    _lib.PS_ALIGN_UNDEFINED: (IndexError, "no final score above -1: the reference's double_argmax is undefined here"),
}


class cSegmentAligner(object):
    def __init__(self, model_means, model_stds, model_durs, skip_penalty, backslip_penalty):
        self.model_means = np.ascontiguousarray(model_means, dtype=np.float64)
        self.model_stds = np.ascontiguousarray(model_stds, dtype=np.float64)
        self.model_dur = np.ascontiguousarray(model_durs, dtype=np.float64)
        self.c_model_dur = np.cumsum(self.model_dur)
        self.skip_penalty = float(skip_penalty)
        self.backslip_penalty = float(backslip_penalty)

    def align_batch_raw(self, seqs, device=None):
        """seqs: list of (means, stds, durs).  Returns (scores float64 [n] = score[s-1, m-1], list of uint32 paths,
        status int32 [n]) as numpy arrays; no exception for per-sequence failures."""
        ctx = engine.context(device)
        lens = [len(s[0]) for s in seqs]
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        dev = torch.device("cuda", ctx.device)
        cols = []
        for k in range(3):
            col = (np.concatenate([np.asarray(s[k], dtype=np.float64).ravel() for s in seqs]) if seqs
                   else np.zeros(0, np.float64))
            assert col.size == off[-1], "means, stds and durations of a sequence must have the same length"
            cols.append(torch.from_numpy(col if col.size else np.zeros(1, np.float64)).to(dev))
        scores, paths, status = ctx.align_batch(self.model_means, self.model_stds, self.model_dur, self.skip_penalty,
                                                self.backslip_penalty, cols[0], cols[1], cols[2], off)
        paths = paths.cpu().numpy().view(np.uint32)
        return scores.cpu().numpy(), [paths[off[q]:off[q + 1]] for q in range(len(seqs))], status.cpu().numpy()

    def align_batch(self, seqs, device=None):
        """Per sequence what align() returns, or the exception INSTANCE the reference would raise."""
        scores, paths, status = self.align_batch_raw(seqs, device)
        out = []
        for q, s in enumerate(seqs):
            if status[q]:
                cls, msg = _ERRORS[int(status[q])]
                out.append(cls(msg))
            else:
                out.append((scores[q] / np.sum(np.asarray(s[2], dtype=np.float64)), paths[q].astype(np.float64)))
        return out

    def align(self, seq_means, seq_stds, seq_durs):
        """calignment.pyx:33-34,100: (score[s-1, m-1] / np.sum(seq_durs), float64 array of model indices)."""
        r = self.align_batch([(seq_means, seq_stds, seq_durs)])[0]
        if isinstance(r, Exception):
            raise r
        return r
