"""`SegmentAligner` with the surface of PyPore/alignment.py:26-46: the wrapper DataTypes-level code uses around
cSegmentAligner.  align() returns (score, order) or (None, None) when the aligner raises ValueError
(alignment.py:43-46); other exceptions propagate as in the reference.  `transform` (alignment.py:48-107) is host
bookkeeping that reads `self.model`, which the reference never sets; it is not part of the accelerated path."""
from .calignment import cSegmentAligner


class SegmentAligner(object):
    def __init__(self, model_means, model_stds, model_durs, skip_penalty, backslip_penalty):
        self.aligner = cSegmentAligner(model_means, model_stds, model_durs, skip_penalty, backslip_penalty)

    def align(self, seq_means, seq_stds, seq_durs):
        try:
            return self.aligner.align(seq_means, seq_stds, seq_durs)
        except ValueError:
            return None, None

    def align_batch(self, seqs):
        """Many sequences in one launch; per sequence (score, order), (None, None) for a ValueError, or the
        exception instance the reference would have raised."""
        return [(None, None) if isinstance(r, ValueError) else r for r in self.aligner.align_batch(seqs)]
