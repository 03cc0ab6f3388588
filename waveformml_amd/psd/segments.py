"""Segment-status tables of the 14 x 11 detector: which segments are read out on one end only.  Host-side mirror of the
part of the reference's ``SingleEndedEvaluator`` that the single-ended-only loss needs (src/evaluation/
SingleEndedEvaluator.py:15-37) and of ``LitBase._format_SE_mask`` (src/engineering/LitBase.py:110-122).

PMT p sits on segment p // 2 (segment s at x = s % 14, y = s // 14), end p % 2; every dead PMT adds 0.5 to its segment's
status: 0 = both ends live, 0.5 = single-ended, 1 = dead.  The loss mask is 1 on single-ended segments and 0 elsewhere.
"""
import numpy as np
import torch

NX, NY = 14, 11

# detector constant: the PMTs that were off in the reference's data-taking period (SingleEndedEvaluator.py:18-21)
SE_DEAD_PMTS = (1, 0, 2, 4, 6, 7, 9, 10, 12, 13, 16, 19, 20, 21, 22, 24, 26, 27, 34, 36, 37, 43, 46, 48, 55, 54, 56, 58, 65,
                68, 72, 80, 82, 85, 88, 93, 95, 97, 96, 105, 111, 112, 120, 122, 137, 138, 139, 141, 147, 158, 166, 173,
                175, 188, 195, 215, 230, 243, 244, 245, 252, 255, 256, 261, 273, 279, 282)


def segment_status(dead_pmts=SE_DEAD_PMTS, nx=NX, ny=NY):
    status = np.zeros((nx, ny), dtype=np.float32)
    for pmt in dead_pmts:
        seg = pmt // 2
        status[seg % nx, seg // nx] += 0.5
    return status


def single_ended_mask(status):
    """[1, 1, nx, ny] float mask: 1 where exactly one end is live."""
    st = torch.as_tensor(status)
    return (st == 0.5).to(torch.float32).unsqueeze(0).unsqueeze(0)
