"""ORACLE — TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).  How close a log-assignment matrix's filter_matches decisions (src/light_glue.cpp:214-266)
sit to their boundaries: which rows a perturbation of a given size can flip, and each row's margin."""
import numpy as np


def fragile_rows(scores, tol, thr=0.1):
    """Rows of a log-assignment matrix whose filter_matches decision a perturbation of at most `tol` per entry can flip:
    the row maximum within `tol` of log(thr), or the runner-up of its row / of its column within 2 tol of the maximum.
    Parity of match SETS is asserted on all other rows; this set must stay (nearly) empty for the test to mean anything."""
    s = scores.astype(np.float64)
    n0, n1 = s.shape
    out = set()
    if n0 == 0 or n1 == 0:
        return out
    rcol = s.argmax(1)
    rval = s[np.arange(n0), rcol]
    lt = np.log(thr)

    def runner_up_gap(m, axis):
        if m.shape[axis] < 2:
            return np.full(m.shape[1 - axis], np.inf)
        part = np.sort(m, axis=axis)
        return (part.take(-1, axis) - part.take(-2, axis))
    rgap = runner_up_gap(s, 1)
    cgap = runner_up_gap(s, 0)
    for i in range(n0):
        if rval[i] < lt - 2 * tol and s[:, rcol[i]].argmax() != i:
            continue                                    # far below threshold AND not mutual: two flips needed
        if abs(rval[i] - lt) <= tol or (rval[i] > lt - tol and (rgap[i] <= 2 * tol or cgap[rcol[i]] <= 2 * tol)):
            out.add(i)
    return out


def decision_margins(scores, thr=0.1):
    """Per row of a log-assignment matrix: how far (in score units) the nearest entry change is that flips the row's filter_matches decision —
    min(|row max - log thr|, half the gap to the runner-up of its row, half the gap to the runner-up of its column).  A row on which the device
    and the oracle DISAGREE must have a margin below twice the measured score error: disagreements are then explained, not exempted."""
    s = np.where(np.isfinite(scores), scores, -1e30).astype(np.float64)
    n0, n1 = s.shape
    if n0 == 0 or n1 == 0:
        return np.zeros(n0)
    rcol = s.argmax(1)
    rval = s[np.arange(n0), rcol]

    def gap(m, axis):
        if m.shape[axis] < 2:
            return np.full(m.shape[1 - axis], np.inf)
        part = np.sort(m, axis=axis)
        return part.take(-1, axis) - part.take(-2, axis)
    rgap, cgap = gap(s, 1), gap(s, 0)
    return np.minimum(np.abs(rval - np.log(thr)), np.minimum(rgap, cgap[rcol]) / 2)
