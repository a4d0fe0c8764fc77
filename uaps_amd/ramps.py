"""Consistency ramp-up schedule (host side, float64) -- reference utilities/ramps.py:19-26."""
import math


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t/R)^2) with t clipped to [0, R]; 1.0 when R == 0 (utilities/ramps.py:19-26)."""
    if rampup_length == 0:
        return 1.0
    t = float(current)
    t = 0.0 if t < 0.0 else (float(rampup_length) if t > rampup_length else t)
    phase = 1.0 - t / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def get_current_consistency_weight(consistency, iter_num, consistency_rampup=200, iters_per_ramp_step=80):
    """UAPS_train.py:81-87 called as at :279-280 with `iter_num // 80`."""
    return consistency * sigmoid_rampup(iter_num // iters_per_ramp_step, consistency_rampup)
