"""IEEE RTS-79 chronological load model of the reference's sequential track.

  case24_loadprofile()  <- Montecarlo_seq/case24_loadprofile.m:17-95 (weekly / daily / hourly factors, bus peaks)
  anloducurve(hours)    <- Montecarlo_seq/anloducurve.m:24-93 (load factor = week x day x hour-of-day(season, daytype))

Quirks kept as in the reference: the day index is
``ceil(mod(h/24, 7))`` with 0 -> 7 (anloducurve.m:39-40), seasons by week (:30-36).
"""
from __future__ import annotations

import math

import numpy as np


def case24_loadprofile() -> dict:
    weekly = np.array([
        0.862, 0.900, 0.878, 0.834, 0.880, 0.841, 0.832, 0.806,
        0.740, 0.737, 0.715, 0.727, 0.704, 0.750, 0.721, 0.800,
        0.754, 0.837, 0.870, 0.880, 0.856, 0.811, 0.900, 0.887,
        0.896, 0.861, 0.755, 0.816, 0.801, 0.880, 0.722, 0.776,
        0.800, 0.729, 0.726, 0.705, 0.780, 0.695, 0.724, 0.723,
        0.743, 0.744, 0.800, 0.881, 0.885, 0.909, 0.940, 0.890,
        0.942, 0.970, 1.000, 0.952])
    daily = np.array([0.93, 1.00, 0.98, 0.96, 0.94, 0.77, 0.75])
    hourly = np.array([
        [0.67, 0.78, 0.64, 0.74, 0.63, 0.75], [0.63, 0.72, 0.60, 0.70, 0.62, 0.73],
        [0.60, 0.68, 0.58, 0.66, 0.60, 0.69], [0.59, 0.66, 0.56, 0.65, 0.58, 0.66],
        [0.59, 0.64, 0.56, 0.64, 0.59, 0.65], [0.60, 0.65, 0.58, 0.62, 0.65, 0.65],
        [0.74, 0.66, 0.64, 0.62, 0.72, 0.68], [0.86, 0.70, 0.76, 0.66, 0.85, 0.74],
        [0.95, 0.80, 0.87, 0.81, 0.95, 0.83], [0.96, 0.88, 0.95, 0.86, 0.99, 0.89],
        [0.96, 0.90, 0.99, 0.91, 1.00, 0.92], [0.95, 0.91, 1.00, 0.93, 0.99, 0.94],
        [0.95, 0.90, 0.99, 0.93, 0.93, 0.91], [0.95, 0.88, 1.00, 0.92, 0.92, 0.90],
        [0.93, 0.87, 1.00, 0.91, 0.90, 0.90], [0.94, 0.87, 0.97, 0.91, 0.88, 0.86],
        [0.99, 0.91, 0.96, 0.92, 0.90, 0.85], [1.00, 1.00, 0.96, 0.94, 0.92, 0.88],
        [1.00, 0.99, 0.93, 0.95, 0.96, 0.92], [0.96, 0.97, 0.92, 0.95, 0.98, 1.00],
        [0.91, 0.94, 0.92, 1.00, 0.96, 0.97], [0.83, 0.92, 0.93, 0.93, 0.90, 0.95],
        [0.73, 0.87, 0.87, 0.88, 0.80, 0.90], [0.63, 0.81, 0.72, 0.80, 0.70, 0.85]])
    busload = np.array([
        [1, 108, 22], [2, 97, 20], [3, 180, 37], [4, 74, 15], [5, 71, 14], [6, 136, 28],
        [7, 125, 25], [8, 171, 35], [9, 175, 36], [10, 195, 40], [13, 265, 54], [14, 194, 39],
        [15, 317, 64], [16, 100, 20], [18, 333, 68], [19, 181, 37], [20, 128, 26]], dtype=np.float64)
    return dict(MW=2850.0, MVAr=580.0, weekly=weekly, daily=daily, hourly=hourly, busload=busload)


def anloducurve(total_hours: int = 8736):
    """Returns (busPd [nbus_with_load x hours], busQd, load_factors [hours]) as anloducurve.m."""
    prof = case24_loadprofile()
    lf = np.zeros(total_hours)
    for h in range(1, total_hours + 1):
        week = math.ceil(h / 168)
        if week <= 8 or week >= 44:
            season = 0            # winter
        elif 18 <= week <= 30:
            season = 1            # summer
        else:
            season = 2            # spring / fall
        day = math.ceil((h / 24) % 7)          # anloducurve.m:39 (kept as is)
        if day == 0:
            day = 7
        weekend = day > 5
        hod = h % 24
        if hod == 0:
            hod = 24
        col = 2 * season + (1 if weekend else 0)
        lf[h - 1] = prof["weekly"][week - 1] * prof["daily"][day - 1] * prof["hourly"][hod - 1, col]
    busPd = np.outer(prof["busload"][:, 1], lf)
    busQd = np.outer(prof["busload"][:, 2], lf)
    return busPd, busQd, lf
