"""ACT/365 time axis in millisecond ticks — /root/reference/src/date_functions.jl:1-58."""
from __future__ import annotations

import datetime as _dt

SECONDS_IN_YEAR_365 = 365 * 86400
MILLISECONDS_IN_YEAR_365 = SECONDS_IN_YEAR_365 * 1000  # date_functions.jl:2
MILLISECONDS_IN_DAY = 86400000

Date = _dt.date
DateTime = _dt.datetime

# Julia's Dates.date2epochdays counts days from 0000-01-01; Python's toordinal() from 0001-01-01 = 1
_EPOCH_SHIFT_DAYS = 365


def to_ticks(x):
    """date_functions.jl:15-41: Date/DateTime -> ms since 0000-01-01; numbers pass through."""
    if isinstance(x, _dt.datetime):
        days = x.toordinal() + _EPOCH_SHIFT_DAYS
        ms = ((x.hour * 60 + x.minute) * 60 + x.second) * 1000 + x.microsecond // 1000
        return days * MILLISECONDS_IN_DAY + ms
    if isinstance(x, _dt.date):
        return (x.toordinal() + _EPOCH_SHIFT_DAYS) * MILLISECONDS_IN_DAY
    return x


def yearfrac(start, stop):
    """date_functions.jl:54-58."""
    return (to_ticks(stop) - to_ticks(start)) / MILLISECONDS_IN_YEAR_365


def add_years(d: _dt.date, years: int) -> _dt.date:
    """`date + Year(n)` of Julia's Dates (Feb 29 clamps to Feb 28)."""
    try:
        return d.replace(year=d.year + years)
    except ValueError:
        return d.replace(year=d.year + years, day=28)
