"""Progbar -- the console progress line `train.py` feeds with the step's `logs` (reference utils/__init__.py:101-263, itself the
Keras utility): `Progbar(target, width=25, verbose=1, interval=0.05, stateful_metrics=None)`, `.update(current, values)`,
`.add(n, values)`.  `values` is the list of (name, value) pairs `optimize_parameters` returns; a name listed in `stateful_metrics`
is shown as-is, every other one as its average over the steps seen so far, weighted by the step sizes (so `add(len(real_H), logs)`
averages per frame).  Same line format as the reference's:

     48/1000 [>........................] - ETA: 1:02 - loss: 0.241273 - lr: 0.001
"""
import math
import sys
import time


class _Mean:
    __slots__ = ("total", "weight")

    def __init__(self):
        self.total, self.weight = 0.0, 0

    def push(self, value, weight):
        self.total += value * weight
        self.weight += weight

    def value(self):
        return self.total / max(1, self.weight)


def _fmt_seconds(s):
    if s > 3600:
        return "%d:%02d:%02d" % (s // 3600, (s % 3600) // 60, s % 60)
    if s > 60:
        return "%d:%02d" % (s // 60, s % 60)
    return "%ds" % s


def _fmt_rate(per_unit):
    if per_unit >= 1:
        return " %.0fs/step" % per_unit
    if per_unit >= 1e-3:
        return " %.0fms/step" % (per_unit * 1e3)
    return " %.0fus/step" % (per_unit * 1e6)


def _fmt_avg(v, signed=True):
    return (" %.6f" if (abs(v) if signed else v) > 1e-3 else " %.6e") % v


class Progbar(object):
    def __init__(self, target, width=25, verbose=1, interval=0.05, stateful_metrics=None):
        self.target = target
        self.width = width
        self.verbose = verbose
        self.interval = interval
        self.stateful_metrics = set(stateful_metrics) if stateful_metrics else set()
        out = sys.stdout
        self._dynamic_display = (hasattr(out, "isatty") and out.isatty()) or "ipykernel" in sys.modules or "posix" in sys.modules
        self._total_width = 0
        self._seen_so_far = 0
        self._values = {}          # name -> _Mean, or the raw value of a stateful metric
        self._values_order = []
        self._start = time.time()
        self._last_update = 0

    # ---- bookkeeping
    def _absorb(self, current, values):
        step = current - self._seen_so_far
        for name, v in values or []:
            if name not in self._values_order:
                self._values_order.append(name)
            if name in self.stateful_metrics:
                self._values[name] = v
            else:
                self._values.setdefault(name, _Mean()).push(v, step)
        self._seen_so_far = current

    def _bar(self, current):
        if self.target is None:
            return "%7d/Unknown" % current
        digits = int(math.floor(math.log10(self.target))) + 1
        filled = int(self.width * float(current) / self.target)
        body = ""
        if filled > 0:
            body = "=" * (filled - 1) + (">" if current < self.target else "=")
        return ("%" + str(digits) + "d/%d [") % (current, self.target) + body + "." * (self.width - filled) + "]"

    def _metrics(self, signed):
        out = ""
        for name in self._values_order:
            v = self._values[name]
            out += " - %s:" % name
            out += _fmt_avg(v.value(), signed) if isinstance(v, _Mean) else " %s" % v
        return out

    # ---- the public pair
    def update(self, current, values=None):
        self._absorb(current, values)
        now = time.time()
        elapsed = now - self._start
        write = sys.stdout.write
        if self.verbose == 1:
            unfinished = self.target is not None and current < self.target
            if now - self._last_update < self.interval and unfinished:
                return
            previous = self._total_width
            write("\b" * previous + "\r" if self._dynamic_display else "\n")
            bar = self._bar(current)
            write(bar)
            per_unit = elapsed / current if current else 0
            if unfinished:
                info = " - ETA: %s" % _fmt_seconds(per_unit * (self.target - current))
            else:
                info = " - %.0fs" % elapsed + _fmt_rate(per_unit)
            info += self._metrics(signed=True)
            self._total_width = len(bar) + len(info)
            if previous > self._total_width:
                info += " " * (previous - self._total_width)
            if self.target is not None and current >= self.target:
                info += "\n"
            write(info)
            sys.stdout.flush()
        elif self.verbose == 2 and (self.target is None or current >= self.target):
            write(" - %.0fs" % elapsed + self._metrics(signed=False) + "\n")
            sys.stdout.flush()
        self._last_update = now
        return self._values

    def add(self, n, values=None):
        return self.update(self._seen_so_far + n, values)
