"""Curriculum (radius / explore-bonus schedule) -- host-side scalar logic of the reference's
PPOTrainer.update (PPOV2.0/model.py:131-164), plus a batched form for thousands of
asynchronous episodes per iteration.

Aggregation rule at scale: finished episodes enter the 120-episode window in
(iteration, global env index, time) order; with one env it is exactly the reference's order.
"""
from __future__ import annotations

import numpy as np

INITIAL_RADIUS, MIN_RADIUS, RADIUS_DECAY = 50.0, 5.0, 0.9      # config.py:27-29
SUCCESS_THRESHOLD, WINDOW_SIZE = 0.6, 120                      # config.py:30-31
EXPLORE_BONUS, DECAY_FACTOR = 0.6, 0.999                       # config.py:21-22


class Curriculum:
    def __init__(self):
        self.current_radius = INITIAL_RADIUS
        self.explore_bonus = EXPLORE_BONUS
        self.env_radius = INITIAL_RADIUS      # what the environment currently uses (lags by one episode)
        self.env_bonus = EXPLORE_BONUS
        self.success_history = []

    def update(self, success):
        """One finished episode -- model.py:131-164 line for line in meaning (dtype included:
        the bonus becomes np.float64 once a window has been processed, model.py:141-142)."""
        self.env_radius, self.env_bonus = self.current_radius, self.explore_bonus          # :132-133
        self.success_history.append(bool(success))
        if len(self.success_history) > WINDOW_SIZE:
            self.success_history.pop(0)
        full = len(self.success_history) >= WINDOW_SIZE
        if full:
            rate = np.mean(self.success_history[-WINDOW_SIZE:])
            self.explore_bonus *= DECAY_FACTOR ** (1 + rate)                               # :140-142
        self.explore_bonus = max(self.explore_bonus, 0.1)                                  # :144
        if full:
            if rate > SUCCESS_THRESHOLD:                                                   # :148-152
                self.current_radius = max(MIN_RADIUS, self.current_radius *
                                          (RADIUS_DECAY ** (2 + 3 * (rate - SUCCESS_THRESHOLD))))
            elif rate < 0.25:                                                              # :153-157
                self.current_radius = min(INITIAL_RADIUS, self.current_radius * 1.1)
            if abs(self.current_radius - self.env_radius) > 5:                             # :160-161
                self.current_radius = self.env_radius + 5 * np.sign(self.current_radius - self.env_radius)
            self.success_history = []                                                      # :164

    def update_many(self, successes):
        """Feed a sequence of finished episodes; identical to calling update() on each in order,
        but only touches Python once per 120-episode window."""
        s = np.asarray(successes, dtype=bool).reshape(-1)
        i = 0
        while i < s.size:
            room = WINDOW_SIZE - len(self.success_history)
            if room > 1:
                k = min(room - 1, s.size - i)          # these cannot complete the window
                self.success_history.extend(s[i:i + k].tolist())
                i += k
                self.env_radius, self.env_bonus = self.current_radius, self.explore_bonus
                self.explore_bonus = max(self.explore_bonus, 0.1)
                continue
            self.update(bool(s[i]))
            i += 1
