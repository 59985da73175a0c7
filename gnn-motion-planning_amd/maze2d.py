"""2-D maze environment (host side): the build's counterpart of the parts of the reference's
``MazeEnv`` (environment/maze_env.py, dim = 2) that the GNN planner touches.  Pure numpy; collision
checking stays on the host CPU (north_star).  Needed so the planner counterpart (planner.py) can be
exercised without the reference tree.

Behaviour restated:
  * 15 x 15 occupancy grid over [-1, 1]^2; cell of a point = trunc((x + 1) * w / 2), top index clipped
    (maze_env.py:249-252); obstacles = occupied cells as (i, j) / w - 0.5 (maze_env.py:73-79).
  * a configuration query counts into ``collision_check_count`` only when the point is inside the
    bounds (maze_env.py:281-288).
  * an edge query = both end points, then recursive midpoint bisection while the end cells are more
    than one grid step apart AND the end points are more than RRT_EPS apart in L1
    (maze_env.py:304-326); evaluation is short-circuit, left half first.
  * goal test: within RRT_EPS of the goal and collision-free (maze_env.py:173-178).
  * rejection sampling: uniform in the bounds, one query per draw, rejected draws are returned as
    "collided" samples (maze_env.py:85-100); draws come from the global numpy RNG, two per sample.
"""
import numpy as np

RRT_EPS = 5e-2                       # environment/env_config.py:3
LIMITS = np.array([1.0, 1.0])        # environment/env_config.py:5 (first two entries)


class AttemptStream:
    """Look-ahead over the global numpy RNG for a SEQUENCE of problems sampled back to back: attempts are drawn in
    large blocks and handed out in order, so consumers see exactly the values one-by-one ``uniform_sample`` calls
    would have produced; :meth:`close` puts the global generator into the state it would have had (rewind to the
    start, advance by the attempts actually consumed).  Saves the per-problem get_state / set_state round trips
    (~0.2 ms each) of :meth:`Maze2D.sample_n_points_arrays`."""

    def __init__(self, block=1 << 15):
        self._state0 = np.random.get_state()
        self._block = block
        self._buf = np.zeros((0, 2))
        self._pos = 0
        self.consumed = 0

    def peek(self, m):
        """The next m attempts [m, 2] (not yet consumed)."""
        while self._buf.shape[0] - self._pos < m:
            fresh = np.random.uniform(-LIMITS, LIMITS, (max(self._block, m), 2))
            self._buf = np.concatenate((self._buf[self._pos:], fresh))
            self._pos = 0
        return self._buf[self._pos:self._pos + m]

    def consume(self, m):
        self._pos += m
        self.consumed += m

    def close(self):
        np.random.set_state(self._state0)
        left = self.consumed
        while left > 0:                                       # two doubles per attempt, same order as the draws above
            step = min(left, 1 << 20)
            np.random.uniform(-LIMITS, LIMITS, (step, 2))
            left -= step


class Maze2D:
    RRT_EPS = RRT_EPS

    def __init__(self, maps, init_states, goal_states):
        self.dim = 2
        self.config_dim = 2
        self.maps = np.asarray(maps)
        self.init_states = np.asarray(init_states)
        self.goal_states = np.asarray(goal_states)
        self.size = self.maps.shape[0]
        self.width = self.maps.shape[1]
        self.bound = (-1, -1, 1, 1)
        self.collision_check_count = 0
        self.episode_i = 0

    @classmethod
    def from_npz(cls, path):
        with np.load(path) as f:
            return cls(f['maps'], f['init_states'], f['goal_states'])

    def __str__(self):
        return 'maze2'

    def init_new_problem(self, index=None):
        if index is None:
            index = self.episode_i
        self.map = self.maps[index]
        self.width = self.map.shape[0]
        self.init_state = self.init_states[index]
        self.goal_state = self.goal_states[index]
        self.episode_i = (self.episode_i + 1) % self.size
        occ = np.argwhere(self.map == 1)                     # row-major (i, j) order
        self.obstacles = occ / self.map.shape[0] - 0.5
        self.collision_check_count = 0
        return {'map': self.map, 'init_state': self.init_state, 'goal_state': self.goal_state}

    # ------------------------------------------------------------------ sampling
    def uniform_sample(self):
        return np.random.uniform(-LIMITS, LIMITS, (1, 2)).reshape(-1)

    def sample_n_points(self, n, need_negative=False):
        free, rejected = [], []
        for _ in range(n):
            while True:
                s = self.uniform_sample()
                if self._state_fp(s):
                    free.append(s)
                    break
                if need_negative:
                    rejected.append(s)
        return (free, rejected) if need_negative else free

    def sample_n_points_fast(self, n, need_negative=False):
        """List form of :meth:`sample_n_points_arrays` (same return type as :meth:`sample_n_points`)."""
        free, rej = self.sample_n_points_arrays(n)
        return (list(free), list(rej)) if need_negative else list(free)

    def sample_n_points_arrays(self, n):
        """Same samples, same collision-check count and same final state of the global numpy RNG as
        :meth:`sample_n_points`, but vectorised: draws are made in blocks, classified with one grid lookup,
        and the generator is then rewound and advanced by exactly the number of draws the one-by-one loop
        would have consumed (two doubles per attempt).  Returns (free [n, 2], rejected [m, 2]) float64."""
        state = np.random.get_state()
        block = max(2 * n, 64)
        while True:
            pts = np.random.uniform(-LIMITS, LIMITS, (block, 2))
            w = self.width
            cells = ((pts + 1.0) * w / 2.0).astype(int)
            cells[cells > w - 1] = w - 1
            free_mask = self.map[cells[:, 0], cells[:, 1]] == 0
            idx = np.flatnonzero(free_mask)
            if idx.size >= n:
                used = int(idx[n - 1]) + 1
                break
            np.random.set_state(state)
            block *= 2
        np.random.set_state(state)
        pts = np.random.uniform(-LIMITS, LIMITS, (used, 2))          # consume exactly `used` attempts
        free_mask = free_mask[:used]
        self.collision_check_count += used
        return pts[free_mask], pts[~free_mask]

    def sample_n_points_stream(self, stream, n):
        """:meth:`sample_n_points_arrays` on an :class:`AttemptStream` shared by consecutive problems."""
        w = self.width
        m = max(2 * n, 64)
        while True:
            pts = stream.peek(m)
            cells = ((pts + 1.0) * w / 2.0).astype(int)
            cells[cells > w - 1] = w - 1
            free_mask = self.map[cells[:, 0], cells[:, 1]] == 0
            idx = np.flatnonzero(free_mask)
            if idx.size >= n:
                used = int(idx[n - 1]) + 1
                break
            m *= 2
        pts, free_mask = pts[:used], free_mask[:used]
        stream.consume(used)
        self.collision_check_count += used
        return pts[free_mask], pts[~free_mask]

    # ------------------------------------------------------------------ geometry
    def distance(self, a, b):
        d = np.abs(b - a)
        return np.sqrt(np.sum(d.reshape(1, -1) ** 2, axis=-1))

    def interpolate(self, a, b, ratio):
        return a + (b - a) * ratio

    def in_goal_region(self, state):
        return bool(self.distance(state, self.goal_state) < RRT_EPS and self._state_fp(state))

    # ------------------------------------------------------------------ collision checks
    def _cell(self, state):
        w = self.width
        c = ((np.array(state)[:2].flatten() + 1.0) * w / 2.0).astype(int)
        c[c > w - 1] = w - 1
        return c

    def _valid(self, state):
        return bool((state >= -LIMITS[:state.size]).all() and (state <= LIMITS[:state.size]).all())

    def _state_fp(self, state):
        if not self._valid(state):
            return False
        self.collision_check_count += 1
        return bool(self.map[tuple(self._cell(state))] == 0)

    def _segment_fp(self, left, right):
        lc, rc = self._cell(left), self._cell(right)
        if np.sum(np.abs(lc - rc)) > 1 and np.sum(np.abs(left - right)) > RRT_EPS:
            mid = (left + right) / 2.0
            if not self._state_fp(mid):
                return False
            return self._segment_fp(left, mid) and self._segment_fp(mid, right)
        return True

    def _edge_fp(self, a, b):
        if not self._valid(a) or not self._valid(b):
            return False
        if not self._state_fp(a) or not self._state_fp(b):
            return False
        return self._segment_fp(a, b)


# --------------------------------------------------------------------------------------------------------------------
# 3-DoF maze: a stick of length 0.2 at (x, y) with orientation coordinate z in [-0.4, 0.4] (theta = z / 0.4 * pi),
# the reference's MazeEnv(dim=3) (environment/maze_env.py with environment/env_config.py:4-5) as far as the GNN planner
# touches it.  Restated behaviour:
#   * a configuration is free iff both stick ends are inside the bounds and in free cells and the 2-D bisection between
#     them finds no occupied cell (maze_env.py:289-302); every in-bounds point query counts one collision check,
#     evaluation is short-circuit (end a, end b, then midpoints, left half first);
#   * an edge query = both configurations, then K - 1 = int(d / 0.015) - 1 interpolated configurations (orientation
#     wrapped across +-0.4), each checked as the 2-D EDGE between its stick ends (maze_env.py:330-347);
#   * distance / goal test wrap the third coordinate (maze_env.py:138-149,173-178);
#   * numpy dtype promotion is part of the behaviour (NumPy >= 2, NEP 50): node rows arrive as float32, the stick ends are
#     float64 (float32 coordinate / float64 limit), interpolation happens in float32.
# --------------------------------------------------------------------------------------------------------------------
STICK_LENGTH = 1.5 * 2 / 15                   # environment/env_config.py:4
LIMITS3 = np.array([1., 1., 8. * RRT_EPS])   # environment/env_config.py:5


class Maze3D(Maze2D):
    def __init__(self, maps, init_states, goal_states):
        super().__init__(maps, init_states, goal_states)
        self.dim = 3
        self.config_dim = 3
        self.bound = (-1, -1, -0.4, 1, 1, 0.4)

    def __str__(self):
        return 'maze3'

    def uniform_sample(self):
        return np.random.uniform(-LIMITS3, LIMITS3, (1, 3)).reshape(-1)

    # sampling: the one-by-one loop of Maze2D.sample_n_points applies (it calls self._state_fp)
    def sample_n_points_arrays(self, n):
        free, rej = self.sample_n_points(n, need_negative=True)
        return np.array(free).reshape(-1, 3), np.array(rej).reshape(-1, 3)

    def sample_n_points_stream(self, stream, n):
        raise NotImplementedError('the look-ahead sampler is 2-D only; Maze3D samples one configuration at a time')

    # The orientation coordinate lives on a circle of circumference ORIENT_PERIOD = 2 * LIMITS3[2] (a numpy float64 scalar:
    # under NEP 50 it promotes float32 operands to float64 before the result is stored back into the float32 row, which
    # is part of the recorded behaviour).  One helper does the wrap for displacements and for configurations alike.
    ORIENT_HALF = LIMITS3[2]
    ORIENT_PERIOD = 2 * LIMITS3[2]

    @classmethod
    def _wrap_orientation(cls, vec):
        """In place: bring vec[2] back into [-ORIENT_HALF, ORIENT_HALF] by one period (maze_env.py:158-170 applies this
        rule to the displacement and to the interpolated configuration)."""
        z = vec[2]
        if np.abs(z) > cls.ORIENT_HALF:
            vec[2] = z - cls.ORIENT_PERIOD if z > 0 else z + cls.ORIENT_PERIOD
        return vec

    def distance(self, a, b):
        """Euclidean distance with the orientation gap measured the short way round (maze_env.py:138-149)."""
        gap = np.atleast_2d(np.abs(b - a))
        gap[:, 2] = np.minimum(gap[:, 2], np.abs(gap[:, 2] - self.ORIENT_PERIOD))
        return np.sqrt(np.sum(gap ** 2, axis=-1))

    def interpolate(self, a, b, ratio):
        """Point at `ratio` of the way from a to b along the short orientation arc (maze_env.py:151-170)."""
        return self._wrap_orientation(a + self._wrap_orientation(b - a) * ratio)

    @staticmethod
    def _ends(coord):
        theta = coord[2] / LIMITS3[2] * np.pi
        orient = np.array([np.cos(theta), np.sin(theta)])
        center = np.array(coord[:2])
        return center - STICK_LENGTH / 2. * orient, center + STICK_LENGTH / 2. * orient

    def _valid(self, state):
        return bool((state >= -LIMITS3[:state.size]).all() and (state <= LIMITS3[:state.size]).all())

    def _point_fp(self, p):
        if not self._valid(p):
            return False
        self.collision_check_count += 1
        return bool(self.map[tuple(self._cell(p))] == 0)

    def _state_fp(self, state):
        if state.size == 2:
            return self._point_fp(state)
        if not self._valid(state):
            return False
        a, b = self._ends(state)
        if not self._point_fp(a) or not self._point_fp(b):
            return False
        return self._segment_fp(a, b)

    def _edge_fp(self, a, b):
        if not self._valid(a) or not self._valid(b):
            return False
        if not self._state_fp(a) or not self._state_fp(b):
            return False
        if a.size == 2:
            return self._segment_fp(a, b)
        # K - 1 interior configurations at equal fractions of the (orientation-wrapped) displacement, each checked as the 2-D
        # edge between its stick ends (maze_env.py:330-347).  The fractions are python floats: they leave the float32 rows
        # in float32, as the recorded runs have it.
        step = self._wrap_orientation(b - a)
        K = int((self.distance(a, b) / 0.015)[0])                # float32 array / python float: float32 division
        return all(self._edge_fp(*self._ends(a + (k * 1. / K) * step)) for k in range(1, K))
