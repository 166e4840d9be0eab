"""Rank-local time-grid layout in O(n): which points a rank owns on every level, its ghost point, the C/F index
sets and the exchange flags.

Restates ``Mgrit.setup_points_and_comm_info`` / ``split_into`` / ``split_points`` (reference
src/pymgrit/core/mgrit.py:728-838) without its O(n_local^2) broadcasting temporaries (4.3 GB at nt=65537, P=1).
Every array and flag must equal the reference bit for bit (tests/test_layout.py vs tests/golden/layout.json);
``index_local_f`` is emitted in the canonical order "runs of consecutive F-points reversed, ascending inside a run"
(the reference's exact order depends on CPython set iteration, SURVEY App. A, and does not affect results).
"""
from dataclasses import dataclass

import numpy as np

NO_RANK = -99  # the reference's "nobody" marker for send_to / get_from (mgrit.py:818-819)


def split_into(number_points: int, number_processes: int) -> np.ndarray:
    """Block sizes of an even split: the first ``n % P`` ranks get one extra point (mgrit.py:829-838)."""
    base, extra = divmod(int(number_points), int(number_processes))
    return np.array([base + 1] * extra + [base] * (number_processes - extra))


def split_points(length: int, size: int, rank: int):
    """(block size, index of first point) of ``rank`` (mgrit.py:728-740)."""
    blocks = split_into(length, size)
    return blocks[rank], (np.sum(blocks[:rank]) if blocks[rank] > 0 else 0)


@dataclass
class LevelLayout:
    t_local: np.ndarray          # Mgrit.t[lvl]: ghost (if any) + owned time values
    cpts: np.ndarray             # global indices of owned C-points
    index_local: np.ndarray      # local slots of owned points
    index_local_c: np.ndarray
    index_local_f: np.ndarray
    comm_front: bool
    comm_back: bool
    first_is_c_point: bool
    first_is_f_point: bool
    last_is_c_point: bool
    last_is_f_point: bool
    send_to: int
    get_from: int
    ghost: int                   # 1 when slot 0 is a ghost point
    first_owned: int             # global index of the first owned point (-1: none)
    is_c_local: np.ndarray       # bool per local slot (ghost included)
    int_start: float
    int_stop: float


def member_mask(values, grid) -> np.ndarray:
    """np.isin(values, grid) -- exact float membership -- for the usual case of a strictly ascending ``grid`` by binary search
    (isin sorts the concatenation of both arrays: 1 ms per call at 65537 points, six calls per constructor)"""
    values, grid = np.asarray(values), np.asarray(grid)
    if grid.ndim != 1 or grid.size < 2 or not bool((grid[1:] > grid[:-1]).all()):
        return np.isin(values, grid)
    at = np.minimum(np.searchsorted(grid, values), grid.size - 1)
    return grid[at] == values


def global_c_mask(t_lvl: np.ndarray, t_next) -> np.ndarray:
    """C-points = exact float membership in the next coarser grid; every point on the coarsest level
    (mgrit.py:767-770)."""
    if t_next is None:
        return np.ones(len(t_lvl), dtype=bool)
    return member_mask(t_lvl, t_next)


def compute_layout(global_t, lvl: int, rank: int, size: int) -> LevelLayout:
    t0, t = np.asarray(global_t[0]), np.asarray(global_t[lvl])
    n_pts = len(t)
    blocks = split_into(len(t0), size)
    first0 = int(np.sum(blocks[:rank]))
    int_start, int_stop = t0[first0], t0[first0 + int(blocks[rank]) - 1]
    if lvl == 0:
        a, z = first0, first0 + int(blocks[rank]) - 1
    else:  # ownership by time value (mgrit.py:764)
        a = int(np.searchsorted(t, int_start, side='left'))
        z = int(np.searchsorted(t, int_stop, side='right')) - 1
    n_owned = max(z - a + 1, 0)
    is_c = global_c_mask(t, global_t[lvl + 1] if lvl + 1 < len(global_t) else None)
    ghost = 1 if (rank != 0 and n_owned > 0) else 0

    owned = np.arange(a, a + n_owned)
    local_global = np.concatenate(([a - 1], owned)).astype(int) if ghost else owned.astype(int)
    index_local = ghost + np.arange(n_owned)
    own_c = is_c[a:a + n_owned] if n_owned else np.zeros(0, dtype=bool)
    index_local_c = index_local[own_c]
    cpts = owned[own_c]
    # F runs, reversed run order
    f_slots = index_local[~own_c]
    if f_slots.size:
        cuts = np.nonzero(np.diff(f_slots) != 1)[0] + 1
        # runs in reversed order, ascending inside a run -- np.concatenate(np.split(f_slots, cuts)[::-1]) without its Python
        # list of 16384 pieces: every element goes to (start of its run in the reversed sequence) + (its place in the run)
        first = np.concatenate(([0], cuts))
        length = np.diff(np.concatenate((first, [f_slots.size])))
        run = np.repeat(np.arange(first.size), length)
        start_rev = np.concatenate((np.cumsum(length[::-1])[::-1][1:], [0]))     # elements of the runs AFTER run r
        index_local_f = np.empty_like(f_slots)
        index_local_f[start_rev[run] + (np.arange(f_slots.size) - first[run])] = f_slots
    else:
        index_local_f = np.zeros(0)  # the reference yields an empty float array here

    def is_f(i):
        return 0 <= i < n_pts and not is_c[i]

    def is_cp(i):
        return 0 <= i < n_pts and bool(is_c[i])

    flags = dict(comm_front=False, comm_back=False, first_is_c_point=False, first_is_f_point=False,
                 last_is_c_point=False, last_is_f_point=False)
    if n_owned:
        own_f = owned[~own_c]
        if own_f.size:
            flags['comm_front'] = is_f(int(own_f[0]) - 1)
            flags['comm_back'] = is_f(int(own_f[-1]) + 1)
        flags['first_is_c_point'] = bool(is_c[a]) and a != 0 and is_f(a - 1)
        flags['first_is_f_point'] = (not is_c[a]) and is_cp(a - 1)
        flags['last_is_c_point'] = bool(is_c[z]) and z != n_pts - 1 and is_f(z + 1)
        flags['last_is_f_point'] = (not is_c[z]) and z != n_pts - 1 and is_cp(z + 1)

    send_to = get_from = NO_RANK
    if local_global.size:
        ends = t0[np.cumsum(blocks) - 1]
        if z != n_pts - 1:
            send_to = int(np.searchsorted(ends, t[z + 1]))
        if ghost or t[local_global[0]] != t0[0]:
            get_from = int(np.searchsorted(ends, t[local_global[0]]))

    return LevelLayout(t_local=t[local_global], cpts=cpts, index_local=index_local, index_local_c=index_local_c,
                       index_local_f=index_local_f, send_to=send_to, get_from=get_from, ghost=ghost,
                       first_owned=a if n_owned else -1, is_c_local=is_c[local_global] if local_global.size else
                       np.zeros(0, dtype=bool), int_start=int_start, int_stop=int_stop, **flags)


class IndexArray:
    """A sequence of equal-length tuples of ints (``arr`` of shape [N, k]) or of ints ([N]) held as ONE int64 array: the run /
    pair / triple / interval lists of a level (16384 six-tuples per list at BASELINE config 3) are built by array arithmetic and
    handed to the device library column by column without ever becoming Python objects -- as lists of tuples they cost ~20 ms of
    the first iteration of ``Mgrit.solve()``. Read like a list of tuples where somebody does (tests, the plugin backend, the
    sharded schedules): len, truth, iteration, integer and slice indexing, comparison with a list, ``+``; like a list built by
    ``Mgrit._cached`` it can carry the backend's device handles as attributes."""

    def __init__(self, arr, width=None):
        arr = np.asarray(arr, dtype=np.int64)
        if width is not None:
            arr = arr.reshape(-1, width)
        self.arr = np.ascontiguousarray(arr)
        self._items = None

    def columns(self, dtype=np.int32):
        """the k columns as contiguous arrays (one for a sequence of ints)"""
        a = self.arr if self.arr.ndim == 2 else self.arr.reshape(-1, 1)
        return [np.ascontiguousarray(a[:, c], dtype=dtype) for c in range(a.shape[1])]

    def tolist(self):
        if self._items is None:
            self._items = [tuple(r) for r in self.arr.tolist()] if self.arr.ndim == 2 else self.arr.tolist()
        return self._items

    def __len__(self):
        return int(self.arr.shape[0])

    def __bool__(self):
        return self.arr.shape[0] > 0

    def __iter__(self):
        return iter(self.tolist())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return IndexArray(self.arr[i])
        row = self.arr[i]
        return tuple(row.tolist()) if self.arr.ndim == 2 else int(row)

    def __eq__(self, other):
        if isinstance(other, IndexArray):
            return self.arr.shape == other.arr.shape and bool(np.array_equal(self.arr, other.arr))
        if isinstance(other, (list, tuple)):
            return self.tolist() == [tuple(x) if isinstance(x, (list, tuple)) else x for x in other]
        return NotImplemented

    def __ne__(self, other):
        eq = self.__eq__(other)
        return eq if eq is NotImplemented else not eq

    __hash__ = None

    def __add__(self, other):
        return self.tolist() + list(other)

    def __radd__(self, other):
        return list(other) + self.tolist()

    def __repr__(self):
        return f"IndexArray({self.tolist()!r})"


def as_index_array(items, width):
    """``items`` (IndexArray, list of tuples, empty list) as an [N, width] int64 array"""
    if isinstance(items, IndexArray):
        return items.arr if items.arr.ndim == 2 else items.arr.reshape(-1, 1)
    if not len(items):
        return np.zeros((0, width), dtype=np.int64)
    return np.asarray(items, dtype=np.int64).reshape(len(items), width)


def consecutive_runs(slots):
    """(start, length) of the maximal runs of consecutive integers in the ascending array ``slots``, as an IndexArray."""
    slots = slots.arr if isinstance(slots, IndexArray) else np.asarray(slots, dtype=np.int64)
    if slots.size == 0:
        return IndexArray(np.zeros((0, 2), dtype=np.int64))
    cuts = np.nonzero(np.diff(slots) != 1)[0] + 1
    first = np.concatenate(([0], cuts))
    length = np.diff(np.concatenate((first, [slots.size])))
    return IndexArray(np.stack((slots[first], length), axis=1))
