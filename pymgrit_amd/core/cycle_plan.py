"""Cycle plan: one MGRIT cycle as a static graph of (sweep, level, block of time points) launches.

The reference walks a cycle level by level (``Mgrit.iteration``, reference src/pymgrit/core/mgrit.py:261-290): every sweep
covers ALL local time points before the next one starts, and the sequential coarsest-level solve (``forward_solve``,
mgrit.py:459-486) sits in the middle with nothing beside it -- on one MI355X that is a latency-bound kernel on 16 of the
256 CUs for a quarter of the cycle. The data dependencies are much weaker than the program order: a sweep over one block
of time points only needs the same block (and the last point of the block before it) of the sweeps in front of it. So the
cycle is RECORDED once -- the unchanged ``Mgrit.iteration`` runs against a recorder that splits every backend call into
blocks of time points and notes which (array, level, block) cells each part reads and writes --, the parts become the
nodes of a dependency graph (read-after-write, write-after-read and write-after-write edges, all from the recorded order,
so ANY topological order computes bit for bit what the program order computes), and a list scheduler lays the nodes out
on two in-order streams: the bandwidth-bound sweeps on one, the coarsest-level chain parts on the other. Replaying the
plan launches the same kernels with the same arguments in the new order; the chain of block k runs beside the sweeps of
the blocks after it (down) and before it (up).

The recorder works at the backend interface (``relax``, ``fas_fused``, ``ec_relax`` ...), so it is the same for the HIP
engine and for the plugin backend (on the CPU the plan is executed serially in its scheduled order: that is how the
dependency rules are tested without a GPU, tests/test_cycle_plan.py).
"""
import numpy as np

from pymgrit_amd.core.options import options


class PlanUnsupported(Exception):
    """the cycle contains a backend call the recorder has no dependency rule for: run it in program order"""


class IndexList(list):
    """list that can carry the backend's device-side handle as an attribute"""


class Node:
    __slots__ = ("idx", "name", "lvl", "chunk", "stream", "fn", "reads", "writes", "cost", "preds", "succs", "start",
                 "finish", "event", "cross_preds", "needs_event")

    def __init__(self, idx, name, lvl, chunk, stream, fn, reads, writes, cost):
        self.idx, self.name, self.lvl, self.chunk, self.stream, self.fn = idx, name, lvl, chunk, stream, fn
        self.reads, self.writes, self.cost = frozenset(reads), frozenset(writes), float(cost)
        self.preds, self.succs = set(), set()
        self.start = self.finish = 0.0
        self.event, self.cross_preds, self.needs_event = None, (), False

    def __repr__(self):
        return f"<{self.idx}:{self.name} L{self.lvl} b{self.chunk} {self.stream}>"


def block_maps(t_levels, n_blocks):
    """block index of every local slot of every level. Blocks are cut at points of the COARSEST level (so a block holds whole
    coarse intervals on every level): block k = the time points in (tau_k, tau_{k+1}], point 0 in block 0 -- a block ends on
    a point that is a C-point on every level, like a rank's share of the time grid does."""
    tc = np.asarray(t_levels[-1], dtype=np.float64)
    n_c = len(tc)
    if n_blocks <= 1 or n_c < 3:
        return 1, [np.zeros(len(t), dtype=np.int64) for t in t_levels]
    stride = max(1, -(-(n_c - 1) // n_blocks))
    cuts = tc[stride:n_c - 1:stride]
    return len(cuts) + 1, [np.searchsorted(cuts, np.asarray(t, dtype=np.float64), side='left').astype(np.int64) for t in t_levels]


class Recorder:
    """Stands in for the sweep backend while ``Mgrit.iteration`` is walked once: nothing is executed, every call is split
    into per-block parts with their footprints. Footprints follow what the backends really touch (backend_plugin.py and the
    kernels of csrc/mgrit_hip.hip do the same reads and writes per item)."""

    def __init__(self, mg, backend, n_blocks):
        self.mg, self.real = mg, backend
        self.K, self.block_of = block_maps(mg.t, n_blocks)
        # blocks at the front of the time grid whose way up goes into one launch (0 / 1: none): half of them when the cycle has
        # five or more blocks (measured on config 3 with six: three 8.24 ms, two 8.35, four 9.2, none 8.46-8.60)
        self.up_merge = int(options.plan_up_merge if options.plan_up_merge is not None else (self.K // 2 if self.K >= 5 else 0))
        self.nodes = []
        self.sends, self.recvs = {}, {}     # (peer rank, channel) -> messages of the cycle on that link (several ranks)
        self.host_after = []        # host-only bookkeeping calls of the cycle: run after every execution of the plan
        self._last_write, self._readers = {}, {}
        self.row_bytes = [8.0 * getattr(backend, "ld", [1] * mg.lvl_max)[lvl] if hasattr(backend, "ld") else 8.0
                          for lvl in range(mg.lvl_max)]

    # anything that is not a sweep (capability queries like can_fuse_ec, plan_blocks; plain attributes) is answered by the real
    # backend; a SWEEP the recorder has no dependency rule for ends the recording (the cycle then runs in program order)
    PASS_THROUGH = ("can_", "plan_")
    PASS_NAMES = frozenset(("device_links", "chain_handover", "chain_state", "_cycle_pre", "ld", "n", "desc", "name", "block_r",
                            "block_sharded", "block_uh"))

    def __getattr__(self, name):
        if name.startswith(self.PASS_THROUGH) or name in self.PASS_NAMES:
            return getattr(self.real, name)
        if name.startswith("__"):
            raise AttributeError(name)
        raise PlanUnsupported(name)

    # ---- graph construction -------------------------------------------------------------------------------------------
    def _add(self, name, lvl, chunk, fn, reads, writes, rows, stream="sweep", cost=None):
        if cost is None:
            cost = 5e-6 + rows * self.row_bytes[lvl] / 5.0e12
        node = Node(len(self.nodes), name, lvl, chunk, stream, fn, reads, writes, cost)
        for res in node.reads | node.writes:
            w = self._last_write.get(res)
            if w is not None:
                node.preds.add(w)                      # read-after-write, write-after-write
        for res in node.writes:
            node.preds.update(self._readers.get(res, ()))   # write-after-read
        node.preds.discard(node.idx)
        for res in node.writes:
            self._last_write[res] = node.idx
            self._readers[res] = set()
        for res in node.reads - node.writes:
            self._readers.setdefault(res, set()).add(node.idx)
        for p in node.preds:
            self.nodes[p].succs.add(node.idx)
        self.nodes.append(node)
        return node

    def _cells(self, arr, lvl, slots):
        slots = np.asarray(slots, dtype=np.int64)
        if slots.size == 0:
            return set()
        return {(arr, lvl, int(b)) for b in np.unique(self.block_of[lvl][slots])}

    def _one_block(self, lvl):
        """the whole level is one block of the plan (the default since the coarsest-level solves are time-parallel): nothing to
        cut, the index lists of the sweeps go through as they are"""
        memo = self.__dict__.setdefault("_one_block_memo", {})
        if lvl not in memo:
            bo = self.block_of[lvl]
            memo[lvl] = bool(bo.size) and bool((bo == bo[0]).all())
        return memo[lvl]

    def _by_block(self, lvl, items, key_slot):
        """items grouped by the block of item[key_slot] on level lvl, ascending block order; [(block, IndexList)]"""
        if not items:
            return []
        if self._one_block(lvl):
            return [(int(self.block_of[lvl][0]), items if hasattr(items, "__dict__") else IndexList(items))]
        blocks = self.block_of[lvl][np.asarray([it[key_slot] for it in items], dtype=np.int64)]
        out = []
        for b in np.unique(blocks):
            part = IndexList(items[i] for i in np.nonzero(blocks == b)[0])
            out.append((int(b), part))
        return out

    def _by_group(self, lvl, items, key_slot):
        """way up: like _by_block, but the first blocks of the time grid -- whose coarsest-level chain parts have long finished
        when the way down of the last block has (the chain starts behind the first block's way down and is faster than the
        sweeps that feed it) -- go into ONE launch: a launch over four sixths of a level runs at nearly the full-width rate,
        four one-sixth launches do not. [(first block, IndexList, member blocks)]"""
        parts = self._by_block(lvl, items, key_slot)
        merge = self.up_merge
        if merge <= 1 or len(parts) <= 1:
            return [(b, part, (b,)) for b, part in parts]
        head = [(b, part) for b, part in parts if b < merge]
        out = []
        if head:
            out.append((head[0][0], IndexList(it for _, part in head for it in part), tuple(b for b, _ in head)))
        out.extend((b, part, (b,)) for b, part in parts if b >= merge)
        return out

    def _split_runs(self, lvl, runs):
        """runs cut at block borders (a run continued in the next block reads the last point of its first part)"""
        if self._one_block(lvl):
            return runs
        out = []
        bo = self.block_of[lvl]
        for st, ln in runs:
            a = st
            while a < st + ln:
                b = bo[a]
                z = a
                while z + 1 < st + ln and bo[z + 1] == b:
                    z += 1
                out.append((int(a), int(z - a + 1)))
                a = z + 1
        return out

    # ---- sweeps ---------------------------------------------------------------------------------------------------------
    def relax(self, lvl, runs, mode):
        if not runs:
            return
        real, mg = self.real, self.mg
        chain = mode == 'CHAIN'
        if chain and getattr(real, "block_r", {}).get(lvl):
            # the time-parallel forward solve takes the level as a whole: ONE node between the way down of the last
            # block and the way up of the first
            blocks = {int(b) for b in np.unique(self.block_of[lvl])}
            n_pts = int(sum(r[1] for r in runs))
            self._add("chain", lvl, 0, lambda: real.relax(lvl, runs, 'CHAIN'),
                      {("u", lvl, b) for b in blocks} | {("g", lvl, b) for b in blocks},
                      {("u", lvl, b) for b in blocks} | {("chain", lvl, 0)}, 4 * n_pts)
            return
        parts = self._by_block(lvl, self._split_runs(lvl, runs), 0)
        first = True
        for b, part in parts:
            st = np.asarray([r[0] for r in part], dtype=np.int64)
            n_pts = int(sum(r[1] for r in part))
            reads = self._cells("v" if mode == 'FC' else "u", lvl, st - 1)   # mode FC starts its runs from v
            if lvl > 0:
                reads |= {("g", lvl, b)}
            if mode == 'C' and mg.weight_c != 1.0:
                reads |= {("u", lvl, b)}
            if chain:
                resume = (not first) and getattr(real, "chain_state", {}).get(lvl) is not None
                fn = (lambda p=part, r=resume: real.relax_chain_part(lvl, p, r)) if hasattr(real, "relax_chain_part") else \
                    (lambda p=part: real.relax(lvl, p, 'CHAIN'))
                # ("chain", lvl): the parts of one forward solve share the engine's granules and hand-over state
                self._add("chain", lvl, b, fn, reads, {("u", lvl, b), ("chain", lvl, 0)}, n_pts, stream="chain",
                          cost=5e-6 + n_pts * 2.0e-6)
            else:
                rows = len(part) + n_pts * (2 if lvl > 0 else 1)
                self._add("relax_" + mode, lvl, b, lambda p=part: real.relax(lvl, p, mode), reads, {("u", lvl, b)}, rows)
            first = False

    def block_solve(self, lvl, phases):
        """a phase of the time-parallel forward solve on a rank of a sharded level (backend_hip.block_solve): the phases and the
        hand-overs between them keep their recorded order (they all write the level's "chain" cell)"""
        real = self.real
        blocks = {int(b) for b in np.unique(self.block_of[lvl])}
        cells = {("u", lvl, b) for b in blocks}
        self._add("block_solve", lvl, 0, lambda: real.block_solve(lvl, phases), cells | {("g", lvl, b) for b in blocks},
                  cells | {("chain", lvl, 0)}, 2 * len(self.block_of[lvl]), stream="chain")

    # ---- exchange points (several ranks; backend_hip.exchange: stream operations of the engine) ------------------------------
    # A send reads the row it sends, a receive writes the ghost row; the messages of one link keep their recorded order (the
    # two owners of a link match messages by order): a pseudo-cell per link and direction that every message "writes".
    # The hand-over of the coarsest level's forward solve (op 5) runs on the chain's stream, like the chain parts around it.
    def _ordinal(self, table, peer, ch):
        k = table.get((int(peer), ch), 0)
        table[(int(peer), ch)] = k + 1
        return k

    def exchange(self, lvl, op, send_idx=None, dest=None, recv_idx=None, src=None, raw=False, ordinals=None):
        from pymgrit_amd.core.comm import CH_CHAIN, CH_SWEEP
        real, ch = self.real, (CH_CHAIN if op == 5 else CH_SWEEP)
        stream = "chain" if op == 5 else "sweep"
        extra = {("chain", lvl, 0)} if op == 5 else set()
        if send_idx is not None:
            k = self._ordinal(self.sends, dest, ch)
            self._add("send", lvl, int(self.block_of[lvl][send_idx]),
                      lambda k=k: real.exchange(lvl, op, send_idx=send_idx, dest=dest, raw=True, ordinals=(k, None)),
                      self._cells("u", lvl, [send_idx]) | extra, {("xs", int(dest), ch)}, 1, stream=stream, cost=8e-6)
        if recv_idx is not None:
            k = self._ordinal(self.recvs, src, ch)
            self._add("recv", lvl, int(self.block_of[lvl][recv_idx]),
                      lambda k=k: real.exchange(lvl, op, recv_idx=recv_idx, src=src, raw=True, ordinals=(None, k)),
                      set(), self._cells("u", lvl, [recv_idx]) | extra | {("xr", int(src), ch)}, 1, stream=stream, cost=8e-6)

    def exchange_staged(self, lvl, op, pair, dest=None, recv_idx=None, src=None, ordinals=None):
        from pymgrit_amd.core.comm import CH_SWEEP
        real = self.real
        fi, co = int(pair[0][0]), int(pair[0][1])
        k = self._ordinal(self.sends, dest, CH_SWEEP)
        self._add("send_corrected", lvl, int(self.block_of[lvl][fi]),
                  lambda k=k: real.exchange_staged(lvl, op, pair, dest=dest, ordinals=(k, None)),
                  self._cells("u", lvl, [fi]) | self._cells("u", lvl + 1, [co]) | self._cells("v", lvl + 1, [co]),
                  {("xs", int(dest), CH_SWEEP), ("stage", lvl, 0)}, 3, cost=1e-5)
        if recv_idx is not None:
            self.exchange(lvl, op, recv_idx=recv_idx, src=src)

    def ec_relax(self, lvl, triples):
        if not triples:
            return
        real = self.real
        for b, part, members in self._by_group(lvl, triples, 0):
            st = np.asarray([t[0] for t in part], dtype=np.int64)
            co = np.asarray([t[2] for t in part], dtype=np.int64)
            n_pts = int(sum(t[1] for t in part))
            front = self._cells("u", lvl, st - 1)
            reads = front | self._cells("u", lvl + 1, co[co >= 0]) | ({("g", lvl, m) for m in members} if lvl > 0 else set())
            writes = {("u", lvl, m) for m in members} | self._cells("u", lvl, (st - 1)[co >= 0])
            self._add("ec_relax", lvl, b, lambda p=part: real.ec_relax(lvl, p), reads, writes,
                      2 * len(part) + int(np.count_nonzero(co >= 0)) + n_pts * (2 if lvl > 0 else 1))

    def error_correction(self, lvl, pairs):
        real = self.real
        for b, part in self._by_block(lvl, pairs, 0):
            fi, co = [np.asarray([p[k] for p in part], dtype=np.int64) for k in (0, 1)]
            reads = {("u", lvl, b)} | self._cells("u", lvl + 1, co) | self._cells("v", lvl + 1, co)
            self._add("error_correction", lvl, b, lambda p=part: real.error_correction(lvl, p), reads, {("u", lvl, b)}, 4 * len(part))

    def interpolate(self, lvl, pairs):
        real = self.real
        for b, part in self._by_block(lvl, pairs, 0):
            co = np.asarray([p[1] for p in part], dtype=np.int64)
            self._add("interpolate", lvl, b, lambda p=part: real.interpolate(lvl, p), self._cells("u", lvl + 1, co),
                      {("u", lvl, b)}, 2 * len(part))

    def restrict_u(self, lvl, pairs):
        real = self.real
        for b, part in self._by_block(lvl, pairs, 0):
            co = np.asarray([p[1] for p in part], dtype=np.int64)
            self._add("restrict_u", lvl, b, lambda p=part: real.restrict_u(lvl, p), {("u", lvl, b)},
                      self._cells("u", lvl + 1, co), 2 * len(part))

    def copy_u_to_v(self, lvl):
        real = self.real
        blocks = {int(b) for b in np.unique(self.block_of[lvl])} if len(self.block_of[lvl]) else set()
        self._add("copy_u_to_v", lvl, 0, lambda: real.copy_u_to_v(lvl), {("u", lvl, b) for b in blocks},
                  {("v", lvl, b) for b in blocks}, 2 * len(self.block_of[lvl]))

    def copy_pairs_u_to_v(self, lvl, pairs):
        real = self.real
        for b, part in self._by_block(lvl, pairs, 0):
            co = np.asarray([p[1] for p in part], dtype=np.int64)
            self._add("copy_pairs_u_to_v", lvl, b, lambda p=part: real.copy_pairs_u_to_v(lvl, p), self._cells("u", lvl + 1, co),
                      self._cells("v", lvl + 1, co), 2 * len(part))

    def fas_rhs(self, lvl, pairs):
        real = self.real
        for b, part in self._by_block(lvl, pairs, 0):
            fi, co = [np.asarray([p[k] for p in part], dtype=np.int64) for k in (0, 1)]
            reads = {("u", lvl, b)} | self._cells("u", lvl, fi - 1) | self._cells("v", lvl + 1, co) | \
                self._cells("v", lvl + 1, co - 1) | self._cells("g", lvl + 1, co)
            if lvl > 0:
                reads |= {("g", lvl, b)}
            self._add("fas_rhs", lvl, b, lambda p=part: real.fas_rhs(lvl, p), reads, self._cells("g", lvl + 1, co), 8 * len(part))

    # whole-level passes of the device backend (backend_hip.cf_fas / ec_relax_res): items are intervals
    # (cstart, cend, cstart_coarse, cend_coarse, res_pos); an interval belongs to the block of the C-point it ends on
    def cf_fas(self, lvl, intervals):
        real = self.real
        for b, part in self._by_block(lvl, intervals, 1):
            cs, ce, jcs, jce = _interval_cells(self, lvl, part)
            # reads: the old last F-point of every interval (its own block), and for the first interval of a chunk the F-point
            # in front of its starting C-point or that C-point itself (possibly the block before)
            reads = {("u", lvl, b)} | self._cells("u", lvl, cs) | self._cells("u", lvl, np.maximum(cs - 1, 0))
            writes = {("u", lvl, b)} | self._cells("u", lvl + 1, jce) | self._cells("v", lvl + 1, jce) | self._cells("g", lvl + 1, jce)
            self._add("cf_fas", lvl, b, lambda p=part: real.cf_fas(lvl, p), reads, writes, 7 * len(part))

    def ec_relax_res(self, lvl, intervals):
        real = self.real
        for b, part, members in self._by_group(lvl, intervals, 1):
            cs, ce, jcs, jce = _interval_cells(self, lvl, part)
            reads = self._cells("u", lvl + 1, jce) | self._cells("v", lvl + 1, jce) | self._cells("u", lvl + 1, jcs[jcs >= 0]) | \
                self._cells("v", lvl + 1, jcs[jcs >= 0]) | self._cells("u", lvl, cs[jcs < 0])
            if lvl > 0:
                reads |= {("g", lvl, m) for m in members}
            n_f = int(np.sum(ce - cs - 1))
            own = {("u", lvl, m) for m in members}
            self._add("ec_relax_res", lvl, b, lambda p=part: real.ec_relax_res(lvl, p), reads,
                      own | {("res", lvl, m) for m in members} if lvl == 0 else own, 3 * len(part) + n_f * (2 if lvl > 0 else 1))

    # the general whole-level passes (backend_hip.gen_down / gen_up): the way down leaves the uncorrected value of every chunk's last
    # C-point in a side slab of the level ("genC"); the way up of a block reads it for the C-point its first chunk starts from (an
    # interval of the block before)
    def gen_down(self, lvl, intervals, parts=3):
        real = self.real
        for b, part in self._by_block(lvl, intervals, 1):
            cs, ce, jcs, jce = _interval_cells(self, lvl, part)
            reads, writes = set(), set()
            if parts & 1:
                reads |= {("u", lvl, b)} | self._cells("u", lvl, cs) | self._cells("u", lvl, np.maximum(cs - 1, 0))
                if lvl > 0:
                    reads |= {("g", lvl, b)} | self._cells("g", lvl, cs)
                writes |= {("u", lvl, b), ("genC", lvl, b)} | self._cells("u", lvl + 1, jce) | \
                    self._cells("v", lvl + 1, jce) | self._cells("g", lvl + 1, jce)
            if parts & 2:
                reads |= self._cells("v", lvl + 1, np.maximum(jce - 1, 0)) | self._cells("g", lvl + 1, jce)
                writes |= self._cells("g", lvl + 1, jce)
            n_f = int(np.sum(ce - cs - 1))
            self._add("gen_down" if parts & 1 else "gen_coarse", lvl, b, lambda p=part: real.gen_down(lvl, p, parts), reads, writes,
                      (6 if parts & 1 else 0) * len(part) + (3 if parts & 2 else 0) * len(part) + (n_f * (1 if lvl > 0 else 0) if parts & 1 else 0))

    def gen_up(self, lvl, intervals, residual=False):
        real = self.real
        for b, part in self._by_block(lvl, intervals, 1):
            cs, ce, jcs, jce = _interval_cells(self, lvl, part)
            reads = self._cells("u", lvl + 1, jce) | self._cells("v", lvl + 1, jce) | self._cells("u", lvl + 1, jcs[jcs >= 0]) | \
                self._cells("v", lvl + 1, jcs[jcs >= 0]) | self._cells("u", lvl, cs[jcs < 0]) | \
                {("genC", lvl, int(x)) for x in np.unique(self.block_of[lvl][cs])} | {("u", lvl, b)}
            if lvl > 0:
                reads |= {("g", lvl, b)}
            n_f = int(np.sum(ce - cs - 1))
            writes = {("u", lvl, b)} | ({("res", lvl, b)} if residual else set())
            self._add("gen_up", lvl, b, lambda p=part: real.gen_up(lvl, p, residual=residual), reads, writes,
                      6 * len(part) + n_f * (2 if lvl > 0 else 1))

    def write_generation(self):
        return getattr(self.real, "write_generation", lambda: None)()

    def residual_ready(self, points):
        self.host_after.append(lambda: self.real.residual_ready(points))

    def fas_fused(self, lvl, triples, **opts):
        real = self.real
        for b, part in self._by_block(lvl, triples, 0):
            fi, pr, co = [np.asarray([p[k] for p in part], dtype=np.int64) for k in (0, 1, 2)]
            reads = {("u", lvl, b)} | self._cells("u", lvl, fi - 1) | self._cells("u", lvl, pr)
            if lvl > 0:
                reads |= {("g", lvl, b)}
            writes = self._cells("u", lvl + 1, co) | self._cells("v", lvl + 1, co) | self._cells("g", lvl + 1, co)
            self._add("fas_fused", lvl, b, lambda p=part: real.fas_fused(lvl, p, **opts), reads, writes,
                      ((10 if opts.get("with_f_relax") else 7) if lvl > 0 else 6) * len(part))


def _interval_cells(rec, lvl, part):
    cs, ce, jcs, jce = [np.asarray([iv[k] for iv in part], dtype=np.int64) for k in range(4)]
    return cs, ce, jcs, jce


class Plan:
    """the scheduled cycle: `order` = nodes in issue order (a topological order of the graph); per node the stream it runs on
    and the nodes of the OTHER stream it has to wait for"""

    def __init__(self, nodes, n_blocks, host_after=(), sends=None, recvs=None):
        self.nodes, self.n_blocks, self.host_after = nodes, n_blocks, list(host_after)
        self.sends, self.recvs = dict(sends or {}), dict(recvs or {})     # messages per (peer rank, channel) of one cycle
        self.order = schedule(nodes)
        self.makespan = max((n.finish for n in nodes), default=0.0)
        self.has_chain = any(n.stream == "chain" for n in nodes)

    def run(self, backend):
        runner = getattr(backend, "plan_run", None)
        if runner is not None:
            runner(self)
        else:
            for node in self.order:      # host backends: the scheduled order, serially
                node.fn()
        for fn in self.host_after:
            fn()

    def describe(self):
        return [f"{n.start * 1e3:8.3f} ms  {n.stream:5s}  {n.name:18s} L{n.lvl} block {n.chunk}" for n in self.order]


def schedule(nodes):
    """List scheduling on two in-order streams ('sweep', 'chain'). Estimated costs only steer the ORDER (any order that
    respects the edges is correct). Among the ready nodes of a stream the one that can start first goes next; ties go to the
    node with the longest path behind it (the chain parts and whatever feeds them)."""
    n = len(nodes)
    bottom = [0.0] * n
    for node in reversed(nodes):        # recorded order is a topological order
        bottom[node.idx] = node.cost + max((bottom[s] for s in node.succs), default=0.0)
    remaining = [len(node.preds) for node in nodes]
    ready = {"sweep": [], "chain": []}
    for node in nodes:
        if remaining[node.idx] == 0:
            ready[node.stream].append(node.idx)
    free_at = {"sweep": 0.0, "chain": 0.0}
    done, order = 0, []
    while done < n:
        best = None
        for stream in ("sweep", "chain"):
            for idx in ready[stream]:
                node = nodes[idx]
                est = max([free_at[stream]] + [nodes[p].finish for p in node.preds])
                key = (est, -bottom[idx], idx)
                if best is None or key < best[0]:
                    best = (key, idx, stream, est)
        if best is None:
            raise RuntimeError("cycle plan: dependency graph is not acyclic")
        _, idx, stream, est = best
        node = nodes[idx]
        node.start, node.finish = est, est + node.cost
        free_at[stream] = node.finish
        ready[stream].remove(idx)
        order.append(node)
        done += 1
        for s in node.succs:
            remaining[s] -= 1
            if remaining[s] == 0:
                ready[nodes[s].stream].append(s)
    for node in order:
        node.cross_preds = tuple(nodes[p] for p in sorted(node.preds) if nodes[p].stream != node.stream)
        for p in node.cross_preds:
            p.needs_event = True
    return order


def record_cycle(mg, backend, n_blocks, walk):
    """walk() = the cycle in program order (it must call the sweeps through mg.backend); returns the Plan or raises
    PlanUnsupported"""
    rec = Recorder(mg, backend, n_blocks)
    mg.backend = rec
    try:
        walk()
    finally:
        mg.backend = backend
    return Plan(rec.nodes, rec.K, rec.host_after, rec.sends, rec.recvs)
