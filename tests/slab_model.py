"""Serial replay of the engine's row-slab protocol on the CPU oracle (TEST INFRASTRUCTURE).

`run_slabs` advances a grid cut into `n_slabs` row slabs exactly as lbm_hip.hip does -- two halo
rows around every slab, whole boundary rows shipped to the ring neighbours once per pass, a pass
advancing two timesteps (the halo-adjacent rows -1 and `rows` relaxed redundantly) or one -- with
either halo treatment:

  lag = 0   synchronous: a pass reads the halo rows of its own timestep (the MPI_Waitall pattern,
            /root/reference/MPI_Waitall/d2q9-bgk.c:225-253); must equal the single-domain oracle;
  lag = 1   stale: passes of ONE timestep; from its second pass on, a pass reads the halo rows its
            neighbours produced one pass earlier (run_steps_stale; the reproducible analogue of
            /root/reference/MPI_Testall_OptimizedVersion/d2q9-bgk.c:256-301).

`calls` lists the step counts of consecutive lbm_run calls: every call starts from fresh halos.
The per-row arithmetic is the oracle's fused row-range form.
"""
import numpy as np

H = 2  # halo rows kept around a slab


class _Slab:
    def __init__(self, oracle, p, ob, full, first, rows):
        self.o, self.p, self.first, self.rows = oracle, p, first, rows
        self.buf = [np.zeros((9, rows + 2 * H, p.nx), dtype=np.float32) for _ in range(2)]
        self.buf[0][:, H:H + rows] = full[:, first:first + rows]
        self.tmp = np.zeros_like(self.buf[0])
        self.mask = np.ascontiguousarray(ob[[(first + r) % p.ny for r in range(-1, rows + 1)]])
        self.lid = None  # local index (-1 .. rows) of the lid row if this slab holds a copy of it
        for shift in (-p.ny, 0, p.ny):
            loc = p.ny - 2 + shift - first
            if -1 <= loc <= rows:
                self.lid = loc

    def accelerate(self, arr, local_row):
        ps = (self.rows + 2 * H) * self.p.nx
        self.o.lib.lbm_oracle_accelerate_row_soa(self.p.nx, self.p.density, self.p.accel, arr.ctypes.data, ps,
                                                 self.mask[local_row + 1].ctypes.data, local_row + H)

    def relax(self, src, dst, lo, hi):
        if hi < lo:
            return 0.0
        p = self.p
        return self.o.fused_rows(p.nx, self.rows + 2, p.density, p.accel, p.omega, src, dst, self.mask, 0,
                                 lo + H, hi + H)


def _deliver(slabs, src, dst, depth):
    n = len(slabs)
    for s, sl in enumerate(slabs):
        north, south = slabs[(s + 1) % n], slabs[(s - 1 + n) % n]
        north.buf[dst][:, H - depth:H] = sl.buf[src][:, H + sl.rows - depth:H + sl.rows]
        south.buf[dst][:, H + south.rows:H + south.rows + depth] = sl.buf[src][:, H:H + depth]


def run_slabs(oracle, lbm, p, ob, calls, n_slabs, lag=0, two_step=True, cells=None):
    """Returns (lattice (ny, nx, 9) float32, tot_u float64[sum(calls)]) after the given lbm_run calls."""
    full = np.ascontiguousarray((oracle.init_cells(p) if cells is None else cells).transpose(2, 0, 1))
    slabs = []
    for s in range(n_slabs):
        first, rows = lbm.partition_rows(p.ny, n_slabs, s)
        slabs.append(_Slab(oracle, p, ob, full, first, rows))
    if any(sl.rows < 4 for sl in slabs) or lag:
        two_step = False      # stale passes always advance one timestep (run_steps_stale)
    depth = 2 if two_step else 1
    tot_u = np.zeros(sum(calls), dtype=np.float64)
    cur, done = 0, 0
    for n_steps in calls:
        t, m = 0, 0
        while t < n_steps:
            two = two_step and t + 1 < n_steps
            # accelerate_flow of the pass's first step on the owned copy (the engine: epilogue of the
            # previous pass, or accelerate_row at the start of a call); halo copies arrive accelerated
            for sl in slabs:
                if sl.lid is not None and 0 <= sl.lid < sl.rows:
                    sl.accelerate(sl.buf[cur], sl.lid)
            if lag == 0:
                _deliver(slabs, cur, cur, depth)
            elif m == 0:
                _deliver(slabs, cur, cur, depth)       # fresh halos for pass 0 ...
                _deliver(slabs, cur, cur ^ 1, depth)   # ... which pass 1 sees one pass late
            else:
                _deliver(slabs, cur, cur ^ 1, depth)   # X'(m): for pass m+1; pass m reads what X'(m-1) left
            for sl in slabs:
                S, D = sl.buf[cur], sl.buf[cur ^ 1]
                if two:
                    sl.relax(S, sl.tmp, -1, -1)
                    tot_u[done + t] += sl.relax(S, sl.tmp, 0, sl.rows - 1)
                    sl.relax(S, sl.tmp, sl.rows, sl.rows)
                    if sl.lid is not None:
                        sl.accelerate(sl.tmp, sl.lid)
                    tot_u[done + t + 1] += sl.relax(sl.tmp, D, 0, sl.rows - 1)
                else:
                    tot_u[done + t] += sl.relax(S, D, 0, sl.rows - 1)
            cur ^= 1
            t += 2 if two else 1
            m += 1
        done += n_steps
    out = np.empty((9, p.ny, p.nx), dtype=np.float32)
    for sl in slabs:
        out[:, sl.first:sl.first + sl.rows] = sl.buf[cur][:, H:H + sl.rows]
    return np.ascontiguousarray(out.transpose(1, 2, 0)), tot_u


def run_slabs_freshest(oracle, lbm, p, ob, calls, n_slabs, schedule=None, p_fresh=None, seed=0, cells=None):
    """LBM_HALO_FRESHEST, i.e. the reference's own rule (/root/reference/MPI_Testall_OptimizedVersion/d2q9-bgk.c:256-301:
    look once, then relax the boundary rows with whatever the halo rows hold) with the engine's two guarantees: in
    every pass each side of each slab finds either its neighbour's row of THIS pass (bit set in schedule[step, slab]:
    bit 0 south, bit 1 north -- the engine's lbm_read_halo_log) or the row of the pass before, which the one-pass-late
    exchange of the stale mode has put there; whole rows.  Without a schedule the bits are drawn with probability
    p_fresh (1 = the synchronous run, 0 = the stale mode).  The first pass of every call starts from fresh halos.
    Returns (lattice (ny, nx, 9), tot_u float64[steps], total density of the fluid cells after every step, the schedule)."""
    rng = np.random.default_rng(seed)
    total = sum(calls)
    if schedule is None:
        schedule = (rng.random((total, n_slabs)) < p_fresh).astype(np.uint8) | ((rng.random((total, n_slabs)) < p_fresh).astype(np.uint8) << 1)
    full = np.ascontiguousarray((oracle.init_cells(p) if cells is None else cells).transpose(2, 0, 1))
    slabs = []
    for s in range(n_slabs):
        first, rows = lbm.partition_rows(p.ny, n_slabs, s)
        slabs.append(_Slab(oracle, p, ob, full, first, rows))
    fluid = [(ob[sl.first:sl.first + sl.rows] == 0) for sl in slabs]
    tot_u = np.zeros(total, dtype=np.float64)
    mass = np.zeros(total, dtype=np.float64)
    cur, done = 0, 0
    for n_steps in calls:
        for m in range(n_steps):
            for sl in slabs:
                if sl.lid is not None and 0 <= sl.lid < sl.rows:
                    sl.accelerate(sl.buf[cur], sl.lid)
            if m == 0:
                _deliver(slabs, cur, cur, 1)
            else:
                for s, sl in enumerate(slabs):      # the rows of this pass, where they made it in time
                    north, south = slabs[(s + 1) % n_slabs], slabs[(s - 1 + n_slabs) % n_slabs]
                    if schedule[done + m, s] & 1:
                        sl.buf[cur][:, H - 1:H] = south.buf[cur][:, H + south.rows - 1:H + south.rows]
                    if schedule[done + m, s] & 2:
                        sl.buf[cur][:, H + sl.rows:H + sl.rows + 1] = north.buf[cur][:, H:H + 1]
            _deliver(slabs, cur, cur ^ 1, 1)        # one pass late for the next pass, guaranteed
            for sl in slabs:
                tot_u[done + m] += sl.relax(sl.buf[cur], sl.buf[cur ^ 1], 0, sl.rows - 1)
            cur ^= 1
            mass[done + m] = sum(float(sl.buf[cur][:, H:H + sl.rows].sum(axis=0, dtype=np.float64)[f].sum()) for sl, f in zip(slabs, fluid))
        done += n_steps
    out = np.empty((9, p.ny, p.nx), dtype=np.float32)
    for sl in slabs:
        out[:, sl.first:sl.first + sl.rows] = sl.buf[cur][:, H:H + sl.rows]
    return np.ascontiguousarray(out.transpose(1, 2, 0)), tot_u, mass, schedule
