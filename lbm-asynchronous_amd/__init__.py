"""Python host-side binding of the MI355X D2Q9-BGK engine (ctypes over the C ABI).

The directory name carries a hyphen (it mirrors the reference repository's name), so load it
with :func:`importlib` -- ``tests/conftest.py`` and ``bench.py`` do::

    spec = importlib.util.spec_from_file_location("lbm_asynchronous_amd", ".../__init__.py")

Everything here is plumbing around ``liblbm_hip.so`` (``include/lbm_hip.h``): numpy arrays in the
reference's host layouts go in, numpy arrays come out.  There is no Python or CPU compute path;
if the shared library is missing or no HIP device is usable, calls raise :class:`LbmError`.

Reference interfaces mirrored (``/root/reference/SerialCode/d2q9-bgk.c``):
``t_param`` (:66-75) -> :class:`Params`; ``initialise`` file formats (:460-613) ->
:func:`read_params`, :func:`read_obstacles`; the ``timestep``/``av_velocity`` loop (:166-170) ->
:meth:`Engine.run`; ``write_values`` (:662-743) -> :func:`write_final_state`, :func:`write_av_vels`.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LBM_LIB") or os.path.join(_HERE, "liblbm_hip.so")   # LBM_LIB: experiments only
CLI_PATH = os.path.join(_HERE, "d2q9-bgk")

MATH_EXACT = 0
MATH_FAST = 1
_MATH = {"exact": MATH_EXACT, "fast": MATH_FAST, MATH_EXACT: MATH_EXACT, MATH_FAST: MATH_FAST}
RCCL_ID_BYTES = 128
MASK_HALO_ROWS = 3        # LBM_MASK_HALO_ROWS of include/lbm_hip.h
HALO_SYNC = 0
HALO_STALE = 1
HALO_FRESHEST = 2
_HALO = {"sync": HALO_SYNC, "stale": HALO_STALE, "freshest": HALO_FRESHEST,
         HALO_SYNC: HALO_SYNC, HALO_STALE: HALO_STALE, HALO_FRESHEST: HALO_FRESHEST}

# every symbol include/lbm_hip.h declares (tests check the .so exports them all)
ABI_SYMBOLS = (
    "lbm_set_error_mode", "lbm_last_error", "lbm_version", "lbm_device_count",
    "lbm_partition_rows", "lbm_halo_plan", "lbm_plan_halo_depth", "lbm_create", "lbm_rccl_unique_id", "lbm_rccl_info", "lbm_create_rank", "lbm_create_rank_rows",
    "lbm_create_tiled", "lbm_create_rank_tiled", "lbm_create_rank_hosted", "lbm_create_rank_hosted_rows",
    "lbm_create_rank_hosted_tiled", "lbm_destroy",
    "lbm_get_info", "lbm_set_halo_mode", "lbm_read_halo_log", "lbm_run", "lbm_sync", "lbm_run_timed", "lbm_read_av_vels", "lbm_read_cells",
    "lbm_read_final_state", "lbm_av_velocity", "lbm_total_density", "lbm_calc_reynolds",
)


class LbmError(RuntimeError):
    pass


class _CParams(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int), ("ny", ctypes.c_int), ("max_iters", ctypes.c_int),
                ("reynolds_dim", ctypes.c_int), ("density", ctypes.c_float),
                ("accel", ctypes.c_float), ("omega", ctypes.c_float)]


class _CInfo(ctypes.Structure):
    _fields_ = [("n_slabs", ctypes.c_int), ("row_first", ctypes.c_int), ("row_count", ctypes.c_int),
                ("fluid_cells", ctypes.c_int), ("steps_done", ctypes.c_int),
                ("math_mode", ctypes.c_int), ("world_rank", ctypes.c_int),
                ("world_size", ctypes.c_int), ("steps_per_launch", ctypes.c_int),
                ("halo_mode", ctypes.c_int), ("band_rows", ctypes.c_int), ("lane_cells", ctypes.c_int),
                ("nontemporal", ctypes.c_int), ("graph_steps", ctypes.c_int),
                ("resident_steps", ctypes.c_int), ("resident_min_steps", ctypes.c_int),
                ("resident_rows", ctypes.c_int), ("resident_group", ctypes.c_int), ("resident_one_xcd", ctypes.c_int)]


class _CRcclStatus(ctypes.Structure):
    _fields_ = [("loaded", ctypes.c_int), ("version", ctypes.c_int), ("n_comms", ctypes.c_int),
                ("nranks", ctypes.c_int), ("rank", ctypes.c_int), ("library", ctypes.c_char * 512)]


class _CHaloOp(ctypes.Structure):
    _fields_ = [("is_send", ctypes.c_int), ("peer", ctypes.c_int), ("row_first", ctypes.c_int),
                ("row_count", ctypes.c_int)]


_EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_CHaloOp),
                                ctypes.POINTER(ctypes.POINTER(ctypes.c_float)), ctypes.c_size_t)
_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int)


class _CHostComm(ctypes.Structure):
    _fields_ = [("exchange", _EXCHANGE_FN), ("allreduce_sum", _ALLREDUCE_FN), ("user", ctypes.c_void_p)]


@dataclass
class Params:
    """The reference's t_param (SerialCode/d2q9-bgk.c:66-75)."""
    nx: int
    ny: int
    max_iters: int
    reynolds_dim: int
    density: float
    accel: float
    omega: float

    def _c(self) -> _CParams:
        return _CParams(self.nx, self.ny, self.max_iters, self.reynolds_dim,
                        self.density, self.accel, self.omega)


# ------------------------------------------------------------------------------------------------
# building and loading the shared library
# ------------------------------------------------------------------------------------------------
def build(force: bool = False) -> None:
    """Compile liblbm_hip.so (gfx950) and the d2q9-bgk host program in-tree with make."""
    cmd = ["make", "-C", _HERE] + (["-B"] if force else [])
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        raise LbmError("building liblbm_hip.so failed:\n" + out.stdout + out.stderr)


_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen liblbm_hip.so and declare the C-ABI prototypes.  Fails loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LbmError(f"{LIB_PATH} not built: run `make -C {_HERE}` (no fallback path exists)")
    lib = ctypes.CDLL(LIB_PATH)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    PF, PI = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    lib.lbm_set_error_mode.argtypes = [I]; lib.lbm_set_error_mode.restype = None
    lib.lbm_last_error.argtypes = []; lib.lbm_last_error.restype = ctypes.c_char_p
    lib.lbm_version.argtypes = []; lib.lbm_version.restype = ctypes.c_char_p
    lib.lbm_device_count.argtypes = []; lib.lbm_device_count.restype = I
    lib.lbm_partition_rows.argtypes = [I, I, I, PI, PI]; lib.lbm_partition_rows.restype = I
    lib.lbm_halo_plan.argtypes = [I, I, I, I, ctypes.POINTER(_CHaloOp)]; lib.lbm_halo_plan.restype = I
    lib.lbm_plan_halo_depth.argtypes = [ctypes.POINTER(_CParams), I, I]; lib.lbm_plan_halo_depth.restype = I
    lib.lbm_create.argtypes = [ctypes.POINTER(_CParams), P, P, I, I]; lib.lbm_create.restype = P
    lib.lbm_rccl_unique_id.argtypes = [P]; lib.lbm_rccl_unique_id.restype = I
    lib.lbm_rccl_info.argtypes = [P, ctypes.POINTER(_CRcclStatus)]; lib.lbm_rccl_info.restype = I
    lib.lbm_create_rank.argtypes = [ctypes.POINTER(_CParams), P, P, I, I, P, I, I]
    lib.lbm_create_rank.restype = P
    lib.lbm_create_rank_rows.argtypes = [ctypes.POINTER(_CParams), P, P, I, I, P, I, I]
    lib.lbm_create_rank_rows.restype = P
    lib.lbm_create_tiled.argtypes = [ctypes.POINTER(_CParams), P, I, I, P, I, I]; lib.lbm_create_tiled.restype = P
    lib.lbm_create_rank_tiled.argtypes = [ctypes.POINTER(_CParams), P, I, I, I, I, P, I, I]
    lib.lbm_create_rank_tiled.restype = P
    lib.lbm_create_rank_hosted.argtypes = [ctypes.POINTER(_CParams), P, P, I, I, ctypes.POINTER(_CHostComm), I, I]
    lib.lbm_create_rank_hosted.restype = P
    lib.lbm_create_rank_hosted_rows.argtypes = [ctypes.POINTER(_CParams), P, P, I, I, ctypes.POINTER(_CHostComm), I, I]
    lib.lbm_create_rank_hosted_rows.restype = P
    lib.lbm_create_rank_hosted_tiled.argtypes = [ctypes.POINTER(_CParams), P, I, I, I, I, ctypes.POINTER(_CHostComm), I, I]
    lib.lbm_create_rank_hosted_tiled.restype = P
    lib.lbm_destroy.argtypes = [P]; lib.lbm_destroy.restype = None
    lib.lbm_get_info.argtypes = [P, ctypes.POINTER(_CInfo)]; lib.lbm_get_info.restype = I
    lib.lbm_set_halo_mode.argtypes = [P, I]; lib.lbm_set_halo_mode.restype = I
    lib.lbm_read_halo_log.argtypes = [P, ctypes.c_void_p, I]; lib.lbm_read_halo_log.restype = I
    lib.lbm_run.argtypes = [P, I]; lib.lbm_run.restype = I
    lib.lbm_sync.argtypes = [P]; lib.lbm_sync.restype = I
    lib.lbm_run_timed.argtypes = [P, I, PF]; lib.lbm_run_timed.restype = I
    lib.lbm_read_av_vels.argtypes = [P, P, I]; lib.lbm_read_av_vels.restype = I
    lib.lbm_read_cells.argtypes = [P, P]; lib.lbm_read_cells.restype = I
    lib.lbm_read_final_state.argtypes = [P, P, P, P, P]; lib.lbm_read_final_state.restype = I
    lib.lbm_av_velocity.argtypes = [P, PF]; lib.lbm_av_velocity.restype = I
    lib.lbm_total_density.argtypes = [P, ctypes.POINTER(ctypes.c_double)]
    lib.lbm_total_density.restype = I
    lib.lbm_calc_reynolds.argtypes = [P, PF]; lib.lbm_calc_reynolds.restype = I
    # a Python host wants exceptions, not exit(): switch from the reference's die() behaviour
    lib.lbm_set_error_mode(1)
    _lib = lib
    return lib


def _check(lib, rc) -> None:
    if rc != 0:
        raise LbmError(lib.lbm_last_error().decode())


def device_count() -> int:
    return int(load_library().lbm_device_count())


def partition_rows(ny: int, parts: int, index: int) -> tuple[int, int]:
    """Rows [first, first+count) owned by part `index` of `parts` (lbm_partition_rows)."""
    lib = load_library()
    first, count = ctypes.c_int(), ctypes.c_int()
    _check(lib, lib.lbm_partition_rows(ny, parts, index, ctypes.byref(first), ctypes.byref(count)))
    return first.value, count.value


def halo_plan(rows: int, parts: int, index: int, depth: int) -> list[dict]:
    """The four halo messages of one pass in the engine's posting order (lbm_halo_plan)."""
    lib = load_library()
    ops = (_CHaloOp * 4)()
    _check(lib, lib.lbm_halo_plan(rows, parts, index, depth, ops))
    return [{"is_send": bool(o.is_send), "peer": o.peer, "row_first": o.row_first, "row_count": o.row_count}
            for o in ops]


def plan_halo_depth(params: "Params", parts: int, math: str | int = "exact") -> int:
    """Halo depth = timesteps per pass the engine uses for `parts` row slabs (lbm_plan_halo_depth)."""
    lib = load_library()
    cp = params._c()
    d = lib.lbm_plan_halo_depth(ctypes.byref(cp), parts, _MATH[math])
    if d <= 0:
        raise LbmError(lib.lbm_last_error().decode())
    return int(d)


def _rccl_status(lib, handle) -> dict:
    st = _CRcclStatus()
    _check(lib, lib.lbm_rccl_info(handle, ctypes.byref(st)))
    v = st.version
    return {"loaded": bool(st.loaded), "version": v,
            "version_string": f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v else None,
            "n_comms": st.n_comms, "nranks": st.nranks, "rank": st.rank,
            "library": st.library.decode() or None}


def rccl_info() -> dict:
    """Bind RCCL as the first multi-GPU create would and report which library and version (lbm_rccl_info(NULL))."""
    return _rccl_status(load_library(), None)


def rccl_unique_id() -> bytes:
    lib = load_library()
    buf = ctypes.create_string_buffer(RCCL_ID_BYTES)
    _check(lib, lib.lbm_rccl_unique_id(buf))
    return buf.raw


# ------------------------------------------------------------------------------------------------
# the engine handle
# ------------------------------------------------------------------------------------------------
class Engine:
    """One lattice on one or more GPUs.  Mirrors the reference's main()-level use of the hot path:
    create (initialise), run (the timestep/av_velocity loop), read results, close (finalise)."""

    def __init__(self, params: Params, obstacles: np.ndarray, cells: np.ndarray | None = None,
                 n_gpus: int = 1, math: str | int = "exact", *, rank: int | None = None,
                 world_size: int | None = None, unique_id: bytes | None = None, device: int = 0,
                 tiled: bool = False, local_rows: bool = False, host_comm=None):
        """obstacles: the global (ny, nx) map; with tiled=True a small (tile_ny, tile_nx) map repeated
        periodically over the grid (lbm_create_tiled / lbm_create_rank_tiled: the mask is expanded on the
        device); with local_rows=True (rank form only) this rank's rows preceded and followed by
        MASK_HALO_ROWS periodic neighbour rows, and `cells` this rank's rows only (lbm_create_rank_rows)."""
        self.lib = load_library()
        self.params = params
        if params.nx < 1 or params.ny < 2 or params.max_iters < 0:
            raise LbmError("lbm_create: invalid parameters")
        obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
        cell_rows = params.ny
        if tiled:
            if obstacles.ndim != 2:
                raise LbmError("lbm_create_tiled: the tile must be a 2-D map")
        elif local_rows:
            if rank is None:
                raise LbmError("local_rows needs the rank form")
            _, cell_rows = partition_rows(params.ny, world_size, rank)
            if obstacles.size != params.nx * (cell_rows + 2 * MASK_HALO_ROWS):
                raise LbmError("lbm_create_rank_rows: obstacle rows do not match (row_count + 2*MASK_HALO_ROWS)*nx")
        else:
            if obstacles.size != params.nx * params.ny:
                raise LbmError("lbm_create: obstacle map does not match nx*ny")
            obstacles = obstacles.reshape(params.ny, params.nx)
        self._obstacles = obstacles
        cptr = None
        if cells is not None:
            cells = np.ascontiguousarray(cells, dtype=np.float32)
            if cells.size != params.nx * cell_rows * 9:
                raise LbmError("lbm_create: cells do not match nx*ny*9")
            cells = cells.reshape(cell_rows, params.nx, 9)
            cptr = cells.ctypes.data
        cp = params._c()
        idbuf = ctypes.create_string_buffer(unique_id, RCCL_ID_BYTES) if unique_id else None
        if host_comm is not None:
            if rank is None:
                raise LbmError("host_comm needs the rank form")
            self._host_comm = self._wrap_host_comm(*host_comm)
            if tiled:
                if cells is not None:
                    raise LbmError("lbm_create_rank_hosted_tiled starts from the uniform equilibrium")
                h = self.lib.lbm_create_rank_hosted_tiled(ctypes.byref(cp), obstacles.ctypes.data, obstacles.shape[1],
                                                          obstacles.shape[0], rank, world_size,
                                                          ctypes.byref(self._host_comm), device, _MATH[math])
            elif local_rows:
                h = self.lib.lbm_create_rank_hosted_rows(ctypes.byref(cp), obstacles.ctypes.data, cptr, rank, world_size,
                                                         ctypes.byref(self._host_comm), device, _MATH[math])
            else:
                h = self.lib.lbm_create_rank_hosted(ctypes.byref(cp), obstacles.ctypes.data, cptr, rank, world_size,
                                                    ctypes.byref(self._host_comm), device, _MATH[math])
        elif rank is None:
            if tiled:
                h = self.lib.lbm_create_tiled(ctypes.byref(cp), obstacles.ctypes.data, obstacles.shape[1],
                                              obstacles.shape[0], cptr, n_gpus, _MATH[math])
            else:
                h = self.lib.lbm_create(ctypes.byref(cp), obstacles.ctypes.data, cptr, n_gpus, _MATH[math])
        elif tiled:
            if cells is not None:
                raise LbmError("lbm_create_rank_tiled starts from the uniform equilibrium")
            h = self.lib.lbm_create_rank_tiled(ctypes.byref(cp), obstacles.ctypes.data, obstacles.shape[1],
                                               obstacles.shape[0], rank, world_size, idbuf, device, _MATH[math])
        elif local_rows:
            h = self.lib.lbm_create_rank_rows(ctypes.byref(cp), obstacles.ctypes.data, cptr, rank,
                                              world_size, idbuf, device, _MATH[math])
        else:
            h = self.lib.lbm_create_rank(ctypes.byref(cp), obstacles.ctypes.data, cptr, rank,
                                         world_size, idbuf, device, _MATH[math])
        if not h:
            raise LbmError(self.lib.lbm_last_error().decode())
        self.handle = ctypes.c_void_p(h)

    @staticmethod
    def _wrap_host_comm(exchange, allreduce_sum):
        def c_exchange(_user, n_ops, ops, buffers, floats):
            try:
                plan = [{"is_send": bool(ops[i].is_send), "peer": ops[i].peer, "row_first": ops[i].row_first,
                         "row_count": ops[i].row_count} for i in range(n_ops)]
                views = [np.ctypeslib.as_array(buffers[i], shape=(floats,)) for i in range(n_ops)]
                exchange(plan, views)
                return 0
            except Exception:                   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        def c_allreduce(_user, values, n):
            try:
                allreduce_sum(np.ctypeslib.as_array(values, shape=(n,)))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        comm = _CHostComm(_EXCHANGE_FN(c_exchange), _ALLREDUCE_FN(c_allreduce), None)
        comm._keep = (c_exchange, c_allreduce)   # the callbacks must outlive the context
        return comm

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.lbm_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        ci = _CInfo()
        _check(self.lib, self.lib.lbm_get_info(self.handle, ctypes.byref(ci)))
        return {name: getattr(ci, name) for name, _ in _CInfo._fields_}

    def rccl_info(self) -> dict:
        """Which RCCL this context's communicators come from and what they say about the ring (lbm_rccl_info)."""
        return _rccl_status(self.lib, self.handle)

    def set_halo_mode(self, mode) -> None:
        """'sync' (halo rows of the same timestep, the MPI_Waitall pattern), 'stale' (one pass
        old: reproducible analogue of the reference's MPI_Testall variant) or 'freshest' (the rows of this
        step where they have arrived by the time the interior rows are done, else the rows of the step before)."""
        if mode not in _HALO:
            raise LbmError(f"unknown halo mode {mode!r}")
        _check(self.lib, self.lib.lbm_set_halo_mode(self.handle, _HALO[mode]))

    def halo_log(self, n_steps: int) -> np.ndarray:
        """uint8 [n_steps, n_slabs]: what each look of the 'freshest' mode found (bit 0 south, bit 1 north halo row
        fresh); lbm_read_halo_log."""
        out = np.zeros((int(n_steps), self.info()["n_slabs"]), dtype=np.uint8)
        _check(self.lib, self.lib.lbm_read_halo_log(self.handle, out.ctypes.data, int(n_steps)))
        return out

    # -- hot path ----------------------------------------------------------------------------
    def run(self, n_steps: int) -> None:
        _check(self.lib, self.lib.lbm_run(self.handle, int(n_steps)))

    def run_timed(self, n_steps: int) -> float:
        """Advance n_steps; returns the average device milliseconds per step (HIP events on the
        compute stream)."""
        ms = ctypes.c_float()
        _check(self.lib, self.lib.lbm_run_timed(self.handle, int(n_steps), ctypes.byref(ms)))
        return float(ms.value)

    def sync(self) -> None:
        _check(self.lib, self.lib.lbm_sync(self.handle))

    # -- results -----------------------------------------------------------------------------
    def av_vels(self, n: int | None = None) -> np.ndarray:
        n = self.info()["steps_done"] if n is None else n
        out = np.empty(n, dtype=np.float32)
        _check(self.lib, self.lib.lbm_read_av_vels(self.handle, out.ctypes.data, n))
        return out

    def cells(self) -> np.ndarray:
        """Owned rows of the lattice in the reference's AoS layout: (rows, nx, 9) float32."""
        rows = self.info()["row_count"]
        out = np.empty((rows, self.params.nx, 9), dtype=np.float32)
        _check(self.lib, self.lib.lbm_read_cells(self.handle, out.ctypes.data))
        return out

    def final_state(self) -> dict:
        rows = self.info()["row_count"]
        shape = (rows, self.params.nx)
        f = {k: np.empty(shape, dtype=np.float32) for k in ("u_x", "u_y", "u", "pressure")}
        _check(self.lib, self.lib.lbm_read_final_state(
            self.handle, f["u_x"].ctypes.data, f["u_y"].ctypes.data, f["u"].ctypes.data,
            f["pressure"].ctypes.data))
        return f

    def av_velocity(self) -> float:
        v = ctypes.c_float()
        _check(self.lib, self.lib.lbm_av_velocity(self.handle, ctypes.byref(v)))
        return float(v.value)

    def total_density(self) -> float:
        v = ctypes.c_double()
        _check(self.lib, self.lib.lbm_total_density(self.handle, ctypes.byref(v)))
        return float(v.value)

    def reynolds(self) -> float:
        v = ctypes.c_float()
        _check(self.lib, self.lib.lbm_calc_reynolds(self.handle, ctypes.byref(v)))
        return float(v.value)


# ------------------------------------------------------------------------------------------------
# file formats of the command line (Python twins of host/lbm_io.c, for tests and bench.py)
# ------------------------------------------------------------------------------------------------
def read_params(path: str) -> Params:
    """7 whitespace-separated values, fixed order (SerialCode/d2q9-bgk.c:480-506)."""
    with open(path) as fh:
        tok = fh.read().split()
    names = ("nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega")
    if len(tok) < 7:
        raise LbmError(f"could not read param file: {names[len(tok)]}")
    try:
        return Params(int(tok[0]), int(tok[1]), int(tok[2]), int(tok[3]),
                      float(tok[4]), float(tok[5]), float(tok[6]))
    except ValueError as exc:
        raise LbmError(f"could not read param file: {exc}") from None


def read_obstacles(path: str, nx: int, ny: int) -> np.ndarray:
    """'x y 1' lines -> int32 (ny, nx) map; the reference's range checks (:590-597)."""
    grid = np.zeros((ny, nx), dtype=np.int32)
    with open(path) as fh:
        tok = fh.read().split()
    if len(tok) % 3:
        raise LbmError("expected 3 values per line in obstacle file")
    try:
        arr = np.array(tok, dtype=np.int64).reshape(-1, 3)
    except ValueError:
        raise LbmError("expected 3 values per line in obstacle file") from None
    if arr.size:
        if (arr[:, 0] < 0).any() or (arr[:, 0] > nx - 1).any():
            raise LbmError("obstacle x-coord out of range")
        if (arr[:, 1] < 0).any() or (arr[:, 1] > ny - 1).any():
            raise LbmError("obstacle y-coord out of range")
        if (arr[:, 2] != 1).any():
            raise LbmError("obstacle blocked value should be 1")
        grid[arr[:, 1], arr[:, 0]] = 1
    return grid


def tile_obstacles(tile: np.ndarray, nx: int, ny: int) -> np.ndarray:
    """Synthetic large grids: repeat a small map periodically."""
    ty, tx = tile.shape
    reps = (-(-ny // ty), -(-nx // tx))
    return np.ascontiguousarray(np.tile(tile, reps)[:ny, :nx], dtype=np.int32)


def write_av_vels(path: str, av_vels: np.ndarray) -> None:
    """'%d:\\t%.12E\\n' (SerialCode/d2q9-bgk.c:735-738)."""
    with open(path, "w") as fh:
        for i, v in enumerate(np.asarray(av_vels, dtype=np.float32)):
            fh.write("%d:\t%.12E\n" % (i, float(v)))


def write_final_state(path: str, fields: dict, obstacles: np.ndarray) -> None:
    """'%d %d %.12E %.12E %.12E %.12E %d\\n', jj outer / ii inner (:679-723)."""
    ny, nx = obstacles.shape
    with open(path, "w") as fh:
        for jj in range(ny):
            ux, uy, u, pr, ob = (fields["u_x"][jj], fields["u_y"][jj], fields["u"][jj],
                                 fields["pressure"][jj], obstacles[jj])
            fh.write("".join("%d %d %.12E %.12E %.12E %.12E %d\n" %
                             (ii, jj, ux[ii], uy[ii], u[ii], pr[ii], ob[ii]) for ii in range(nx)))


# ------------------------------------------------------------------------------------------------
# the acceptance rule of the reference's check/check.py (:83-99, :136-148)
# ------------------------------------------------------------------------------------------------
def check_rule(ref: np.ndarray, sim: np.ndarray) -> dict:
    """max over entries of 100*(ref-sim)/sim, as check.py computes it (diff/(ref-diff))."""
    ref = np.asarray(ref, dtype=np.float64).ravel()
    sim = np.asarray(sim, dtype=np.float64).ravel()
    if ref.size != sim.size:
        raise LbmError("Different number of steps in av_vels files")
    diff = ref - sim
    with np.errstate(divide="ignore", invalid="ignore"):
        pct = 100.0 * (diff / (ref - diff))
    k = int(np.argmax(np.abs(pct)))
    return {"index": k, "max_diff": float(diff[k]), "max_diff_pcnt": float(pct[k]),
            "sim_val": float(sim[k]), "ref_val": float(ref[k]), "total": float(np.abs(diff).sum())}


def check_passes(ref: np.ndarray, sim: np.ndarray, tolerance_pct: float = 1.0) -> bool:
    d = check_rule(ref, sim)
    return bool(np.isfinite(d["max_diff_pcnt"]) and abs(d["max_diff_pcnt"]) <= tolerance_pct)
