"""ctypes binding of the CPU oracle (oracle/liblbm_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liblbm_oracle.so")
CLI = os.path.join(ORACLE_DIR, "lbm_oracle_cli")


class OParams(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int), ("ny", ctypes.c_int), ("max_iters", ctypes.c_int),
                ("reynolds_dim", ctypes.c_int), ("density", ctypes.c_float),
                ("accel", ctypes.c_float), ("omega", ctypes.c_float)]


def build():
    out = subprocess.run(["make", "-C", ORACLE_DIR, "liblbm_oracle.so", "lbm_oracle_cli"],
                         capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + out.stdout + out.stderr)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        P, I, F, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long
        PP = ctypes.POINTER(OParams)
        lib.lbm_oracle_init_cells.argtypes = [PP, P]
        lib.lbm_oracle_timestep.argtypes = [PP, P, P, P]
        lib.lbm_oracle_accelerate_flow.argtypes = [PP, P, P]
        lib.lbm_oracle_propagate.argtypes = [PP, P, P]
        lib.lbm_oracle_rebound.argtypes = [PP, P, P, P]
        lib.lbm_oracle_collision.argtypes = [PP, P, P, P]
        lib.lbm_oracle_av_velocity.argtypes = [PP, P, P]; lib.lbm_oracle_av_velocity.restype = F
        lib.lbm_oracle_sum_velocity.argtypes = [PP, P, P, ctypes.POINTER(I)]
        lib.lbm_oracle_sum_velocity.restype = F
        lib.lbm_oracle_run.argtypes = [PP, P, P, P, P, I]
        lib.lbm_oracle_calc_reynolds.argtypes = [PP, P, P]; lib.lbm_oracle_calc_reynolds.restype = F
        lib.lbm_oracle_total_density.argtypes = [PP, P]; lib.lbm_oracle_total_density.restype = F
        lib.lbm_oracle_final_state.argtypes = [PP, P, P, P, P, P, P]
        lib.lbm_oracle_fused_rows.argtypes = [I, I, F, F, F, P, P, L, P, I, I, I]
        lib.lbm_oracle_fused_rows.restype = F
        lib.lbm_oracle_accelerate_row_soa.argtypes = [I, F, F, P, L, P, I]
        lib.lbm_oracle_fused_step_periodic.argtypes = [PP, P, P, P]
        lib.lbm_oracle_fused_step_periodic.restype = F
        lib.lbm_oracle_aos_to_soa.argtypes = [I, P, P, L]
        lib.lbm_oracle_soa_to_aos.argtypes = [I, P, L, P]

    @staticmethod
    def cparams(p):
        return OParams(p.nx, p.ny, p.max_iters, p.reynolds_dim, p.density, p.accel, p.omega)

    # ---- AoS four-sweep form ----
    def init_cells(self, p):
        cells = np.empty((p.ny, p.nx, 9), dtype=np.float32)
        self.lib.lbm_oracle_init_cells(ctypes.byref(self.cparams(p)), cells.ctypes.data)
        return cells

    def run(self, p, cells, obstacles, n_steps):
        """Advance `cells` (ny,nx,9 float32, in place) n_steps; returns av_vels (float32[n])."""
        assert cells.dtype == np.float32 and cells.flags.c_contiguous
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        tmp = np.empty_like(cells)
        av = np.empty(max(n_steps, 1), dtype=np.float32)
        self.lib.lbm_oracle_run(ctypes.byref(self.cparams(p)), cells.ctypes.data, tmp.ctypes.data,
                                ob.ctypes.data, av.ctypes.data, n_steps)
        return av[:n_steps]

    def av_velocity(self, p, cells, obstacles):
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        return float(self.lib.lbm_oracle_av_velocity(ctypes.byref(self.cparams(p)), cells.ctypes.data,
                                                     ob.ctypes.data))

    def calc_reynolds(self, p, cells, obstacles):
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        return float(self.lib.lbm_oracle_calc_reynolds(ctypes.byref(self.cparams(p)),
                                                       cells.ctypes.data, ob.ctypes.data))

    def total_density(self, p, cells):
        return float(self.lib.lbm_oracle_total_density(ctypes.byref(self.cparams(p)), cells.ctypes.data))

    def final_state(self, p, cells, obstacles):
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        f = {k: np.empty((p.ny, p.nx), dtype=np.float32) for k in ("u_x", "u_y", "u", "pressure")}
        self.lib.lbm_oracle_final_state(ctypes.byref(self.cparams(p)), cells.ctypes.data, ob.ctypes.data,
                                        f["u_x"].ctypes.data, f["u_y"].ctypes.data, f["u"].ctypes.data,
                                        f["pressure"].ctypes.data)
        return f

    # ---- SoA fused forms ----
    def fused_step_periodic(self, p, src, dst, obstacles):
        """src/dst: (9, ny, nx) float32 planes; returns the fp32 sum of |u|."""
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        return float(self.lib.lbm_oracle_fused_step_periodic(ctypes.byref(self.cparams(p)),
                                                             src.ctypes.data, dst.ctypes.data,
                                                             ob.ctypes.data))

    def fused_rows(self, nx, rows, density, accel, omega, src, dst, obstacles, accel_row,
                   row_first, row_last):
        """src/dst: (9, rows+2, nx) halo-padded slabs; obstacles: (rows, nx)."""
        assert src.shape == (9, rows + 2, nx) and dst.shape == src.shape
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        return float(self.lib.lbm_oracle_fused_rows(nx, rows, density, accel, omega, src.ctypes.data,
                                                    dst.ctypes.data, (rows + 2) * nx, ob.ctypes.data,
                                                    accel_row, row_first, row_last))


_oracle = None


def load():
    global _oracle
    if _oracle is None:
        if not os.path.exists(LIB) or not os.path.exists(CLI):
            build()
        _oracle = Oracle(ctypes.CDLL(LIB))
    return _oracle
