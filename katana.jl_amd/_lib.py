"""ctypes binding of libkatana_hip.so (the C ABI of include/katana_hip.h)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KTN_LIB") or os.path.join(HERE, "libkatana_hip.so")      # (KTN_LIB: an alternative build, for A/B timing)

KTN_OK = 0
E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5
STATUS_NONE, STATUS_OPTIMAL, STATUS_UNBOUNDED, STATUS_INFEASIBLE, STATUS_USERLIMIT, STATUS_ERROR = range(6)
MIN, MAX = 0, 1
ROW_SEP, ROW_TAPE, ROW_HOST = 0, 1, 2
ATOM_LIN, ATOM_QUAD, ATOM_EXP, ATOM_NEGLOG = 0, 1, 2, 3
(OP_CONST, OP_VAR, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_POWC, OP_EXP, OP_LOG, OP_SQRT, OP_SIN,
 OP_COS) = range(13)

c_i64, c_i32, c_f64, c_u8 = C.c_int64, C.c_int32, C.c_double, C.c_uint8
P = C.POINTER


class KtnParams(C.Structure):
    _fields_ = [("f_tol", c_f64), ("cut_coef_rng", c_f64), ("log_level", c_i32), ("iter_cap", c_i32),
                ("obj_eps", c_f64), ("vis_data", c_i32), ("device", c_i32), ("lp_max_iter", c_i32),
                ("lp_check_every", c_i32), ("lp_ruiz_iters", c_i32), ("lp_tol_scale", c_f64),
                ("lp_tol_floor", c_f64), ("lp_tol_cap", c_f64), ("lp_gap_floor", c_f64), ("lp_gap_cap", c_f64),
                ("lp_dual_inherit", c_i32), ("profile", c_i32), ("purge_age", c_i32), ("purge_margin", c_f64),
                ("purge_min_frac", c_f64), ("purge_min_rows", c_i64), ("lp_dense_after", c_i32), ("cut_cap_factor", c_f64), ("cut_cap_min", c_i64), ("lp_stag_factor", c_f64),
                ("lp_ruiz_warm", c_i32), ("lp_tiled_nnz", c_i64), ("lp_near_check", c_i32), ("dedupe_eps", c_f64), ("polish_factor", c_f64), ("polish_max_var", c_i32), ("polish_max_iter", c_i32),
                ("epi_shift", c_i32), ("obj_cert_tol", c_f64), ("lp_mid_max_var", c_i32)]


class KtnNlpDesc(C.Structure):
    _fields_ = [("num_var", c_i64), ("num_constr", c_i64), ("rowptr", P(c_i64)), ("col", P(c_i32)),
                ("row_kind", P(c_u8)), ("row_linear", P(c_u8)), ("rconst", P(c_f64)), ("atom_kind", P(c_u8)),
                ("p0", P(c_f64)), ("p1", P(c_f64)), ("tape_ptr", P(c_i64)), ("tape_op", P(c_i32)),
                ("tape_arg", P(c_f64)), ("obj_linear", c_i32), ("obj_kind", c_i32), ("obj_nnz", c_i64),
                ("obj_col", P(c_i32)), ("obj_atom_kind", P(c_u8)), ("obj_p0", P(c_f64)), ("obj_p1", P(c_f64)),
                ("obj_const", c_f64), ("obj_tape_len", c_i64), ("obj_tape_op", P(c_i32)), ("obj_tape_arg", P(c_f64)),
                ("eval_rows", C.c_void_p), ("eval_obj", C.c_void_p), ("eval_user", C.c_void_p)]


# ktn_eval_rows_cb / ktn_eval_obj_cb (include/katana_hip.h)
EVAL_ROWS_CB = C.CFUNCTYPE(c_i32, C.c_void_p, P(c_f64), P(c_f64), P(c_f64))
EVAL_OBJ_CB = C.CFUNCTYPE(c_i32, C.c_void_p, P(c_f64), P(c_f64), P(c_f64))
# ktn_exchange_cb: (user, what, first_new_row, scalars, nscalars)
EXCHANGE_CB = C.CFUNCTYPE(c_i32, C.c_void_p, c_i32, c_i64, P(c_f64), c_i32)


ALLREDUCE_CB = C.CFUNCTYPE(c_i32, C.c_void_p, P(c_f64), c_i64, c_i32)

# name -> (restype, argtypes); every function declared in include/katana_hip.h
PROTOTYPES = {
    "ktn_abi_version": (c_i32, []),
    "ktn_sizeof_params": (c_i64, []),
    "ktn_sizeof_nlp_desc": (c_i64, []),
    "ktn_default_params": (None, [P(KtnParams)]),
    "ktn_create": (c_i32, [P(KtnParams), P(C.c_void_p)]),
    "ktn_destroy": (None, [C.c_void_p]),
    "ktn_last_error": (C.c_char_p, [C.c_void_p]),
    "ktn_loadproblem": (c_i32, [C.c_void_p, c_i64, c_i64, P(c_f64), P(c_f64), P(c_f64), P(c_f64), c_i32,
                                P(KtnNlpDesc)]),
    "ktn_optimize": (c_i32, [C.c_void_p]),
    "ktn_optimize_begin": (c_i32, [C.c_void_p]),
    "ktn_ecp_step": (c_i32, [C.c_void_p, P(c_i32)]),
    "ktn_optimize_end": (c_i32, [C.c_void_p]),
    "ktn_reset": (c_i32, [C.c_void_p]),
    "ktn_get_status": (c_i32, [C.c_void_p]),
    "ktn_get_objval": (c_f64, [C.c_void_p]),
    "ktn_get_num_var": (c_i64, [C.c_void_p]),
    "ktn_get_solution": (c_i32, [C.c_void_p, P(c_f64), c_i64]),
    "ktn_get_solvetime": (c_f64, [C.c_void_p]),
    "ktn_numiters": (c_i64, [C.c_void_p]),
    "ktn_numcuts": (c_i64, [C.c_void_p]),
    "ktn_setwarmstart": (c_i32, [C.c_void_p, P(c_f64), c_i64]),
    "ktn_sep_precompute": (c_i32, [C.c_void_p, P(c_f64), c_i64]),
    "ktn_sep_num_constr": (c_i64, [C.c_void_p]),
    "ktn_sep_jac_nnz": (c_i64, [C.c_void_p]),
    "ktn_sep_get_g": (c_i32, [C.c_void_p, P(c_f64), c_i64]),
    "ktn_sep_get_jac": (c_i32, [C.c_void_p, P(c_f64), c_i64]),
    "ktn_sep_get_structure": (c_i32, [C.c_void_p, P(c_i64), P(c_i32)]),
    "ktn_sep_isconstrsat": (c_i32, [C.c_void_p, c_i64, c_f64, c_f64, c_f64]),
    "ktn_sep_gencut": (c_i32, [C.c_void_p, c_i64, P(c_i32), P(c_f64), P(c_i64), P(c_f64)]),
    "ktn_sep_sweep": (c_i32, [C.c_void_p, c_f64, P(c_i64), P(c_f64)]),
    "ktn_lp_num_rows": (c_i64, [C.c_void_p]),
    "ktn_lp_nnz": (c_i64, [C.c_void_p]),
    "ktn_lp_get_rows": (c_i32, [C.c_void_p, P(c_i64), P(c_i32), P(c_f64), P(c_f64), P(c_f64)]),
    "ktn_lp_get_objective": (c_i32, [C.c_void_p, P(c_f64), c_i64, P(c_f64)]),
    "ktn_lp_get_duals": (c_i32, [C.c_void_p, P(c_f64), c_i64]),
    "ktn_lp_solve": (c_i32, [C.c_void_p, c_f64, c_f64, P(c_i32), P(c_i64)]),
    "ktn_lp_pdhg_raw": (c_i32, [C.c_void_p, P(c_f64), P(c_f64), c_f64, c_f64, c_i64, P(c_f64), P(c_f64)]),
    "ktn_num_lp_sols": (c_i64, [C.c_void_p]),
    "ktn_get_lp_sol": (c_i32, [C.c_void_p, c_i64, P(c_f64), c_i64]),
    "ktn_get_stat": (c_f64, [C.c_void_p, C.c_char_p]),
    "ktn_sweep_lp_point": (c_i32, [C.c_void_p, c_f64, P(c_i64), P(c_f64)]),
    "ktn_objective_certificate": (c_i32, [C.c_void_p, c_i64, P(c_f64)]),
    "ktn_lp_nnz_from": (c_i64, [C.c_void_p, c_i64]),
    "ktn_lp_get_rows_from": (c_i32, [C.c_void_p, c_i64, P(c_i64), P(c_i32), P(c_f64), P(c_f64), P(c_f64)]),
    "ktn_lp_truncate": (c_i32, [C.c_void_p, c_i64]),
    "ktn_lp_pack_rows_dev": (c_i32, [C.c_void_p, c_i64, c_i64, C.c_void_p, c_i64, P(c_i64), P(c_i64)]),
    "ktn_lp_append_packed_dev": (c_i32, [C.c_void_p, c_i64, c_i64, C.c_void_p]),
    "ktn_lp_purge": (c_i32, [C.c_void_p, P(c_i64)]),
    "ktn_lp_enable_global_lists": (c_i32, [C.c_void_p, c_i64]),
    "ktn_set_cut_exchange": (c_i32, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64]),
    "ktn_dist_release_ipc": (c_i32, [C.c_void_p]),
    "ktn_last_sweep_slots": (c_i32, [C.c_void_p, P(c_i64), c_i64, P(c_i64)]),
    "ktn_lp_append_rows_nl": (c_i32, [C.c_void_p, c_i64, P(c_i64), P(c_i32), P(c_f64), P(c_f64), P(c_f64), P(c_i64)]),
    "ktn_set_blocks": (c_i32, [C.c_void_p, c_i64, P(c_i64)]),
    "ktn_optimize_blocks": (c_i32, [C.c_void_p, c_i32]),
    "ktn_dist_unique_id": (c_i32, [C.c_char_p]),
    "ktn_dist_init_rccl": (c_i32, [C.c_void_p, C.c_char_p, c_i32, c_i32]),
    "ktn_dist_init_callback": (c_i32, [C.c_void_p, c_i32, c_i32, C.c_void_p, C.c_void_p]),
    "ktn_dist_ipc_export": (c_i32, [C.c_void_p, c_i32, c_i32, c_i64, C.c_char_p]),
    "ktn_dist_init_ipc": (c_i32, [C.c_void_p, c_i32, c_i32, C.c_char_p]),
    "ktn_dist_allreduce_probe": (c_i32, [C.c_void_p, c_i64, c_i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ktn_lp_append_rows": (c_i32, [C.c_void_p, c_i64, P(c_i64), P(c_i32), P(c_f64), P(c_f64), P(c_f64)]),
}

_lib = None


class KatanaHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkatana_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load the HIP library (once).  Fails loudly when it has not been built: there is no
    CPU path to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libkatana_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  The Katana HIP engine has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)      # AttributeError here == ABI symbol missing
            fn.restype = res
            fn.argtypes = args
        if L.ktn_sizeof_params() != C.sizeof(KtnParams) or L.ktn_sizeof_nlp_desc() != C.sizeof(KtnNlpDesc):
            raise RuntimeError("libkatana_hip.so was built from a different include/katana_hip.h than this binding mirrors "
                               "(ktn_params %d vs %d bytes, ktn_nlp_desc %d vs %d): rebuild the library" % (
                                   L.ktn_sizeof_params(), C.sizeof(KtnParams), L.ktn_sizeof_nlp_desc(), C.sizeof(KtnNlpDesc)))
        _lib = L
    return _lib


def check(handle, code):
    if code < 0:
        msg = lib().ktn_last_error(handle)
        raise KatanaHipError(code, msg.decode() if msg else "")
    return code
