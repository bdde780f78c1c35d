/* Host-side AddressSanitizer harness of the C-ABI layer (SURVEY.md section 5 "race detection / sanitizers": the reference
 * is serial Julia and has none; the build plan asks for a -fsanitize=address host harness).  Built by `make asan` against
 * libkatana_hip_asan.so (host code of the csrc units instrumented, device code untouched) and run on the CPU box, where
 * ktn_create refuses with KTN_E_NODEVICE: what runs under the sanitizer is the layer every binding goes through first --
 * parameter defaults, handle creation and its failure path, and the argument validation of every entry point (NULL
 * handles, NULL output pointers), none of which may read or write through what it was given.
 * Exit 0: every call answered as documented and the sanitizer stayed silent.  (Not for the GPU box: sanitizer runs are CPU-only.) */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "katana_hip.h"

static int fails = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) { printf("FAIL line %d: %s\n", __LINE__, #cond); ++fails; } \
    } while (0)

static int cb(void* user, double* buf, int64_t n, int32_t op) { (void)user; (void)buf; (void)n; (void)op; return 0; }

int main(void) {
    {   /* never initialise a GPU under the sanitizer (the GPU pool refuses such runs): with a device node present, bail out */
        FILE* kfd = fopen("/dev/kfd", "r");
        if (kfd) { fclose(kfd); printf("skipped: /dev/kfd present -- the ASan harness is for the CPU tier\n"); return 77; }
    }
    ktn_params p;
    memset(&p, 0xAB, sizeof p);
    ktn_default_params(&p);
    ktn_default_params(NULL);
    EXPECT(p.f_tol == 1e-6 && p.cut_coef_rng == 1e9 && p.log_level == 10 && p.iter_cap == 10000 && p.obj_eps == -1.0);   /* src/solver.jl:34-43 */
    EXPECT(p.epi_shift == 1 && p.polish_max_iter == 30 && p.obj_cert_tol == 1e-6 && p.lp_mid_max_var == 512);               /* the last fields: the whole struct was written */
    EXPECT(ktn_abi_version() == KTN_ABI_VERSION);
    EXPECT(ktn_sizeof_params() == (int64_t)sizeof(ktn_params) && ktn_sizeof_nlp_desc() == (int64_t)sizeof(ktn_nlp_desc));

    EXPECT(ktn_create(&p, NULL) == KTN_E_INVALID);
    ktn_handle h = (ktn_handle)0x1;
    int rc = ktn_create(&p, &h);
    if (rc == KTN_OK) {                                                 /* a device is visible: nothing more to do here */
        ktn_destroy(h);
        printf("device present: the sanitizer harness is for the CPU tier\n");
        return fails ? 1 : 77;
    }
    EXPECT(rc == KTN_E_NODEVICE && h == NULL);                          /* the failure path frees what it allocated */
    rc = ktn_create(NULL, &h);                                          /* NULL params = defaults */
    EXPECT(rc == KTN_E_NODEVICE && h == NULL);

    /* every entry point with a NULL handle: an error code (or a neutral value), never a dereference */
    double d8[8]; int64_t i8[8]; int32_t i4[8]; char uid[128];
    ktn_nlp_desc desc; memset(&desc, 0, sizeof desc);
    ktn_destroy(NULL);
    EXPECT(ktn_last_error(NULL) != NULL);
    EXPECT(ktn_loadproblem(NULL, 1, 0, d8, d8, d8, d8, KTN_MIN, &desc) == KTN_E_INVALID);
    EXPECT(ktn_optimize(NULL) == KTN_E_INVALID && ktn_optimize_begin(NULL) == KTN_E_INVALID && ktn_optimize_end(NULL) == KTN_E_INVALID);
    EXPECT(ktn_ecp_step(NULL, i4) == KTN_E_INVALID && ktn_reset(NULL) == KTN_E_INVALID);
    EXPECT(ktn_get_status(NULL) == KTN_E_INVALID && isnan(ktn_get_objval(NULL)) && ktn_get_num_var(NULL) < 0);
    EXPECT(ktn_get_solution(NULL, d8, 8) == KTN_E_INVALID && ktn_numiters(NULL) < 0 && ktn_numcuts(NULL) < 0);
    EXPECT(!(ktn_get_solvetime(NULL) > 0.0));
    EXPECT(ktn_setwarmstart(NULL, d8, 8) == KTN_E_INVALID);
    EXPECT(ktn_sep_precompute(NULL, d8, 8) == KTN_E_INVALID && ktn_sep_num_constr(NULL) < 0 && ktn_sep_jac_nnz(NULL) < 0);
    EXPECT(ktn_sep_get_g(NULL, d8, 8) == KTN_E_INVALID && ktn_sep_get_jac(NULL, d8, 8) == KTN_E_INVALID);
    EXPECT(ktn_sep_get_structure(NULL, i8, i4) == KTN_E_INVALID && ktn_sep_isconstrsat(NULL, 0, 0.0, 0.0, 1e-6) == KTN_E_INVALID);
    i8[0] = 8;
    EXPECT(ktn_sep_gencut(NULL, 0, i4, d8, i8, d8) == KTN_E_INVALID && ktn_sep_sweep(NULL, 1e-6, i8, d8) == KTN_E_INVALID);
    EXPECT(ktn_lp_num_rows(NULL) < 0 && ktn_lp_nnz(NULL) < 0 && ktn_lp_get_rows(NULL, i8, i4, d8, d8, d8) == KTN_E_INVALID);
    EXPECT(ktn_lp_get_objective(NULL, d8, 8, d8) == KTN_E_INVALID && ktn_lp_get_duals(NULL, d8, 8) == KTN_E_INVALID);
    EXPECT(ktn_lp_solve(NULL, 1e-6, 1e-6, i4, i8) == KTN_E_INVALID && ktn_lp_pdhg_raw(NULL, d8, d8, 1.0, 1.0, 1, d8, d8) == KTN_E_INVALID);
    EXPECT(ktn_num_lp_sols(NULL) < 0 && ktn_get_lp_sol(NULL, 0, d8, 8) == KTN_E_INVALID);
    EXPECT(!(ktn_get_stat(NULL, "pdhg_iters") > 0.0) && !(ktn_get_stat(NULL, NULL) > 0.0));
    EXPECT(ktn_sweep_lp_point(NULL, 1e-6, i8, d8) == KTN_E_INVALID && ktn_lp_nnz_from(NULL, 0) < 0);
    EXPECT(ktn_lp_get_rows_from(NULL, 0, i8, i4, d8, d8, d8) == KTN_E_INVALID && ktn_lp_truncate(NULL, 0) == KTN_E_INVALID);
    EXPECT(ktn_lp_enable_global_lists(NULL, 4) == KTN_E_INVALID && ktn_last_sweep_slots(NULL, i8, 8, i8) == KTN_E_INVALID);
    EXPECT(ktn_lp_append_rows_nl(NULL, 0, i8, i4, d8, d8, d8, i8) == KTN_E_INVALID && ktn_lp_append_rows(NULL, 0, i8, i4, d8, d8, d8) == KTN_E_INVALID);
    EXPECT(ktn_set_cut_exchange(NULL, NULL, NULL, 0) == KTN_E_INVALID);
    EXPECT(ktn_lp_purge(NULL, i8) == KTN_E_INVALID && ktn_set_blocks(NULL, 1, i8) == KTN_E_INVALID && ktn_optimize_blocks(NULL, 0) == KTN_E_INVALID);
    EXPECT(ktn_dist_unique_id(NULL) == KTN_E_INVALID);
    memset(uid, 0, sizeof uid);
    EXPECT(ktn_dist_init_rccl(NULL, uid, 0, 1) == KTN_E_INVALID && ktn_dist_init_callback(NULL, 0, 1, cb, NULL) == KTN_E_INVALID);
    EXPECT(ktn_dist_ipc_export(NULL, 0, 2, 64, uid) == KTN_E_INVALID && ktn_dist_init_ipc(NULL, 0, 2, uid) == KTN_E_INVALID);
    EXPECT(ktn_dist_allreduce_probe(NULL, 8, 1, d8, d8) == KTN_E_INVALID);
    EXPECT(ktn_objective_certificate(NULL, 0, d8) == KTN_E_INVALID);
    EXPECT(ktn_lp_pack_rows_dev(NULL, 0, 0, d8, 8, i8, i8) == KTN_E_INVALID && ktn_lp_append_packed_dev(NULL, 0, 0, d8) == KTN_E_INVALID);
    printf("%s (%d failures)\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
