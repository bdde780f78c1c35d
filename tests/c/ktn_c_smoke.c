/* Plain-C caller of the drop-in boundary (include/katana_hip.h): what a cgo / ccall / JNI binding does.
 * Model: test/2d.jl 101_01 -- min -x - y  s.t.  x^2 + y^2 <= 1, x and y free -- as separable rows.
 * Exit codes: 0 solved and objective == -sqrt(2) to 1e-6; 77 no HIP device (CPU test tier); 1 anything else. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "katana_hip.h"

int main(void) {
    ktn_params p;
    ktn_default_params(&p);
    p.log_level = 0;
    ktn_handle h = NULL;
    int rc = ktn_create(&p, &h);
    if (rc == KTN_E_NODEVICE) { printf("no device\n"); return 77; }
    if (rc != KTN_OK) { printf("ktn_create failed: %d\n", rc); return 1; }

    /* one constraint row: 1*(x-0)^2 + 1*(y-0)^2 - 1 <= 0 */
    const int64_t rowptr[2] = {0, 2};
    const int32_t col[2] = {0, 1};
    const uint8_t row_kind[1] = {KTN_ROW_SEP}, row_linear[1] = {0};
    const double rconst[1] = {-1.0};
    const uint8_t atom_kind[2] = {KTN_ATOM_QUAD, KTN_ATOM_QUAD};
    const double p0[2] = {1.0, 1.0}, p1[2] = {0.0, 0.0};
    /* objective -x - y (linear) */
    const int32_t ocol[2] = {0, 1};
    const uint8_t okind[2] = {KTN_ATOM_LIN, KTN_ATOM_LIN};
    const double op0[2] = {-1.0, -1.0}, op1[2] = {0.0, 0.0};
    ktn_nlp_desc d;
    memset(&d, 0, sizeof d);
    d.num_var = 2; d.num_constr = 1;
    d.rowptr = rowptr; d.col = col; d.row_kind = row_kind; d.row_linear = row_linear; d.rconst = rconst;
    d.atom_kind = atom_kind; d.p0 = p0; d.p1 = p1;
    d.obj_linear = 1; d.obj_kind = KTN_ROW_SEP; d.obj_nnz = 2; d.obj_col = ocol; d.obj_atom_kind = okind;
    d.obj_p0 = op0; d.obj_p1 = op1;

    const double inf = INFINITY;
    const double l_var[2] = {-inf, -inf}, u_var[2] = {inf, inf}, l_con[1] = {-inf}, u_con[1] = {0.0};
    rc = ktn_loadproblem(h, 2, 1, l_var, u_var, l_con, u_con, KTN_MIN, &d);
    if (rc != KTN_OK) { printf("loadproblem: %d %s\n", rc, ktn_last_error(h)); return 1; }
    rc = ktn_optimize(h);
    if (rc < 0) { printf("optimize: %d %s\n", rc, ktn_last_error(h)); return 1; }
    double x[2] = {0.0, 0.0};
    ktn_get_solution(h, x, 2);
    const double obj = ktn_get_objval(h);
    printf("status %d objective %.9f x %.6f %.6f iterations %lld cuts %lld\n", ktn_get_status(h), obj, x[0], x[1],
           (long long)ktn_numiters(h), (long long)ktn_numcuts(h));
    const int ok = ktn_get_status(h) == KTN_STATUS_OPTIMAL && fabs(obj + sqrt(2.0)) <= 1e-6 &&
                   fabs(x[0] - sqrt(0.5)) <= 1e-3 && fabs(x[1] - sqrt(0.5)) <= 1e-3;
    ktn_destroy(h);
    return ok ? 0 : 1;
}
