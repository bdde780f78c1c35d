"""dev: mean launch duration of the LP step kernels (profile=1 events) against the LP size.  usage: dev_kernel_us.py n [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
for n in [int(a) for a in sys.argv[1:]] or [25000, 50000, 100000, 200000]:
    inst = ktn.instances.make_instance(n=n, m_nl=n // 10, k=32, family="explog", seed=0)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=1))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    st = m.optimize()
    kx, ky = m.stat("kx_time_s") / max(m.stat("kx_launches"), 1), m.stat("ky_time_s") / max(m.stat("ky_launches"), 1)
    print("n=%7d rows=%7d nnz=%8d %s  k_pdhg_x %.2f us (%.1f MB)  k_pdhg_y %.2f us (%.1f MB)" % (
        n, m.lp_num_rows(), int(m._lib.ktn_lp_nnz(m._h)), st, 1e6 * kx, m.stat("kx_bytes") / max(m.stat("kx_launches"), 1) / 1e6,
        1e6 * ky, m.stat("ky_bytes") / max(m.stat("ky_launches"), 1) / 1e6), flush=True)
