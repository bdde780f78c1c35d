// dist.hip -- collectives of the row-sharded LP: RCCL, peer buffers over hipIpc, host callback  (struct Engine: engine.hpp)
#include "engine.hpp"
#include "launch.hpp"
#include "kernels.hpp"

namespace ktn {

// signal "my slot of this epoch is complete" to every rank and wait for theirs; returns the slot offset to read
int64_t Engine::ipc_barrier() {
    const int64_t off = ipc_off();
    dist.ipc.epoch += 1;
    hipLaunchKernelGGL(k_ipc_barrier, dim3(1), dim3(64), 0, stream, dist.ipc.P, dist.rank, dist.world, dist.ipc.epoch,
                       dist.ipc.timeout_ticks, dist.ipc.h_err_dev);
    return off;
}

void Engine::allreduce(double* d, size_t n, int op) {          // in place; op 0: sum, 1: max
    if (!row_sharded() || n == 0) return;
    stats["allreduce_calls"] += 1.0;
    stats["allreduce_bytes"] += 8.0 * (double)n;
    if (dist.ipc.on) {
        KTN_REQUIRE((int64_t)n <= dist.ipc.cap, "peer-buffer transport: vector longer than the exposed slot");
        size_t ea = 0, eb = 0;
        if (prm.profile) { ea = ev_get(); eb = ev_get(); KTN_HIP(hipEventRecord(ev_pool[ea], stream)); }
        if (d != ipc_slot()) KTN_HIP(hipMemcpyAsync(ipc_slot(), d, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        const int64_t off = ipc_barrier();
        if (op) hipLaunchKernelGGL((k_ipc_reduce<1>), dim3(ceil_div((int64_t)n, kBlock)), dim3(kBlock), 0, stream, (int64_t)n, dist.ipc.P, dist.world, off, d);
        else hipLaunchKernelGGL((k_ipc_reduce<0>), dim3(ceil_div((int64_t)n, kBlock)), dim3(kBlock), 0, stream, (int64_t)n, dist.ipc.P, dist.world, off, d);
        check_launch();
        if (prm.profile) { KTN_HIP(hipEventRecord(ev_pool[eb], stream)); ev_recs.push_back({3, ea, eb, 8.0 * (double)n}); }
    } else if (dist.comm) {
        size_t ea = 0, eb = 0;
        if (prm.profile) { ea = ev_get(); eb = ev_get(); KTN_HIP(hipEventRecord(ev_pool[ea], stream)); }
        const ncclResult_t r = ncclAllReduce(d, d, n, ncclDouble, op ? ncclMax : ncclSum, dist.comm, stream);
        if (r != ncclSuccess) throw Error(KTN_E_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        if (prm.profile) { KTN_HIP(hipEventRecord(ev_pool[eb], stream)); ev_recs.push_back({3, ea, eb, 8.0 * (double)n}); }
    } else {
        KTN_REQUIRE(dist.cb != nullptr, "row-sharded handle without a collective transport");
        dist.hbuf.resize(n);
        KTN_HIP(hipMemcpyAsync(dist.hbuf.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        if (dist.cb(dist.user, dist.hbuf.data(), (int64_t)n, op) != 0) throw Error(KTN_E_CALLBACK, "all-reduce callback failed");
        KTN_HIP(hipMemcpyAsync(d, dist.hbuf.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
        sync();
    }
}

// k values reduced over the ranks (host in, host out); every rank gets the identical result
void Engine::allreduce_host(double* v, int k, int op) {
    if (!row_sharded()) return;
    d_red.resize(64, stream);
    KTN_REQUIRE(k <= 64, "allreduce_host: too many values");
    KTN_HIP(hipMemcpyAsync(d_red.p, v, (size_t)k * sizeof(double), hipMemcpyHostToDevice, stream));
    allreduce(d_red.p, (size_t)k, op);
    KTN_HIP(hipMemcpyAsync(v, d_red.p, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    ipc_check();
}

void Engine::probe_fill(int64_t n, double* v, double value) {
    hipLaunchKernelGGL(k_probe_fill, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, stream, n, v, value);
}

// Leave the peer-buffer transport: one last barrier (after it no peer kernel of an earlier epoch can still be reading this
// rank's slots, and this rank reads nobody's), then unmap the peers' buffers and free the exposed ones.
void Engine::ipc_release() {
    auto& I = dist.ipc;
    if (I.on) {
        (void)hipSetDevice(device);
        I.timeout_ticks = std::max<long long>(I.timeout_ticks / 10, 1);       // (a peer that is gone already must not hold the teardown up)
        ipc_barrier();
        (void)hipStreamSynchronize(stream);
        I.on = false;
    }
    for (void*& p : I.opened) if (p) { (void)hipIpcCloseMemHandle(p); p = nullptr; }
    if (I.data) { (void)hipFree(I.data); I.data = nullptr; }
    if (I.flags) { (void)hipFree(I.flags); I.flags = nullptr; }
    if (I.h_err) { (void)hipHostFree(I.h_err); I.h_err = nullptr; I.h_err_dev = nullptr; }
}

}  // namespace ktn
