// thfhe_dag.h -- the gate-DAG front end shared by the single-key and the 3-gen multi-key engines:
// an ASAP levelising scheduler for the reference's circuits (src/KNN_medical_data.cpp:127-489, J/3gen_mk_gates.jl:183-362), and the
// gather / scatter kernels of the device-resident executor.
#ifndef THFHE_DAG_H
#define THFHE_DAG_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"

namespace {
using namespace thfhe;

// gate-DAG executor plumbing.  Wires live in one device table [instances][n_wires][words]: `instances` independent evaluations of
// the same gate list (the reference's loop over test records around one circuit, src/KNN_medical_data.cpp:676-691).  The gates of a
// level are numbered G = q * cnt + g (instance q, gate g of the level); a launch handles the slice [first, first + total) of them: its
// operands are gathered into the contiguous staging arrays the bootstrap kernels read, its outputs scattered back.
__global__ __launch_bounds__(256) void dag_gather_kernel(const int32_t *__restrict__ wires, const int32_t *__restrict__ idx, int32_t *__restrict__ dst,
                                                         long first, long total, long cnt, size_t n_wires, int words,
                                                         const int32_t *__restrict__ ops, int32_t *__restrict__ ops_out) {
    const long j = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (j >= total || i >= words) return;
    const long G = first + j, q = G / cnt, g = G - q * cnt;
    dst[j * words + i] = wires[((size_t)q * n_wires + idx[g]) * words + i];
    if (ops_out && i == 0) ops_out[j] = ops[g];   // per-gate opcodes of the slice, in staging order
}
__global__ __launch_bounds__(256) void dag_scatter_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int32_t *__restrict__ wires,
                                                          long first, long total, long cnt, size_t n_wires, int words) {
    const long j = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (j >= total || i >= words) return;
    const long G = first + j, q = G / cnt, g = G - q * cnt;
    wires[((size_t)q * n_wires + idx[g]) * words + i] = src[j * words + i];
}
// NOT / COPY gates of one sub-level (no gate of the launch reads another's output), every instance
__global__ __launch_bounds__(256) void dag_wire_linear_kernel(int32_t *__restrict__ wires, const int32_t *__restrict__ in_idx,
                                                              const int32_t *__restrict__ out_idx, const int32_t *__restrict__ ops, long total, long cnt,
                                                              size_t n_wires, int words) {
    const long G = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (G >= total || i >= words) return;
    const long q = G / cnt, g = G - q * cnt;
    const size_t base = (size_t)q * n_wires;
    const uint32_t v = (uint32_t)wires[(base + in_idx[g]) * words + i];
    wires[(base + out_idx[g]) * words + i] = (int32_t)(ops[g] == THFHE_NOT ? 0u - v : v);
}


// One launch group of the schedule: `count` gates of one class whose operands are all available.
struct DagBatch {
    int32_t depth, sub, cls;  // cls: engine-defined gate class; 2 = NOT / COPY (no bootstrap)
    size_t off, count;        // index table slice: [ops | in0 | in1 | in2 | out], `count` entries each, at tab[off]
};
struct DagPlan {
    std::vector<DagBatch> batches;
    std::vector<int32_t> tab;
    size_t max_width = 0, max_rot = 0;
    int64_t rotations = 0;
    int32_t max_depth = 0;
    void fill_stats(int64_t *stats) const {
        stats[0] = max_depth, stats[1] = 0, stats[2] = rotations, stats[3] = (int64_t)max_width;
        for (const auto &b : batches) stats[1] += b.cls != 2;
    }
};

// ASAP schedule.  gates: int32[n_gates][4] = (opcode, in0, in1, in2) in topological order; gate g defines wire n_inputs + g.
// classify(op) -> class id (0 = two-input bootstrapped gate, 1 = MUX, 2 = NOT / COPY, 3 = three-input bootstrapped gate) or -1.
// Bootstrapped gates add one level; NOT / COPY ride on their operand's level as sub-levels (a NOT may read a NOT of the same depth).
template <typename Classify>
int dag_plan(const int32_t *gates, size_t n_inputs, size_t n_gates, Classify classify, DagPlan &plan) {
    const size_t n_wires = n_inputs + n_gates;
    if (n_wires > (size_t)INT32_MAX / 2) return thfhe_fail(THFHE_E_INVALID, "too many wires");
    std::vector<int32_t> depth(n_wires, 0), sub(n_wires, 0), cls(n_gates, 0);
    int32_t max_depth = 0;
    for (size_t g = 0; g < n_gates; g++) {
        const int32_t op = gates[4 * g], w = (int32_t)(n_inputs + g);
        const int k = classify(op);
        if (k < 0) return thfhe_fail(THFHE_E_INVALID, "gate opcode not defined for this engine");
        cls[g] = k;
        const int nin = k == 2 ? 1 : (k == 0 ? 2 : 3);
        int32_t d = 0, s = 0;
        for (int q = 0; q < nin; q++) {
            const int32_t in = gates[4 * g + 1 + q];
            if (in < 0 || in >= w) return thfhe_fail(THFHE_E_INVALID, "gate operand is not an earlier wire (gates must be in topological order)");
            if (depth[in] > d || (depth[in] == d && sub[in] > s)) d = depth[in], s = sub[in];
        }
        if (k == 2) s += 1; else d += 1, s = 0;
        depth[w] = d, sub[w] = s;
        if (d > max_depth) max_depth = d;
    }
    plan.max_depth = max_depth;
    // bucket: (depth, sub, class); bootstrapped classes first (sub 0), then the linear sub-levels in order
    std::vector<std::vector<std::vector<int32_t>>> boot(max_depth + 1, std::vector<std::vector<int32_t>>(4)), lin(max_depth + 1);
    for (size_t g = 0; g < n_gates; g++) {
        const int32_t w = (int32_t)(n_inputs + g);
        if (cls[g] == 2) {
            auto &L = lin[depth[w]];
            if ((int)L.size() < sub[w]) L.resize(sub[w]);
            L[sub[w] - 1].push_back((int32_t)g);
        } else {
            boot[depth[w]][cls[g]].push_back((int32_t)g);
        }
    }
    plan.tab.reserve(5 * n_gates);
    auto emit = [&](int32_t d, int32_t s, int32_t k, const std::vector<int32_t> &G) {
        if (G.empty()) return;
        DagBatch b{d, s, k, plan.tab.size(), G.size()};
        for (int col = 0; col < 5; col++)
            for (int32_t g : G) plan.tab.push_back(col == 4 ? (int32_t)(n_inputs + g) : (col == 0 ? gates[4 * g] : (gates[4 * g + col] < 0 ? 0 : gates[4 * g + col])));
        plan.batches.push_back(b);
        if (G.size() > plan.max_width) plan.max_width = G.size();
        const size_t rot = k == 2 ? 0 : (k == 1 ? 2 * G.size() : G.size());
        if (rot > plan.max_rot) plan.max_rot = rot;
        plan.rotations += (int64_t)rot;
    };
    for (int32_t d = 0; d <= max_depth; d++) {
        for (int32_t k : {0, 3, 1}) emit(d, 0, k, boot[d][k]);
        for (size_t q = 0; q < lin[d].size(); q++) emit(d, (int32_t)q + 1, 2, lin[d][q]);
    }
    return THFHE_OK;
}

// Device buffers of the executor (grow-only, owned by the engine's context and reused by every run on it).
struct DagBuffers {
    size_t cap_wires = 0, cap_tab = 0, cap_ops = 0, cap_pack = 0;   // bytes
    int32_t *d_wires = nullptr, *d_tab = nullptr, *d_ops = nullptr, *d_pack = nullptr;
    static hipError_t grow(int32_t *&p, size_t &cap, size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        (void)hipFree(p);
        p = nullptr, cap = 0;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        for (int32_t **p : {&d_wires, &d_tab, &d_ops, &d_pack}) (void)hipFree(*p), *p = nullptr;
        cap_wires = cap_tab = cap_ops = cap_pack = 0;
    }
};

// Device-resident executor.  Level by level, each class of a level as slices of at most `slice_cap` gates over ALL instances: gather ->
// run(cls, d_ops, n) (the engine's prologue + blind rotations + key switch from its staging arrays stage_in[0..2] into stage_out) ->
// scatter.  Nothing synchronises with the host between levels.
//   h_inputs  int32[instances][n_inputs][words]
//   h_sel     wire ids to return (n_sel of them) or null = every gate wire [n_inputs, n_wires)
//   h_out     int32[instances][n_sel or n_gates][words]
// ensure(max_gates_per_slice) sizes the engine's workspace and staging and returns its staging pointers through the out-parameters.
template <typename Ensure, typename Run>
int dag_execute(const DagPlan &plan, DagBuffers &B, hipStream_t stream, int words, size_t n_inputs, size_t n_gates, size_t instances,
                const int32_t *h_inputs, const int32_t *h_sel, size_t n_sel, int32_t *h_out, size_t slice_cap, Ensure ensure, Run run) {
    const size_t n_wires = n_inputs + n_gates;
    if (instances == 0 || n_gates == 0) return THFHE_OK;
    if (n_wires * instances > ((size_t)1 << 40) / (size_t)words) return thfhe_fail(THFHE_E_INVALID, "wire table too large");
    for (size_t s = 0; s < n_sel; s++)
        if (h_sel[s] < 0 || (size_t)h_sel[s] >= n_wires) return thfhe_fail(THFHE_E_INVALID, "output wire id out of range");
    const size_t widest = plan.max_width * instances, slice = widest < slice_cap ? widest : slice_cap;
    int32_t *stage_in[3] = {nullptr, nullptr, nullptr}, *stage_out = nullptr;
    int rc = ensure(slice ? slice : 1, stage_in, &stage_out);
    if (rc) return rc;
    const size_t rec = (size_t)words * sizeof(int32_t);
    hipError_t e = DagBuffers::grow(B.d_wires, B.cap_wires, instances * n_wires * rec);
    if (e == hipSuccess) e = DagBuffers::grow(B.d_tab, B.cap_tab, (plan.tab.size() + n_sel) * sizeof(int32_t));
    if (e == hipSuccess) e = DagBuffers::grow(B.d_ops, B.cap_ops, (slice ? slice : 1) * sizeof(int32_t));
    if (e == hipSuccess && h_sel) e = DagBuffers::grow(B.d_pack, B.cap_pack, instances * n_sel * rec);
    if (e != hipSuccess) return thfhe_fail_hip(e, "gate-DAG executor: device tables");
    int32_t *const d_wires = B.d_wires, *const d_tab = B.d_tab, *const d_sel = B.d_tab + plan.tab.size();
    if (n_inputs) e = hipMemcpy2DAsync(d_wires, n_wires * rec, h_inputs, n_inputs * rec, n_inputs * rec, instances, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab, plan.tab.data(), plan.tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && n_sel) e = hipMemcpyAsync(d_sel, h_sel, n_sel * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    rc = e == hipSuccess ? THFHE_OK : thfhe_fail_hip(e, "gate-DAG executor: upload");
    const unsigned wb = (unsigned)((words + 255) / 256);
    const dim3 block(256);
    for (size_t b = 0; b < plan.batches.size() && rc == THFHE_OK; b++) {
        const long cnt = (long)plan.batches[b].count, all = cnt * (long)instances;
        const int cls = plan.batches[b].cls;
        const int32_t *t_ops = d_tab + plan.batches[b].off, *t0 = t_ops + cnt, *t1 = t0 + cnt, *t2 = t1 + cnt, *t_out = t2 + cnt;
        if (cls == 2) {
            hipLaunchKernelGGL(dag_wire_linear_kernel, dim3((unsigned)all, wb), block, 0, stream, d_wires, t0, t_out, t_ops, all, cnt, n_wires, words);
            continue;
        }
        for (long first = 0; first < all && rc == THFHE_OK; first += (long)slice) {
            const long n = all - first < (long)slice ? all - first : (long)slice;
            const dim3 grid((unsigned)n, wb);
            hipLaunchKernelGGL(dag_gather_kernel, grid, block, 0, stream, d_wires, t0, stage_in[0], first, n, cnt, n_wires, words, t_ops, B.d_ops);
            hipLaunchKernelGGL(dag_gather_kernel, grid, block, 0, stream, d_wires, t1, stage_in[1], first, n, cnt, n_wires, words, nullptr, nullptr);
            if (cls != 0) hipLaunchKernelGGL(dag_gather_kernel, grid, block, 0, stream, d_wires, t2, stage_in[2], first, n, cnt, n_wires, words, nullptr, nullptr);
            rc = run(cls, B.d_ops, (size_t)n);
            if (!rc) hipLaunchKernelGGL(dag_scatter_kernel, grid, block, 0, stream, stage_out, t_out, d_wires, first, n, cnt, n_wires, words);
        }
    }
    if (rc == THFHE_OK) {
        e = hipGetLastError();
        if (e == hipSuccess && h_sel && n_sel) {
            const long all = (long)(n_sel * instances);
            hipLaunchKernelGGL(dag_gather_kernel, dim3((unsigned)all, wb), block, 0, stream, d_wires, d_sel, B.d_pack, 0L, all, (long)n_sel, n_wires, words, nullptr, nullptr);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(h_out, B.d_pack, instances * n_sel * rec, hipMemcpyDeviceToHost, stream);
        } else if (e == hipSuccess && !h_sel) {
            e = hipMemcpy2DAsync(h_out, n_gates * rec, d_wires + n_inputs * (size_t)words, n_wires * rec, n_gates * rec, instances, hipMemcpyDeviceToHost, stream);
        }
        if (e != hipSuccess) rc = thfhe_fail_hip(e, "gate-DAG executor");
    }
    e = hipStreamSynchronize(stream);
    if (rc == THFHE_OK && e != hipSuccess) rc = thfhe_fail_hip(e, "gate-DAG executor: sync");
    return rc;
}

}  // namespace

#endif  // THFHE_DAG_H
