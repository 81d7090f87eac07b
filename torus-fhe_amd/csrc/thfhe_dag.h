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

// gate-DAG executor plumbing: wires live in one device table [n_wires][words]; a level's operands are gathered into the
// contiguous staging arrays the bootstrap kernels read, its outputs scattered back
__global__ __launch_bounds__(256) void dag_gather_kernel(const int32_t *__restrict__ wires, const int32_t *__restrict__ idx, int32_t *__restrict__ dst,
                                                         long count, int words) {
    const long g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (g < count && i < words) dst[g * words + i] = wires[(size_t)idx[g] * words + i];
}
__global__ __launch_bounds__(256) void dag_scatter_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int32_t *__restrict__ wires,
                                                          long count, int words) {
    const long g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (g < count && i < words) wires[(size_t)idx[g] * words + i] = src[g * words + i];
}
// NOT / COPY gates of one sub-level (no gate of the launch reads another's output)
__global__ __launch_bounds__(256) void dag_wire_linear_kernel(int32_t *__restrict__ wires, const int32_t *__restrict__ in_idx,
                                                              const int32_t *__restrict__ out_idx, const int32_t *__restrict__ ops, long count, int words) {
    const long g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (g >= count || i >= words) return;
    const uint32_t v = (uint32_t)wires[(size_t)in_idx[g] * words + i];
    wires[(size_t)out_idx[g] * words + i] = (int32_t)(ops[g] == THFHE_NOT ? 0u - v : v);
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

}  // namespace

#endif  // THFHE_DAG_H
