// hc_graph.h — launch interface of hc_graph_kernels.hip (duplicate resolution + adjacency on the device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcedge.h"
#include "hc_device.h"

namespace hc {

struct GraphParams {
    const ReadDesc* reads;  // the store's read descriptors (sequence lengths, paired flag)
    uint32_t n_reads;
    const uint32_t* vtx;  // vertex id of every read, nullptr = identity
    uint32_t n_vertices;
    uint32_t ignore_inclusions;  // --ignore_inclusions: mark OverlapGraph::inclusions (EdgeCalculator.cpp:459-468)
};

size_t graph_temp_bytes(uint32_t m, uint32_t V);
hipError_t graph_build_and_replay(const GraphParams& gp, const hc_admit_rec* A, uint32_t m, hc_edge_rec* E, uint64_t* key0,
                                  uint64_t* key1, uint32_t* idx0, uint32_t* idx1, uint8_t* keep, uint8_t* inclusions,
                                  unsigned long long* counters, uint32_t* survivors, unsigned long long* d_count, void* temp,
                                  size_t temp_bytes, hipStream_t s);
hipError_t graph_orders(const GraphParams& gp, const hc_edge_rec* E, const uint32_t* survivors, uint32_t n, uint32_t order,
                        uint32_t* k32a, uint32_t* k32b, uint64_t* k64a, uint64_t* k64b, uint32_t* tmp_idx, uint32_t* O_out,
                        uint32_t* O_in, unsigned long long* out_off, unsigned long long* in_off, uint8_t* tied, void* temp,
                        size_t temp_bytes, hipStream_t s);
hipError_t graph_select_tied(const uint8_t* tied, uint32_t V, uint32_t* out, unsigned long long* d_count, void* temp, size_t temp_bytes,
                             hipStream_t s);
hipError_t graph_gather(const hc_edge_rec* E, const uint32_t* O_out, const uint32_t* O_in, uint32_t n, hc_edge_rec* edges_out,
                        uint32_t* in_nodes, hipStream_t s);

}  // namespace hc
