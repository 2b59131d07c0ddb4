// Types.h — shared types of the host-side mirror of the reference's data model
// (reference src/Types.h:19-102).  Same names and meaning; own implementation.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>

namespace hc {

typedef unsigned long read_id_t;   // src/Types.h:93
typedef unsigned long node_id_t;   // src/Types.h:94

// The options of src/ViralQuasispecies.cpp:49-99 bound into one struct (src/Types.h:19-67).
// Fields the edge-calculation path does not read are carried only so that the CLI accepts
// the reference's full flag surface.
struct ProgramSettings {
    std::string fastq_file, singles_file, paired1_file, paired2_file, overlaps_file, output_dir, id_correspondence;
    unsigned long max_overlaps = 100000000;
    unsigned int n_threads = 1;
    unsigned long max_reads = 100000000;
    unsigned int min_clique_size = 4;
    double min_qual = 0.9;
    unsigned int min_overlap_perc = 0;
    unsigned int min_overlap_len = 150;
    double edge_threshold = 0.99;
    double ov_threshold = 0.9;
    bool allow_spaces = false;
    bool first_it = true;
    bool add_duplicates = false;
    bool resolve_orientations = true;
    unsigned int keep_singletons = 0;
    bool error_correction = false;
    bool cliques = false;
    bool graph_only = false;
    int fno = 2;
    unsigned long original_readcount = 0;
    bool ignore_inclusions = false;
    double mismatch = 0;
    bool optimize = true;
    bool no_inclusions = false;
    double merge_contigs = 0;
    bool remove_multi_occ = false;
    unsigned int remove_trans = 0;
    bool remove_branches = false;
    bool remove_tips = true;
    unsigned int min_read_len = 0;
    std::string base_path = ".";
    bool verbose = false;
    bool diploid = false;
    unsigned int max_tip_len = 150;
    bool store_tips_separately = true;
    bool relax_PE_edges = false;
    std::string original_fastq;
    bool branch_reduction = false;
    unsigned int branch_SE_c = 0;
    unsigned int branch_PE_c = 0;
    bool careful = true;
    int device = 0;  // build-owned addition: HIP device ordinal
    unsigned int device_mask = 0;  // build-owned addition: bit d = the stage scores blocks on device d too (0 = `device` alone)
    std::string sfo_file;          // build-owned addition (hc-edgecalc --sfo): the SFO file of rust-overlaps in the overlaps file's place
};

// src/Types.h:99-102: strtoul with base auto-detection ("0x..", leading 0 = octal, junk = 0)
inline read_id_t str_to_read_id(const std::string& s) { return strtoul(s.c_str(), nullptr, 0); }

// Thrown where the reference calls exit(1) or trips an assert: the C ABI turns it into a status code.
struct FatalError {
    int status;
    std::string what;
};

}  // namespace hc
