// hc_cli.cpp — hc-edgecalc, the edge-calculation stage of ViralQuasispecies as a stand-alone program: its body (hc_cli_main, in
// libhcedge.so; the executable cli/hc_edgecalc_main.cpp is a launcher that loads the library — or, with --resident, hands the
// command line to a resident process that has it loaded already: hc_cli_daemon below).
// It accepts the reference binary's complete flag surface (src/ViralQuasispecies.cpp:49-99,
// same names, short forms, defaults, `--x=v` and `--x v` spellings, same validation messages and
// exit codes, :103-154), runs the stages up to and including EdgeCalculator::construct_edges()
// (:233-293) on the MI355X, and stops there: graph cleaning, cliques, super-reads and FNO are
// outside this build's scope.  Outputs: nonedge_overlaps.txt (as the reference),
// viralquasispecies.log (settings block, :160-218), edges.tsv — the admitted edges in adjacency-list order as
// construct_edges leaves them, one line per Edge, score and mismatch rate in the shortest decimal form that reads back to the same bits
// (std::to_chars; until round 4 "%.17g": 0.1 was written 0.10000000000000001 — same value, other bytes) — and edges_sorted.tsv, the same
// after overlap_graph->sortEdges() (:297), the order every later stage of the reference sees.
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <sys/time.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../host/EdgeCalculator.h"

using namespace hc;

struct Opt {
    std::string name;
    char shortname;
    bool is_flag_without_value;
    std::function<bool(const std::string&)> set;
    std::string help;
};

static bool to_bool(const std::string& v, bool& out) {  // boost::program_options bool: true/false/1/0/yes/no/on/off
    std::string s;
    for (char c : v) s.push_back((char)tolower((unsigned char)c));
    if (s == "true" || s == "1" || s == "yes" || s == "on") { out = true; return true; }
    if (s == "false" || s == "0" || s == "no" || s == "off") { out = false; return true; }
    return false;
}

template <typename T>
static bool to_num(const std::string& v, T& out) {
    if (v.empty()) return false;
    char* end = nullptr;
    if (std::is_floating_point<T>::value) {
        const double d = strtod(v.c_str(), &end);
        if (*end) return false;
        out = (T)d;
    } else if (std::is_signed<T>::value) {
        const long long d = strtoll(v.c_str(), &end, 10);
        if (*end) return false;
        out = (T)d;
    } else {
        if (v[0] == '-') return false;
        const unsigned long long d = strtoull(v.c_str(), &end, 10);
        if (*end) return false;
        out = (T)d;
    }
    return true;
}

static bool g_device_fault = false;

static double now_s() {
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

// The program.  on_done (resident mode; nullptr: a process of its own): called exactly once, with the exit code, when every output file is
// written and closed — before the stage is torn down; the function then tears down in an orderly way and returns the code.  Without it the
// function leaves the process through _exit once the outputs are closed (see the end of the function).
extern "C" int hc_cli_main(int argc, char** argv, void (*on_done)(int code, void* arg), void* on_done_arg) {
    bool reported = false;
    auto done = [&](int code) {
        if (on_done && !reported) {
            fflush(stdout);
            fflush(stderr);
            on_done(code, on_done_arg);
        }
        reported = true;
        return code;
    };
    const double t_main = now_s();
    ProgramSettings ps;
    std::vector<Opt> opts;
    std::map<std::string, int> seen;
    auto S = [&](const char* n, char sh, std::string* p, const char* h) {
        opts.push_back({n, sh, false, [p](const std::string& v) { *p = v; return true; }, h});
    };
    auto B = [&](const char* n, char sh, bool* p, const char* h) {
        opts.push_back({n, sh, false, [p](const std::string& v) { return to_bool(v, *p); }, h});
    };
#define NUM(n, sh, p, h) opts.push_back({n, sh, false, [&](const std::string& v) { return to_num(v, p); }, h})
    opts.push_back({"help", 0, true, [](const std::string&) { return true; }, "produce help message"});
    S("fastq", 0, &ps.fastq_file, "path to fastq files: paired_1.fastq, paired_2.fastq and single.fastq");
    S("singles", 's', &ps.singles_file, "path to single-end read fastq file");
    S("paired1", 0, &ps.paired1_file, "path to paired-end read /1 fastq file");
    S("paired2", 0, &ps.paired2_file, "path to paired-end read /2 fastq file");
    S("overlaps", 0, &ps.overlaps_file, "path to overlap file");
    S("output", 'O', &ps.output_dir, "path to output files");
    S("IDs", 0, &ps.id_correspondence, "path to ID correspondence file");
    NUM("max_ov", 0, ps.max_overlaps, "set the maximum number of overlaps considered");
    NUM("max_reads", 0, ps.max_reads, "set the maximum number of reads used");
    NUM("threads", 't', ps.n_threads, "set the number of threads used");
    NUM("min_clique_size", 0, ps.min_clique_size, "set the minimum clique size for a superread");
    NUM("min_qual", 0, ps.min_qual, "set the minimum base quality for a superread");
    NUM("min_overlap_perc", 0, ps.min_overlap_perc, "set the minimum overlap percentage");
    NUM("min_overlap_len", 0, ps.min_overlap_len, "set the minimum overlap length (bp)");
    NUM("edge_threshold", 0, ps.edge_threshold, "set the minimal overlap score for creating an edge");
    NUM("ov_threshold", 0, ps.ov_threshold, "set the minimal overlap score for keeping non-edge overlap");
    B("allow_spaced_overlaps", 0, &ps.allow_spaces, "allow space-delimited overlaps instead of tabs");
    B("first_it", 0, &ps.first_it, "set to true when there is no subreads file");
    B("add_duplicates", 0, &ps.add_duplicates, "deal with reverse complements by adding duplicate vertices");
    B("resolve_orientations", 0, &ps.resolve_orientations, "deal with reverse complements by labelling vertices");
    NUM("keep_singletons", 0, ps.keep_singletons, "minimal read length for singletons not to be removed");
    B("error_correction", 0, &ps.error_correction, "only do error correction");
    B("cliques", 0, &ps.cliques, "clique-merging instead of edge-merging");
    B("ignore_inclusions", 0, &ps.ignore_inclusions, "ignore full inclusion edges in overlap graph");
    B("graph_only", 0, &ps.graph_only, "only do the graph construction");
    NUM("FNO", 0, ps.fno, "set the FindNextOverlaps function desired");
    NUM("original_readcount", 0, ps.original_readcount, "the number of original reads");
    NUM("mismatch", 0, ps.mismatch, "minimal score per position in overlap");
    B("optimize", 0, &ps.optimize, "optimize FNO by not reconsidering non-edge overlaps");
    B("no_inclusion_overlaps", 0, &ps.no_inclusions, "do not add full inclusion overlaps");
    NUM("merge_contigs", 0, ps.merge_contigs, "allow edge construction based on <merge_contigs> mismatch rate");
    B("remove_multi_occ", 0, &ps.remove_multi_occ, "remove clique nodes when used before");
    NUM("remove_trans", 0, ps.remove_trans, "remove (0) no, (1) single, (2) double, (3) triple transitive edges");
    B("remove_branches", 0, &ps.remove_branches, "remove branches from overlap graph");
    B("remove_tips", 0, &ps.remove_tips, "remove tips from overlap graph to reduce branching");
    NUM("min_read_len", 0, ps.min_read_len, "set the minimum read length (bp) for allowing edges");
    NUM("max_tip_len", 0, ps.max_tip_len, "set the maximum extension length for a node to be considered a tip");
    B("separate_tips", 0, &ps.store_tips_separately, "store tip-sequences in a separate file");
    S("base_path", 0, &ps.base_path, "set path to SAVAGE directory containing quick-cliques-1.0");
    B("diploid", 0, &ps.diploid, "apply edge filtering for diploid genomes");
    B("relax_PE_edges", 0, &ps.relax_PE_edges, "relax edge restrictions for paired-end overlaps");
    S("original_fastq", 0, &ps.original_fastq, "original reads for applying read-based branch reduction");
    B("branch_reduction", 0, &ps.branch_reduction, "read-based branch reduction");
    NUM("branch_SE_c", 0, ps.branch_SE_c, "number of single-end input reads in original fastq");
    NUM("branch_PE_c", 0, ps.branch_PE_c, "number of paired-end input reads in original fastq");
    B("careful_diploid", 0, &ps.careful, "more careful merging by avoiding neighboring components");
    B("verbose", 'v', &ps.verbose, "output additional information during assembly");
    NUM("device", 0, ps.device, "[hc-edgecalc] HIP device ordinal");
    NUM("device_mask", 0, ps.device_mask, "[hc-edgecalc] bit d set: score blocks on HIP device d too (0 = --device alone)");
    S("sfo", 0, &ps.sfo_file, "[hc-edgecalc] the SFO file of rust-overlaps IN PLACE of --overlaps: scripts/sfo2overlaps.py's ingest runs inside (on the device)");

    auto usage = [&]() {
        puts("Program options:");
        for (const Opt& o : opts) {
            std::string n = "  --" + o.name;
            if (o.shortname) n = std::string("  -") + o.shortname + " [ --" + o.name + " ]";
            if (!o.is_flag_without_value) n += " arg";
            printf("%-44s %s\n", n.c_str(), o.help.c_str());
        }
        puts("");
    };
    auto find = [&](const std::string& n, char sh) -> Opt* {
        for (Opt& o : opts)
            if ((!n.empty() && o.name == n) || (sh && o.shortname == sh)) return &o;
        return nullptr;
    };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], name, val;
        bool has_val = false;
        Opt* o = nullptr;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { val = a.substr(eq + 1); has_val = true; }
            o = find(name, 0);
        } else if (a.size() >= 2 && a[0] == '-') {
            o = find("", a[1]);
            name = a.substr(1, 1);
            if (a.size() > 2) { val = a.substr(2); has_val = true; }
        }
        if (!o) {
            fprintf(stderr, "unrecognised option '%s'\n", a.c_str());
            return done(1);
        }
        if (!o->is_flag_without_value && !has_val) {
            if (i + 1 >= argc) {
                fprintf(stderr, "the required argument for option '--%s' is missing\n", o->name.c_str());
                return done(1);
            }
            val = argv[++i];
        }
        if (!o->set(val)) {
            fprintf(stderr, "the argument ('%s') for option '--%s' is invalid\n", val.c_str(), o->name.c_str());
            return done(1);
        }
        seen[o->name]++;
    }
    // options with a default_value() count as present in the reference's variables_map (vm.count)
    for (const char* d : {"singles", "paired1", "paired2"}) seen[d]++;
    auto count = [&](const char* n) { return seen.count(n) ? seen[n] : 0; };

    if (count("help")) {  // src/ViralQuasispecies.cpp:104-107
        usage();
        return done(0);
    }
    if (!(count("fastq") || count("singles") || count("paired1") || count("paired2"))) {  // :111-115
        fputs("No fastq file(s) provided.\n\n", stderr);
        usage();
        return done(1);
    } else if (count("fastq") && (!ps.singles_file.empty() || !ps.paired1_file.empty() || !ps.paired2_file.empty())) {  // :116-120
        fputs("Cannot combine --fastq option with --singles, --paired1 or --paired2. \n\n", stderr);
        usage();
        return done(1);
    }
    if (count("overlaps") && count("sfo")) {
        fputs("--overlaps and --sfo are exclusive options, use one.\n\n", stderr);
        usage();
        return done(1);
    }
    if (!count("overlaps") && !count("sfo")) {  // :128-132
        fputs("No overlaps file provided.\n\n", stderr);
        usage();
        return done(1);
    }
    if (!count("original_readcount")) {  // :134-138
        fputs("No original readcount provided.\n\n", stderr);
        usage();
        return done(1);
    }
    if (ps.add_duplicates && ps.resolve_orientations) {  // :144-148
        fputs("Add duplicates and resolve orientations are exclusive options, use at most 1.\n\n", stderr);
        usage();
        return done(1);
    }
    if (ps.error_correction && !ps.cliques) {  // :150-154
        fputs("Error correction requires clique enumeration. Set --cliques=true.\n", stderr);
        usage();
        return done(1);
    }

    {  // settings block of viralquasispecies.log, :160-218 (the fields the edge-calculation stage reads)
        FILE* lf = fopen((ps.output_dir + "viralquasispecies.log").c_str(), "w");
        if (lf) {
            time_t raw;
            time(&raw);
            fprintf(lf, "%s\n\nInput:\n%s\n%s\n%s\n%s\n\n", ctime(&raw), ps.singles_file.c_str(), ps.paired1_file.c_str(),
                    ps.paired2_file.c_str(), (ps.overlaps_file.empty() ? ps.sfo_file : ps.overlaps_file).c_str());
            fprintf(lf, "Output directory: %s\nMaximum number of overlaps: %lu\nThreads: %u\n", ps.output_dir.c_str(),
                    ps.max_overlaps, ps.n_threads);
            fprintf(lf, "Minimal overlap percentage: %u\nMinimal overlap length: %u\nEdge threshold: %g\nOverlap threshold: %g\n",
                    ps.min_overlap_perc, ps.min_overlap_len, ps.edge_threshold, ps.ov_threshold);
            fprintf(lf, "Add duplicates: %d\nResolve read orientations: %d\nIgnore inclusions: %d\nMismatch prob: %g\n",
                    ps.add_duplicates, ps.resolve_orientations, ps.ignore_inclusions, ps.mismatch);
            fprintf(lf, "Merge contigs: %g\nMinimal read length: %u\nRelax PE edges: %d\nVerbose: %d\n", ps.merge_contigs,
                    ps.min_read_len, ps.relax_PE_edges, ps.verbose);
            fclose(lf);
        }
    }
    if (count("fastq")) {  // :226-230
        ps.singles_file = ps.fastq_file + "/singles.fastq";
        ps.paired1_file = ps.fastq_file + "/paired1.fastq";
        ps.paired2_file = ps.fastq_file + "/paired2.fastq";
    }
    try {
        double t0 = now_s();
        // the HIP runtime's start and the kernels' load (0.1 s of a process's first HIP calls) happen beside the FASTQ parsing when that lasts
        // long enough to hide them (64 MiB of FASTQ and more), as in hc_ec_open
        std::thread warm;
        struct JoinWarm {
            std::thread& t;
            ~JoinWarm() {
                if (t.joinable()) t.join();
            }
        } join_warm{warm};
        if (hc::warm_up_pays(ps)) warm = std::thread(hc::warm_device_code, hc::to_hc_settings(ps));
        auto fastq = std::make_shared<FastqStorage>(ps);  // :233
        if (ps.verbose) printf("FastqStorage ready! Construction took %g seconds.\n", now_s() - t0);
        t0 = now_s();
        auto graph = std::make_shared<OverlapGraph>(ps.add_duplicates ? 2 * fastq->get_readcount() : fastq->get_readcount(), fastq, ps);  // :246-261
        if (ps.verbose) puts("Adding vertices...");
        for (Read* r : fastq->m_read_vec) r->set_vertex_id(true, graph->addVertex(r->get_read_id()));  // :259-263
        if (ps.add_duplicates)  // a vertex for each reverse complementary read, :265-271
            for (Read* r : fastq->m_read_vec) r->set_vertex_id(false, graph->addVertex(r->get_read_id()));
        if (ps.verbose) {
            printf("Overlap graph ready! Construction took %g seconds.\n", now_s() - t0);
            printf("Number of vertices: %u\n", graph->getVertexCount());
        }
        // where a short run's time goes (verbose): the HIP runtime's start is paid by the first HIP call of a process — here what is
        // left of it once the reads are in memory
        t0 = now_s();
        if (warm.joinable()) warm.join();
        const int n_devices = hc_device_count();
        const double t_hip = now_s() - t0;
        t0 = now_s();
        EdgeCalculator calc(fastq, graph, ps);  // :279
        const double t_ctor = now_s() - t0;
        if (ps.verbose)
            printf("[hc-edgecalc] HIP runtime start %.3f s (%d device(s)), EdgeCalculator (contexts, read store, text blocks) %.3f s\n", t_hip, n_devices,
                   t_ctor);
        t0 = now_s();
        if (!ps.sfo_file.empty()) {  // savage.py:664-717's three steps in one: the SFO file -> the script's ingest -> construct_edges
            uint64_t n_rec = 0, n_lines = 0;
            int on_device = 0;
            calc.construct_edges_from_sfo(ps.sfo_file, false, &n_rec, &n_lines, &on_device);
            if (ps.verbose)
                printf("[hc-edgecalc] --sfo: %lu SFO records -> %lu overlap lines (%s)\n", (unsigned long)n_rec, (unsigned long)n_lines,
                       on_device ? "ingest on the device" : "ingest on the host");
        } else {
            calc.construct_edges();  // :281
        }
        const double dt = now_s() - t0;
        if (graph->getEdgeCount() == 0) {  // :284-291
            if (ps.verbose) puts("There were no edges constructed, so there is nothing to be done.");
            remove((ps.output_dir + "graph.txt").c_str());
            return done(0);
        } else if (ps.verbose) {
            printf("%u edges have been constructed in %g seconds.\n", graph->getEdgeCount(), dt);
            printf("[hc-edgecalc] parse %.3f s, score (H2D + kernel + D2H) %.3f s, insert %.3f s, write %.3f s; %lu candidates scored\n",
                   calc.stats.t_parse, calc.stats.t_score, calc.stats.t_insert, calc.stats.t_write,
                   (unsigned long)calc.stats.scored);
        }
        // The graph as text, one edge per line in list order.  Formatted by --threads threads, each its stretch of vertices into its own
        // buffer (fprintf with two %.17g per line took 40 ms of the SAVAGE example's 0.19 s process: a fifth of it); the doubles in their
        // shortest form that reads back to the same bits (std::to_chars).
        auto write_edges = [&](const char* name) {
            FILE* ef = fopen((ps.output_dir + name).c_str(), "w");
            if (!ef) return;
            const size_t V = graph->adj_out.size();
            const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(1u, ps.n_threads), (size_t)16, graph->getEdgeCount() / 4096 + 1}));
            std::vector<std::string> part(T);
            auto format = [&](unsigned t) {
                std::string& buf = part[t];
                char tmp[40];
                auto num = [&](long long v, char end) {
                    auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
                    buf.append(tmp, r.ptr);
                    buf.push_back(end);
                };
                auto unum = [&](unsigned long long v) {
                    auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
                    buf.append(tmp, r.ptr);
                    buf.push_back('\t');
                };
                auto dbl = [&](double v, char end) {
                    auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
                    buf.append(tmp, r.ptr);
                    buf.push_back(end);
                };
                for (size_t v = V * t / T; v < V * (t + 1) / T; v++)
                    for (const Edge& e : graph->adj_out[v]) {
                        unum(e.get_vertex(1));
                        unum(e.get_vertex(2));
                        unum(e.get_read(1)->get_read_id());
                        unum(e.get_read(2)->get_read_id());
                        num(e.get_pos(1), '\t');
                        num(e.get_pos(2), '\t');
                        num(e.get_extra_pos(1), '\t');
                        num(e.get_extra_pos(2), '\t');
                        buf.push_back(e.get_ori(1) ? '+' : '-');
                        buf.push_back('\t');
                        buf.push_back(e.get_ori(2) ? '+' : '-');
                        buf.push_back('\t');
                        buf.push_back(e.get_ord() ? e.get_ord() : '-');
                        buf.push_back('\t');
                        num(e.get_perc(), '\t');
                        num(e.get_len(0), '\t');
                        num(e.get_len(1), '\t');
                        num(e.get_len(2), '\t');
                        dbl(e.get_score(), '\t');
                        dbl(e.get_mismatch_rate(), '\n');
                    }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; t++) th.emplace_back(format, t);
            format(0);
            for (auto& x : th) x.join();
            for (const std::string& b : part) fwrite(b.data(), 1, b.size(), ef);
            fclose(ef);
        };
        double t_w0 = now_s();
        write_edges("edges.tsv");
        const double t_w1 = now_s();
        {  // overlap_graph->sortEdges(), :297
            std::vector<uint32_t> len(fastq->m_read_vec.size());
            for (size_t r = 0; r < len.size(); r++) len[r] = fastq->m_read_vec[r]->get_len();
            graph->sortEdges(len.data(), ps.n_threads);
        }
        const double t_w2 = now_s();
        write_edges("edges_sorted.tsv");
        if (ps.verbose) printf("[hc-edgecalc] edges.tsv %.3f s, sortEdges %.3f s, edges_sorted.tsv %.3f s\n", t_w1 - t_w0, t_w2 - t_w1, now_s() - t_w2);
        FILE* sf = fopen((ps.output_dir + "edgecalc_stats.txt").c_str(), "w");
        if (sf) {
            fprintf(sf, "vertex_count\t%u\nedge_count\t%u\ninclusion_count\t%u\ndup_count\t%u\nself_overlap_count\t%u\n",
                    graph->getVertexCount(), graph->getEdgeCount(), calc.inclusion_count, calc.dup_count, calc.self_overlap_count);
            fclose(sf);
        }
        if (ps.verbose) printf("[hc-edgecalc] %.3f s since main() started\n", now_s() - t_main);
        // Every output file is written and closed and the device is idle: leave without tearing the stage down (unpinning its text
        // buffers, freeing device memory, joining the worker threads and the HIP runtime's own exit handlers took 0.10 - 0.15 s of a
        // 0.3 - 0.4 s run on the SAVAGE example, profiles/r04_c1_process.json) — a pipeline starts this program once per iteration.
        // HC_CLI_TEARDOWN=1 keeps the orderly way out (sanitizer and leak-check runs).  What this relies on: every output of the
        // program is a FILE* opened and fclose'd in this function before this point, and the stage's clean-up thread (EdgeCalculator::
        // defer_cleanup) only gives memory back — nothing that writes may ever be handed to it.
        if (on_done) {
            done(0);  // the client leaves now; what is left of the stage goes behind its back (its devices stay: keep_devices_resident)
        } else if (!getenv("HC_CLI_TEARDOWN")) {
            fflush(stdout);
            fflush(stderr);
            _exit(0);
        }
    } catch (const FatalError& e) {  // every exit(1) / assert of the reference on this path
        fprintf(stderr, "%s\n", e.what.c_str());
        if (e.status == HC_ERR_HIP || e.status == HC_ERR_NO_DEVICE) g_device_fault = true;  // a resident process does not outlive a device fault
        return done(1);
    }
    return done(0);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The resident process (round 5).  A pipeline calls the binary once per stage and iteration (scripts/pipeline_per_stage.py:223-247,272-298:
// subprocess.check_call, strictly one after the other), and at the SAVAGE example's size two thirds of a call were the process's own
// start-up: the HIP runtime (0.05 - 0.06 s), the code object, contexts (profiles/r04_c1_process.json).  `hc-edgecalc --resident <the usual
// arguments>` (cli/hc_edgecalc_main.cpp) therefore only forwards its arguments, working directory, HC_* environment and its stdout / stderr
// DESCRIPTORS over a Unix socket to this loop, which runs hc_cli_main in a process that has all of that loaded, one request at a time, and
// sends the exit code back once the outputs are closed.  The process serves one user (socket in a 0700 directory, peer uid checked), leaves
// after `idle_s` seconds without a request, and leaves with a non-zero code — after answering — when a job met a device fault.
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>

extern char** environ;

namespace {
constexpr uint32_t kResidentMagic = 0x48435235u;  // "HCR5"

bool read_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) {
        const ssize_t k = read(fd, c, n);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            return false;
        }
        c += k;
        n -= (size_t)k;
    }
    return true;
}
bool write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        const ssize_t k = write(fd, c, n);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            return false;
        }
        c += k;
        n -= (size_t)k;
    }
    return true;
}
struct Reply {
    int fd;
    bool sent;
};
void send_code(int code, void* arg) {
    Reply* r = (Reply*)arg;
    if (!r || r->sent) return;
    r->sent = true;
    const int32_t c = code;
    (void)write_all(r->fd, &c, sizeof c);
}
}  // namespace

// Returns the process's exit code: 0 after an idle time-out or a stop request, 3 after a device fault, 2 when the socket cannot be served.
// lib_stamp: what libhcedge.so looked like when the launcher loaded it (mtime, size); a request that carries another stamp — the library was
// rebuilt under this process — is answered with kStaleLibrary and the process leaves: the client starts a new one.
extern "C" int hc_cli_daemon(const char* sock_path, int idle_s, unsigned long long lib_stamp) {
    const std::string path = sock_path;
    const std::string dir = path.substr(0, path.rfind('/'));
    {  // the HC_* environment of a job is its CLIENT's: what this process inherited from the client that started it does not stay behind
        std::vector<std::string> mine;
        for (char** e = environ; e && *e; e++)
            if (strncmp(*e, "HC_", 3) == 0 && strncmp(*e, "HC_RESIDENT_", 12) != 0) mine.push_back(std::string(*e, strcspn(*e, "=")));
        for (const std::string& n : mine) unsetenv(n.c_str());
    }
    hc::keep_devices_resident(true);  // contexts, text blocks and their page-locked buffers serve one request after the other
    signal(SIGPIPE, SIG_IGN);  // a client whose stdout is a closed pipe (`| head`) must not end the resident process
    // one resident process per socket: whoever holds the lock serves it
    // (the holder is either about to serve the socket — then there is nothing left to do here — or a resident process on its way out: a few
    // seconds of polite waiting decide which)
    const int lock = open((dir + "/lock").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    if (lock < 0) return 0;
    auto served = [&] {
        sockaddr_un probe;
        memset(&probe, 0, sizeof probe);
        probe.sun_family = AF_UNIX;
        const int ps = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
        if (ps < 0 || path.size() >= sizeof probe.sun_path) return false;
        strcpy(probe.sun_path, path.c_str());
        const bool yes = connect(ps, (sockaddr*)&probe, sizeof probe) == 0;
        close(ps);
        return yes;
    };
    bool mine = false;
    for (int k = 0; k < 500 && !mine; k++) {
        if (flock(lock, LOCK_EX | LOCK_NB) == 0) {
            mine = true;
        } else {
            if (served()) return 0;  // somebody else is the resident process
            timespec ts{0, 10000000};
            nanosleep(&ts, nullptr);
        }
    }
    if (!mine || served()) return 0;
    unlink(path.c_str());
    const int ls = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (ls < 0 || path.size() >= sizeof addr.sun_path) return 2;
    strcpy(addr.sun_path, path.c_str());
    if (bind(ls, (sockaddr*)&addr, sizeof addr) != 0 || listen(ls, 16) != 0) return 2;
    if (FILE* pf = fopen((dir + "/pid").c_str(), "w")) {  // who serves the socket (for the operator; tests tell one resident process from the next by it)
        fprintf(pf, "%ld\n", (long)getpid());
        fclose(pf);
    }
    // the HIP runtime and the kernels' code object, beside the first request's way here
    std::thread warm([] {
        if (hc_device_count() > 0) {
            hc_settings s;
            memset(&s, 0, sizeof s);
            s.edge_threshold = 0.97;
            hc::warm_device_code(s);
        }
    });
    int rc = 0;
    bool left = false;
    auto leave = [&] {  // stop serving: nobody connects to this process any more, a successor may take the lock while this one tears down
        if (left) return;
        left = true;
        close(ls);
        unlink(path.c_str());
        unlink((dir + "/pid").c_str());
        flock(lock, LOCK_UN);
        close(lock);
    };
    bool draining = false;
    for (;;) {
        pollfd pf{ls, POLLIN, 0};
        const int pr = poll(&pf, 1, draining ? 0 : (idle_s > 0 ? idle_s * 1000 : -1));
        if (pr == 0) {  // idle: the socket's NAME goes first (whoever comes now starts a successor, which waits for the lock), then the
            if (draining) break;  // clients that connected in the meantime are served — nobody is left un-accepted in the backlog
            unlink(path.c_str());
            draining = true;
            continue;
        }
        if (pr < 0) {
            if (errno == EINTR) continue;
            rc = 2;
            break;
        }
        const int cs = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
        if (cs < 0) continue;
        ucred cred;
        socklen_t cl = sizeof cred;
        if (getsockopt(cs, SOL_SOCKET, SO_PEERCRED, &cred, &cl) != 0 || cred.uid != getuid()) {
            close(cs);
            continue;
        }
        {  // a peer that connects and then says nothing must not hold the (single-threaded) loop: its request has five seconds to arrive
            timeval tv{5, 0};
            setsockopt(cs, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        }
        // header + the client's stdout and stderr (SCM_RIGHTS)
        uint32_t head[7];  // magic, argc, n_env, cwd bytes, flags (1 = stop), the client's library stamp (low, high)
        int fds[2] = {-1, -1};
        {
            iovec iov{head, sizeof head};
            char ctl[CMSG_SPACE(sizeof fds)];
            msghdr mh;
            memset(&mh, 0, sizeof mh);
            mh.msg_iov = &iov;
            mh.msg_iovlen = 1;
            mh.msg_control = ctl;
            mh.msg_controllen = sizeof ctl;
            const ssize_t k = recvmsg(cs, &mh, MSG_WAITALL | MSG_CMSG_CLOEXEC);
            bool ok = k == (ssize_t)sizeof head && head[0] == kResidentMagic;
            for (cmsghdr* cm = ok ? CMSG_FIRSTHDR(&mh) : nullptr; cm; cm = CMSG_NXTHDR(&mh, cm))
                if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS && cm->cmsg_len == CMSG_LEN(sizeof fds)) memcpy(fds, CMSG_DATA(cm), sizeof fds);
            if (!ok || (!(head[4] & 1u) && (fds[0] < 0 || fds[1] < 0))) {
                for (int f : fds)
                    if (f >= 0) close(f);
                close(cs);
                continue;
            }
        }
        if (!(head[4] & 1u) && (((unsigned long long)head[6] << 32) | head[5]) != lib_stamp) {
            const int32_t stale = -1000;  // kStaleLibrary (cli/hc_edgecalc_main.cpp)
            leave();
            (void)write_all(cs, &stale, sizeof stale);
            close(fds[0]);
            close(fds[1]);
            close(cs);
            break;
        }
        if (head[4] & 1u) {  // hc-edgecalc --resident_stop: the socket and the lock go BEFORE the answer — the client's next call starts a new process
            leave();
            const int32_t zero = 0;
            (void)write_all(cs, &zero, sizeof zero);
            close(cs);
            break;
        }
        auto read_str = [&](std::string& out) {
            uint32_t n = 0;
            if (!read_all(cs, &n, sizeof n) || n > (1u << 20)) return false;
            out.resize(n);
            return n == 0 || read_all(cs, &out[0], n);
        };
        std::vector<std::string> args(head[1]), env(head[2]);
        std::string cwd;
        bool ok = head[1] >= 1 && head[1] < 4096 && head[2] < 4096 && read_str(cwd);
        for (std::string& a : args) ok = ok && read_str(a);
        for (std::string& e : env) ok = ok && read_str(e);
        if (!ok) {
            close(fds[0]);
            close(fds[1]);
            close(cs);
            continue;
        }
        if (warm.joinable()) warm.join();
        // the job sees the client's working directory, HC_* environment, stdout and stderr
        std::vector<std::string> set_names;
        for (const std::string& e : env) {
            const size_t eq = e.find('=');
            if (eq == std::string::npos || e.compare(0, 3, "HC_") != 0) continue;
            setenv(e.substr(0, eq).c_str(), e.c_str() + eq + 1, 1);
            set_names.push_back(e.substr(0, eq));
        }
        fflush(stdout);
        fflush(stderr);
        const int keep1 = dup(1), keep2 = dup(2);
        dup2(fds[0], 1);
        dup2(fds[1], 2);
        close(fds[0]);
        close(fds[1]);
        int code = 1;
        Reply rp{cs, false};
        if (chdir(cwd.c_str()) != 0) {  // (the client's directory may have been removed, or be unreadable to nobody but it)
            fprintf(stderr, "hc-edgecalc --resident: cannot enter %s\n", cwd.c_str());
            send_code(1, &rp);
        } else {
            std::vector<char*> argv;
            for (std::string& a : args) argv.push_back(&a[0]);
            argv.push_back(nullptr);
            g_device_fault = false;
            try {
                code = hc_cli_main((int)args.size(), argv.data(), send_code, &rp);  // answers as soon as the outputs are closed, then tears down
            } catch (const std::exception& e) {  // (std::bad_alloc and the like: the job is lost, the resident process and its client's descriptors are not)
                fprintf(stderr, "hc-edgecalc --resident: %s\n", e.what());
                code = 1;
            } catch (...) {
                fprintf(stderr, "hc-edgecalc --resident: the job ended in an exception\n");
                code = 1;
            }
            if (!rp.sent) send_code(code, &rp);
        }
        fflush(stdout);
        fflush(stderr);
        dup2(keep1, 1);
        dup2(keep2, 2);
        close(keep1);
        close(keep2);
        close(cs);
        for (const std::string& n : set_names) unsetenv(n.c_str());
        (void)code;
        if (g_device_fault) {  // never go on (and never re-exec) in a process whose device has faulted
            rc = 3;
            break;
        }
    }
    leave();
    if (warm.joinable()) warm.join();
    hc::keep_devices_resident(false);
    return rc;
}
