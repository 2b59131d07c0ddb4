// EdgeCalculator.cpp — construct_edges / process_overlaps of the reference (src/EdgeCalculator.cpp:389-666) with the
// OpenMP scoring loop replaced by the HIP path behind include/hcedge.h, and — into an empty graph, every pipeline call —
// the serial half (duplicate resolution with the reference's tie-break chain, adjacency lists) by the device's
// resolution of all admitted candidates at once (hc_graph_resolve).  nonedge_overlaps.txt bookkeeping and the per-edge
// insert into a graph that already holds edges stay on the host.
#include "EdgeCalculator.h"
#include "NumaBind.h"

#include <sched.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <type_traits>

struct hc_ctx;
int hc_found_to_overlaps_text(hc_ctx* c, uint64_t num_singles, uint64_t num_pairs, std::string& text, uint64_t* n_lines);  // hc_api_finder.cpp
namespace hc {
bool sfo_text_to_records(const char* sfo_text, size_t sfo_bytes, std::vector<hc_sfo_rec, DefaultInitAllocator<hc_sfo_rec>>& recs);  // Sfo2Overlaps.cpp
std::string sfo_to_overlaps(const char* sfo_text, size_t sfo_bytes, long ns, long np, uint64_t& n_lines);  // Sfo2Overlaps.cpp
}

namespace hc {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void check(int status, const char* where) {
    if (status != HC_OK) throw FatalError{status, std::string(where) + ": " + hc_strerror(status) + " " + hc_last_error()};
}

// The devices of the stage: HC_DEVICE_LIST ("0,1,1": a test knob — an ordinal may repeat, giving several contexts on
// one device), else the bits of --device_mask, else --device alone.
static std::vector<int> device_list(const ProgramSettings& ps) {
    std::vector<int> d;
    if (const char* e = getenv("HC_DEVICE_LIST")) {
        for (const char* p = e; *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            d.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    } else if (ps.device_mask) {
        for (int b = 0; b < 32; b++)
            if (ps.device_mask & (1u << b)) d.push_back(b);
    }
    if (d.empty()) d.push_back(ps.device);
    return d;
}

void EdgeCalculator::bind_here() const { bind_thread_to(m_node_cpus); }

struct EdgeCalculator::Appender {
    hc_ctx* ctx;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<const hc_admit_rec*, size_t>> q;
    bool no_more = false;
    FatalError error{0, ""};
    std::atomic<bool> failed{false};  // consume_block looks here: a failed append stops the stage at the block it is at, not after the whole file
    Appender(hc_ctx* c, std::function<void()> on_start) : ctx(c) {
        th = std::thread([this, on_start] {
            on_start();
            for (;;) {
                std::pair<const hc_admit_rec*, size_t> job;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return !q.empty() || no_more; });
                    if (q.empty()) return;
                    job = q.front();
                    q.pop_front();
                }
                if (error.status) continue;  // drain
                const int rc = hc_graph_append(ctx, job.first, job.second);
                if (rc != HC_OK) {
                    error = FatalError{rc, std::string("hc_graph_append: ") + hc_strerror(rc) + " " + hc_last_error()};
                    failed.store(true, std::memory_order_release);
                }
            }
        });
    }
    void push(const hc_admit_rec* p, size_t n) {
        {
            std::lock_guard<std::mutex> g(mu);
            q.emplace_back(p, n);
        }
        cv.notify_one();
    }
    void finish() {
        {
            std::lock_guard<std::mutex> g(mu);
            no_more = true;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
    ~Appender() { finish(); }
};

EdgeCalculator::EdgeCalculator(std::shared_ptr<FastqStorage> fastq, std::shared_ptr<OverlapGraph> graph,
                               const ProgramSettings& ps)
    : program_settings(ps), fastq_storage(std::move(fastq)), overlap_graph(std::move(graph)) {
    if (const char* m = getenv("HC_INSERT_MODE")) m_serial_insert = std::string(m) == "serial";
    if (const char* m = getenv("HC_RESOLVE")) m_host_resolve = std::string(m) == "host";
    if (const char* m = getenv("HC_PARSE")) m_host_parse = std::string(m) == "host";
    if (const char* m = getenv("HC_PARSE_FALLBACK")) {  // "block": a line the device does not read sends its whole block to the host (round 4's route)
        if (std::string(m) == "block") m_odd_line_cap = 0;
        else if (atoi(m) > 0) m_odd_line_cap = (uint32_t)atoi(m);  // test knob: entries of a block's list
    }
    if (const char* m = getenv("HC_TEXT_BLOCK")) m_text_block = std::max<size_t>(4096, (size_t)strtoull(m, nullptr, 10));
    m_cs = to_hc_settings(ps);
    if (const char* m = getenv("HC_TEXT_DEPTH")) {
        m_text_depth = std::max<size_t>(1, (size_t)atoi(m));
    } else {  // as deep as the overlaps file has blocks for one device, between 2 and 10: a small file does not pay for buffers it never fills
        struct stat sb;
        const size_t n_dev = std::max<size_t>(1, device_list(ps).size());
        if (stat(ps.overlaps_file.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
            // a file shorter than one block: the blocks are made for ITS size (page-locking and device buffers for 16 MiB took 38 ms of
            // the SAVAGE example's process, whose file has 3 MB); a longer text on a later call is cut into more blocks of this size
            if (!getenv("HC_TEXT_BLOCK") && (size_t)sb.st_size < m_text_block)
                m_text_block = std::max<size_t>((size_t)64 << 10, (((size_t)sb.st_size + 4096) + 65535) & ~(size_t)65535);
            const size_t blocks = ((size_t)sb.st_size + m_text_block - 1) / m_text_block;
            m_text_depth = std::min<size_t>(m_text_depth, std::max<size_t>(2, (blocks + n_dev - 1) / n_dev));
        }
    }
    m_node_cpus = cpus_near_device(m_cs.device);
    if (ps.n_threads > 1) m_pool.reset(new WorkerPool(ps.n_threads - 1, [this] { bind_here(); }));
    const FastqStorage& f = *fastq_storage;
    // The resident process (keep_devices_resident): the contexts, text blocks and small blocks of the stage before this one are taken over —
    // their streams, events, device scratch and page-locked buffers are what a stage's set-up and tear-down spend their time on — when
    // that stage ran on the same devices with blocks of text at least as large; hc_reset gives every context the new settings.
    std::vector<Device> parked;
    {
        std::lock_guard<std::mutex> g(g_park.mu);
        const std::vector<int> want = device_list(ps);
        if (!m_host_parse && !g_park.devices.empty() && g_park.device_ids == want && g_park.text_block >= m_text_block && g_park.odd_line_cap == m_odd_line_cap) {
            parked = std::move(g_park.devices);
            m_text_block = g_park.text_block;
            m_odd_blk = g_park.odd_blk;
            m_odd_blk_cap = g_park.odd_blk_cap;
            g_park.odd_blk = nullptr;
        } else {
            park_destroy_locked();
        }
        g_park.devices.clear();
    }
    try {
        size_t n_dev_done = 0;
        for (int d : device_list(ps)) {  // the read store is replicated: candidates are independent given the reads
            hc_settings cs = m_cs;
            cs.device = d;
            Device dev;
            const double tc0 = now_s();
            if (n_dev_done < parked.size()) {
                dev = parked[n_dev_done];
                parked[n_dev_done] = Device();  // (ours now: the catch block below destroys what m_dev holds)
                m_dev.push_back(dev);
                check(hc_reset(dev.ctx, &cs), "hc_reset");
            } else {
                check(hc_create(&dev.ctx, &cs), "hc_create");
                dev.device_id = d;
                m_dev.push_back(dev);
            }
            n_dev_done++;
            const double tc1 = now_s();
            // the blocks of text construct_edges streams the overlaps file through, with their page-locked buffers (device memory
            // and page-locking belong to setting the stage up, like the read store), are made by a second thread while this one
            // uploads the reads: page-locking is the driver's work on the host, the upload is the copy engine's
            Device& dv = m_dev.back();
            std::string blocks_error;
            int blocks_rc = HC_OK;
            double blocks_s = 0;
            std::thread blocks;
            struct Join {
                std::thread& t;
                ~Join() {
                    if (t.joinable()) t.join();
                }
            } join_blocks{blocks};
            if (!m_host_parse) {
                if (dv.tblk.size() < m_text_depth) dv.tblk.resize(m_text_depth, nullptr);
                blocks = std::thread([&] {
                    bind_here();
                    const double tb = now_s();
                    for (hc_textblock*& b : dv.tblk) {
                        if (b) continue;  // taken over from the stage before
                        blocks_rc = hc_textblock_create(dv.ctx, m_text_block, &b);
                        if (blocks_rc == HC_OK && m_odd_line_cap) blocks_rc = hc_textblock_list_nonplain(b, m_odd_line_cap);
                        if (blocks_rc == HC_OK && !hc_textblock_buffer(b)) {
                            blocks_rc = HC_ERR_NOMEM;
                            blocks_error = "EdgeCalculator: no page-locked buffer for a block of text";
                        } else if (blocks_rc != HC_OK) {
                            blocks_error = std::string("hc_textblock_create: ") + hc_strerror(blocks_rc) + " " + hc_last_error();
                        }
                        if (blocks_rc != HC_OK) break;
                    }
                    // the small block the odd lines of a text block are scored in (score_odd_lines), beside the upload as well: its eight
                    // allocations are a millisecond or two the first file would otherwise pay in the middle of the stage
                    if (blocks_rc == HC_OK && m_odd_line_cap && m_dev.size() == 1 && !m_odd_blk) {  // (the first device's context)
                        m_odd_blk_cap = 4096;
                        blocks_rc = hc_block_create(dv.ctx, m_odd_blk_cap, &m_odd_blk);
                        if (blocks_rc != HC_OK) blocks_error = std::string("hc_block_create: ") + hc_strerror(blocks_rc) + " " + hc_last_error();
                    }
                    blocks_s = now_s() - tb;
                });
            }
            check(hc_set_reads(dev.ctx, f.bases().data(), f.quals().data(), f.seq_off().data(), f.read_first_seq().data(),
                               f.get_readcount()),
                  "hc_set_reads");
            const double tc2 = now_s();
            double tc3 = tc2;
            if (!m_host_parse) {  // the device's text parser looks read ids up itself
                std::vector<uint64_t> ids(f.m_read_vec.size());
                for (size_t r = 0; r < ids.size(); r++) ids[r] = f.m_read_vec[r]->get_read_id();
                check(hc_text_set_ids(dev.ctx, ids.data(), (uint32_t)ids.size()), "hc_text_set_ids");
                tc3 = now_s();
                blocks.join();
                if (blocks_rc != HC_OK) throw FatalError{blocks_rc, blocks_error};
                if (getenv("HC_STAGE_TIMING"))
                    fprintf(stderr, "[hc stage] device %d: context %.3f s, read store %.3f s, id table %.3f s, %zu text blocks %.3f s beside them (+ %.3f s)\n", d, tc1 - tc0,
                            tc2 - tc1, tc3 - tc2, dv.tblk.size(), blocks_s, now_s() - tc3);
            }
        }
    } catch (...) {
        hc_block_destroy(m_odd_blk);
        for (std::vector<Device>* set : {&m_dev, &parked})
            for (Device& d : *set) {
                for (hc_block* b : d.blk) hc_block_destroy(b);
                for (hc_textblock* b : d.tblk) hc_textblock_destroy(b);
                hc_destroy(d.ctx);
            }
        throw;
    }
    m_ctx = m_dev[0].ctx;
}

// keep_devices_resident: what the last stage of this process leaves behind for the next one
EdgeCalculator::Park EdgeCalculator::g_park;

void EdgeCalculator::park_destroy_locked() {
    hc_block_destroy(g_park.odd_blk);
    g_park.odd_blk = nullptr;
    for (Device& d : g_park.devices) {
        for (hc_block* b : d.blk) hc_block_destroy(b);
        for (hc_textblock* b : d.tblk) hc_textblock_destroy(b);
        hc_destroy(d.ctx);
    }
    g_park.devices.clear();
    g_park.device_ids.clear();
}

void keep_devices_resident(bool on) {
    std::lock_guard<std::mutex> g(EdgeCalculator::g_park.mu);
    EdgeCalculator::g_park.keep = on;
    if (!on) EdgeCalculator::park_destroy_locked();
}

EdgeCalculator::~EdgeCalculator() {
    if (m_cleanup.joinable()) m_cleanup.join();
    {
        std::lock_guard<std::mutex> g(g_park.mu);
        if (g_park.keep && !m_host_parse && !m_dev.empty()) {  // the next stage of this process takes them over (constructor)
            park_destroy_locked();
            g_park.devices = std::move(m_dev);
            m_dev.clear();
            for (const Device& d : g_park.devices) g_park.device_ids.push_back(d.device_id);
            g_park.text_block = m_text_block;
            g_park.odd_line_cap = m_odd_line_cap;
            g_park.odd_blk = m_odd_blk;
            g_park.odd_blk_cap = m_odd_blk_cap;
            m_odd_blk = nullptr;
        }
    }
    hc_block_destroy(m_odd_blk);
    for (Device& d : m_dev) {
        for (hc_block* b : d.blk) hc_block_destroy(b);
        for (hc_textblock* b : d.tblk) hc_textblock_destroy(b);
        hc_destroy(d.ctx);
    }
}

// Work that only gives memory back: run behind the caller's back, one piece after the other
void EdgeCalculator::defer_cleanup(std::function<void()> work) {
    std::thread prev = std::move(m_cleanup);
    m_cleanup = std::thread([p = std::make_shared<std::thread>(std::move(prev)), w = std::move(work)]() mutable {
        if (p->joinable()) p->join();
        w();
    });
}

double EdgeCalculator::phred_to_prob(int phred) const { return pow(10, -phred / 10.0); }  // :59-63

double EdgeCalculator::overlap_score(const std::string& seq1, const std::string& seq2, const std::string& score1,
                                     const std::string& score2, unsigned int pos, double& mismatch_rate) {
    // a scratch context holding the two strings as two single-end reads
    if (seq1.empty() || seq2.empty() || seq1.size() != score1.size() || seq2.size() != score2.size())
        throw FatalError{HC_ERR_ARG, "overlap_score: empty or mismatched strings"};
    hc_settings cs = m_cs;
    hc_ctx* ctx = nullptr;
    check(hc_create(&ctx, &cs), "hc_create");
    std::string bases = seq1 + seq2, quals = score1 + score2;
    const uint64_t off[3] = {0, seq1.size(), seq1.size() + seq2.size()};
    const uint32_t first[3] = {0, 1, 2};
    hc_overlap_rec rec;
    memset(&rec, 0, sizeof rec);
    rec.read1 = 0; rec.read2 = 1; rec.pos1 = pos; rec.ori1 = rec.ori2 = 1; rec.ord = '-';
    hc_result_rec res;
    int st = hc_set_reads(ctx, (const uint8_t*)bases.data(), (const uint8_t*)quals.data(), off, first, 2);
    if (st == HC_OK) st = hc_score_batch(ctx, &rec, 1, &res);
    hc_destroy(ctx);
    check(st, "overlap_score");
    double score;
    uint32_t cls;
    check(hc_finalize(&cs, &res, &score, &mismatch_rate, &cls), "hc_finalize");
    return score;
}

void EdgeCalculator::collect_read_info() {
    const FastqStorage& f = *fastq_storage;
    const std::vector<uint32_t>& first = f.read_first_seq();  // lengths straight from the flat layout (Read::get_seq_len goes there too)
    const std::vector<uint64_t>& off = f.seq_off();
    const size_t n = f.m_read_vec.size();
    m_read_info.resize(n);
    auto fill = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            Read* r = f.m_read_vec[i];
            ReadInfo& x = m_read_info[i];
            const uint32_t q = first[i];
            x.read = r;
            x.paired = r->is_paired();
            x.len_a = (uint32_t)(off[q + 1] - off[q]);
            x.len_b = x.paired ? (uint32_t)(off[q + 2] - off[q + 1]) : 0;
            x.vertex_set = r->has_vertex_id(true);
            x.vertex = x.vertex_set ? r->get_vertex_id(true) : 0;
            x.vertex_rev_set = r->has_vertex_id(false);
            x.vertex_rev = x.vertex_rev_set ? r->get_vertex_id(false) : 0;
            x.pad = 0;
        }
    };
    const unsigned T = n < (1u << 16) ? 1u : std::max(1u, std::min<unsigned>(program_settings.n_threads, 8u));
    if (T == 1) {
        fill(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(fill, n * t / T, n * (t + 1) / T);
    fill(0, n / T);
    for (auto& x : th) x.join();
}

// The Edge as compute_overlap builds it, :219-232 / :254-270 / :292-308 / :353-379 (the device's edge_build_kernel
// states the same arithmetic for the bulk path).
Edge edge_from_admit(const hc_admit_rec& o, const ReadInfo* read_info, bool add_duplicates) {
    const ReadInfo& i1 = read_info[o.read1];
    const ReadInfo& i2 = read_info[o.read2];
    const bool rev1 = add_duplicates && !o.ori1, rev2 = add_duplicates && !o.ori2;  // :176-183
    if (!(rev1 ? i1.vertex_rev_set : i1.vertex_set) || !(rev2 ? i2.vertex_rev_set : i2.vertex_set))
        throw FatalError{HC_ERR_STATE, "Read::get_vertex_id: vertex id not set"};  // Read.h:111-120 asserts it
    const bool p1 = i1.paired, p2 = i2.paired;
    const int pos1 = (int)o.pos1, pos2 = (int)o.pos2;
    int pos3, pos4 = 0;
    if (!p1 && !p2) {
        pos3 = (int)i1.len_a - pos1 - (int)i2.len_a;                 // :222
    } else if (!p1 && p2) {
        pos3 = (int)i1.len_a - pos2 - (int)i2.len_b;                 // :262
        pos4 = (int)i1.len_a - pos1 - (int)i2.len_a;                 // :263
    } else if (p1 && !p2) {
        pos3 = (int)i1.len_b + pos2 - (int)i2.len_a;                 // :300
        pos4 = (int)i2.len_a + pos1 - (int)i1.len_a;                 // :301
    } else {
        pos3 = o.ord == '1' ? (int)i1.len_b - pos2 - (int)i2.len_b   // :363
                            : (int)i1.len_b + pos2 - (int)i2.len_b;  // :370
        pos4 = (int)i1.len_a - pos1 - (int)i2.len_a;                 // :372
    }
    Edge e(o.score, pos1, pos2, o.ori1 != 0, o.ori2 != 0, (char)o.ord, i1.read, i2.read);
    e.set_vertices(rev1 ? i1.vertex_rev : i1.vertex, rev2 ? i2.vertex_rev : i2.vertex);
    e.set_extra_pos(pos3, pos4);
    e.set_perc((int)o.perc);
    e.set_len((int)o.len1, (!p1 && !p2) ? 0 : (int)o.len2);  // :227 / :268
    e.set_mismatch((float)o.mm / (double)o.n);                // :132
    return e;
}

// src/EdgeCalculator.cpp:404-414 for the records of a block that the device did not drop: the class (host libm for the
// guard band), exp() of the admitted ones, the lines of the non-edges; sequence order is kept by concatenating the
// threads' pieces in order.
void EdgeCalculator::finalize_block(const ParsedBatch& batch, const hc_gather_row* rows, uint64_t n_rows, uint64_t base, BlockOut& out) {
    out.admitted.clear();
    out.nonedge_text.clear();
    out.nonedges = out.ambiguous = 0;
    if (n_rows == 0) return;
    struct Piece {
        std::vector<hc_admit_rec> admitted;
        std::string nonedge_text;
        uint64_t nonedges = 0, ambiguous = 0;
        FatalError error{0, ""};
    };
    auto build = [&](uint64_t kb, uint64_t ke, Piece& pc) {
        char linebuf[192];
        pc.admitted.reserve((size_t)(ke - kb));
        for (uint64_t k = kb; k < ke; k++) {
            const hc_gather_row& row = rows[k];
            const size_t i = (size_t)(row.index - base);
            if (row.index < base || i >= batch.size()) {
                pc.error = FatalError{HC_ERR_STATE, "scored record outside its block"};
                return;
            }
            static_assert(sizeof(hc_gather_row) == 32 && sizeof(hc_result_rec) == 24, "a row is an index followed by a result record");
            const hc_result_rec& r = *(const hc_result_rec*)&row.x1;
            uint32_t cls = HC_RES_CLS(r);
            const Overlap& line = batch.lines[i];
            if (cls == HC_CLS_ERROR) {
                pc.error = FatalError{HC_ERR_DATA, "overlap " + line.get_overlap_line() + " touches an invalid base or quality byte"};
                return;
            }
            if (cls == HC_CLS_NONEDGE) {  // :410-413
                pc.nonedge_text.append(linebuf, line.write_line(linebuf));
                pc.nonedges++;
                continue;
            }
            double score, mismatch_rate;
            if (cls == HC_CLS_AMBIG) pc.ambiguous++;
            const int st = hc_finalize(&m_cs, &r, &score, &mismatch_rate, &cls);  // exp() with the host libm
            if (st != HC_OK) {
                pc.error = FatalError{st, "hc_finalize"};
                return;
            }
            if (cls == HC_CLS_DROP) continue;
            if (cls == HC_CLS_NONEDGE) {
                pc.nonedge_text.append(linebuf, line.write_line(linebuf));
                pc.nonedges++;
                continue;
            }
            const hc_cand_rec& c = batch.recs[i];
            hc_admit_rec a;
            a.score = score;
            a.read1 = c.read1;
            a.read2 = c.read2;
            a.pos1 = line.m_pos1;
            a.pos2 = line.m_pos2;
            a.mm = r.mm;
            a.n = HC_RES_N(r);
            a.len1 = line.m_len1;
            a.len2 = line.m_len2;
            a.perc = line.get_perc();
            a.ori1 = line.m_ori1 == '+';
            a.ori2 = line.m_ori2 == '+';
            a.ord = (uint8_t)line.m_ord;
            a.pad = 0;
            pc.admitted.push_back(a);
        }
    };
    constexpr unsigned build_cap = 8u;  // more threads per block bought nothing (a round-2 knob, gone)
    unsigned T = program_settings.n_threads > 1 ? std::min<unsigned>(program_settings.n_threads, std::max(1u, build_cap)) : 1;
    if (n_rows < 4096) T = 1;
    std::vector<Piece> pieces(T);
    if (T == 1) {
        build(0, n_rows, pieces[0]);
    } else {
        if (!m_build_pool || m_build_pool->workers() + 1 < T) m_build_pool.reset(new WorkerPool(T - 1));
        m_build_pool->run(T, [&](unsigned int t) {
            try {
                build(n_rows * t / T, n_rows * (t + 1) / T, pieces[t]);
            } catch (const FatalError& e) {
                pieces[t].error = e;
            } catch (const std::exception& e) {  // nothing may leave a pool thread
                pieces[t].error = FatalError{HC_ERR_NOMEM, e.what()};
            }
        });
    }
    size_t n_adm = 0;
    for (const Piece& pc : pieces) {
        if (pc.error.status) throw pc.error;  // the first one in sequence order
        n_adm += pc.admitted.size();
    }
    out.admitted.reserve(n_adm);
    for (Piece& pc : pieces) {
        out.admitted.insert(out.admitted.end(), pc.admitted.begin(), pc.admitted.end());
        out.nonedge_text += pc.nonedge_text;
        out.nonedges += pc.nonedges;
        out.ambiguous += pc.ambiguous;
    }
}

// src/EdgeCalculator.cpp:431-555: the serial half for one block
void EdgeCalculator::consume_block(BlockOut& blk) {
    const double t1 = now_s();
    stats.nonedges_written += blk.nonedges;
    stats.ambiguous += blk.ambiguous;
    if (m_collect) {  // resolved once, after the last block; the device receives its copy now, behind the scoring of later blocks
        // (the records stay where they are until the graph is resolved: m_admitted owns the buffer from here on)
        const hc_admit_rec* recs = blk.admitted.data();
        const size_t n_recs = blk.admitted.size();
        m_admitted.emplace_back(std::move(blk.admitted));
        if (m_device_resolve && n_recs) {
            if (m_appender) {
                if (m_appender->failed.load(std::memory_order_acquire)) finish_appender(true);  // throws what the append failed with, here and now
                if (m_appender) m_appender->push(recs, n_recs);
            }
            else check(hc_graph_append(m_ctx, recs, n_recs), "hc_graph_append");
        }
        blk.admitted = std::vector<hc_admit_rec>();
    } else {
        // The second read of an edge is a random place in the graph's slot index and in the in-lists: ask for the
        // slot and the list header 2*kAhead edges early, and for the end of the list (its header is in cache by
        // then) kAhead edges early.
        const unsigned int dups_before = dup_count;
        const uint64_t added_before = stats.edges_added;
        std::vector<Edge> edges;
        edges.reserve(blk.admitted.size());
        for (const hc_admit_rec& a : blk.admitted) edges.push_back(edge_from_admit(a, m_read_info.data(), program_settings.add_duplicates));
        constexpr size_t kAhead = 8;
        const size_t m = edges.size();
        for (size_t k = 0; k < m; k++) {
            if (k + 2 * kAhead < m) {
                const Edge& a = edges[k + 2 * kAhead];
                overlap_graph->prefetch_slot(a.get_vertex(1), a.get_vertex(2), a.get_ori(1) == a.get_ori(2));
                if (a.get_pos(1) == 0) overlap_graph->prefetch_slot(a.get_vertex(2), a.get_vertex(1), false);  // may be swapped, :443-448
            }
            if (k + kAhead < m) {
                const Edge& a = edges[k + kAhead];
                overlap_graph->prefetch_in_list(a.get_vertex(2));
                if (a.get_pos(1) == 0) overlap_graph->prefetch_in_list(a.get_vertex(1));
            }
            InsertCounters ic;
            insert_edge(*overlap_graph, program_settings, edges[k], ic);
            inclusion_count += ic.inclusion_count;
            dup_count += ic.dup_count;
            stats.edges_added += ic.edges_added;
        }
        if (program_settings.verbose) {
            printf("Number of edges found: %lu\n", (unsigned long)(stats.edges_added - added_before));
            printf("Number of duplicates: %u\n", dup_count - dups_before);
        }
    }
    const double t2 = now_s();
    stats.t_insert += t2 - t1;
    // :546-555 (the file is opened in append mode even when nothing is written)
    FILE* fo = fopen((program_settings.output_dir + "nonedge_overlaps.txt").c_str(), "a");
    if (fo) {
        fwrite(blk.nonedge_text.data(), 1, blk.nonedge_text.size(), fo);
        fclose(fo);
    }
    stats.t_write += now_s() - t2;
}

void EdgeCalculator::start_appender() {
    finish_appender(false);
    m_appender.reset(new Appender(m_ctx, [this] { bind_here(); }));
}

void EdgeCalculator::finish_appender(bool rethrow) {
    if (!m_appender) return;
    m_appender->finish();
    const FatalError e = m_appender->error;
    m_appender.reset();
    if (rethrow && e.status) throw e;
}

// The serial half for the whole file on the device: hc_graph_resolve + hc_graph_fetch, then the
// lists of the graph simply point into the arrays that came back.
void EdgeCalculator::resolve_on_device(bool sorted) {
    const bool timing = getenv("HC_STAGE_TIMING") != nullptr;
    double tp = now_s();
    auto lap = [&](const char* what) {
        if (timing) {
            const double t = now_s();
            fprintf(stderr, "[hc stage] device resolve: %s %.3f s\n", what, t - tp);
            tp = t;
        }
    };
    const size_t V = overlap_graph->adj_out.size();
    const size_t R = m_read_info.size();
    std::vector<uint32_t> vtx;  // only when the vertex ids are not the read indices
    bool identity = true;
    for (size_t r = 0; r < R && identity; r++) identity = m_read_info[r].vertex_set && m_read_info[r].vertex == r;
    if (!identity) {
        vtx.resize(R);
        for (size_t r = 0; r < R; r++) vtx[r] = m_read_info[r].vertex_set ? (uint32_t)m_read_info[r].vertex : 0xFFFFFFFFu;  // unset: out of range
    }
    hc_graph_counts gc;
    size_t total = 0;
    for (const auto& b : m_admitted) total += b.size();
    // destination of the fetch: plain uninitialised memory (a zero-filled std::vector would touch 300 MB at C3 first), sized for the
    // upper bound (every admitted record an edge) BEFORE the device resolves, so that a few threads can fault its pages in while it
    // does: the copy into fresh pages ran at 15 GB/s, into touched ones it runs at the runtime's pageable-copy rate
    struct Raw {
        void* p = nullptr;
        size_t bytes = 0;
        explicit Raw(size_t want) {
            const size_t huge = (size_t)2 << 20;
            bytes = (std::max<size_t>(want, 1) + huge - 1) & ~(huge - 1);
            if (posix_memalign(&p, huge, bytes) != 0) throw FatalError{HC_ERR_NOMEM, "construct_edges: out of memory"};
            madvise(p, bytes, MADV_HUGEPAGE);
        }
        ~Raw() { free(p); }
        Raw(const Raw&) = delete;
        Raw& operator=(const Raw&) = delete;
    };
    // (freed by the clean-up thread, off the caller's clock: unmapping 300 MB takes as long as the fetch's copy)
    struct Fetched {
        Raw edges, in, seq;
        Fetched(size_t a, size_t b, size_t c) : edges(a), in(b), seq(c) {}
    };
    std::unique_ptr<Fetched> fetched(new Fetched(total * sizeof(hc_edge_rec), total * sizeof(uint32_t), total * sizeof(uint32_t)));
    struct HandOver {
        std::unique_ptr<Fetched>& f;
        EdgeCalculator* self;
        ~HandOver() { self->defer_cleanup([p = f.release()] { delete p; }); }
    } hand_over{fetched, this};
    std::vector<std::thread> touchers;
    struct JoinAll {
        std::vector<std::thread>& t;
        ~JoinAll() {
            for (auto& x : t)
                if (x.joinable()) x.join();
        }
    } join_all{touchers};
    {
        const unsigned T = total * sizeof(hc_edge_rec) < ((size_t)64 << 20) ? 0u : std::max(1u, std::min(8u, program_settings.n_threads));
        for (unsigned t = 0; t < T; t++)
            touchers.emplace_back([&, t, T] {
                bind_here();
                for (Raw* r : {&fetched->edges, &fetched->in}) {
                    volatile char* q = (volatile char*)r->p;
                    for (size_t at = r->bytes * t / T & ~(size_t)4095; at < r->bytes * (t + 1) / T; at += 4096) q[at] = 0;
                }
            });
    }
    check(hc_graph_resolve(m_ctx, nullptr, total, V, identity ? nullptr : vtx.data(), sorted ? HC_GRAPH_SORTED : HC_GRAPH_INSERTION_ORDER, &gc),
          "hc_graph_resolve");
    for (auto& x : touchers) x.join();
    touchers.clear();
    lap("resolve");
    if (gc.first_bad >= 0) {  // the record the reference's Edge rejects: say what it says
        size_t at = (size_t)gc.first_bad;
        for (const auto& b : m_admitted) {
            if (at < b.size()) {
                (void)edge_from_admit(b[at], m_read_info.data());
                break;
            }
            at -= b.size();
        }
        throw FatalError{HC_ERR_STATE, "hc_graph_resolve rejected an admitted record the host accepts"};
    }
    const size_t E = (size_t)gc.n_edges;
    if (E > total) throw FatalError{HC_ERR_STATE, "hc_graph_resolve returned more edges than admitted records"};
    hc_edge_rec* edges = (hc_edge_rec*)fetched->edges.p;
    uint32_t* in_nodes = (uint32_t*)fetched->in.p;
    uint32_t* seq = gc.n_tied_lists ? (uint32_t*)fetched->seq.p : nullptr;
    std::vector<uint64_t> out_off(V + 1), in_off(V + 1);
    std::vector<uint32_t> tied((size_t)gc.n_tied_lists);
    std::vector<uint8_t> incl(V);
    std::vector<Read*>& reads = fastq_storage->m_read_vec;
    // HC_FETCH_PIECE_BYTES: bytes of edges per piece (default 32 MiB; 0: the whole graph in one fetch, then adopt — round 3's order; tests
    // set a few KiB to run small graphs through the pieces)
    const size_t piece_bytes = getenv("HC_FETCH_PIECE_BYTES") ? (size_t)strtoull(getenv("HC_FETCH_PIECE_BYTES"), nullptr, 10) : ((size_t)32 << 20);
    if (piece_bytes == 0 || E * sizeof(hc_edge_rec) <= piece_bytes) {
        check(hc_graph_fetch(m_ctx, edges, out_off.data(), in_nodes, in_off.data(), seq, incl.data(), tied.empty() ? nullptr : tied.data()),
              "hc_graph_fetch");
        lap("fetch");
        overlap_graph->adopt_csr(edges, out_off.data(), in_nodes, in_off.data(), incl.data(), reads.data(), reads.size(), program_settings.n_threads);
        lap("adopt");
    } else {
        // The small arrays first (offsets, in-lists, inclusions: 20 MB at C3), then the 300 MB of edges in pieces of 32 MiB while the host
        // threads turn what has arrived into the graph's lists (round 3: fetch 7 ms, then adopt 5 - 7 ms, one after the other)
        check(hc_graph_fetch(m_ctx, nullptr, out_off.data(), in_nodes, in_off.data(), seq, incl.data(), tied.empty() ? nullptr : tied.data()),
              "hc_graph_fetch");
        lap("fetch of the offsets and in-lists");
        // The CALLING thread copies (it holds the HIP runtime's per-thread state: a fresh thread's first copy cost 20 - 50 ms on a process's
        // first file), a helper thread runs the adoption, whose workers wait for the copy's front.
        std::atomic<size_t> arrived{0};
        std::atomic<bool> abandon{false};
        std::exception_ptr adopt_error;
        std::thread adopter([&] {
            bind_here();
            try {
                overlap_graph->adopt_csr(edges, out_off.data(), in_nodes, in_off.data(), incl.data(), reads.data(), reads.size(), program_settings.n_threads,
                                         &arrived, &abandon);
            } catch (...) {
                adopt_error = std::current_exception();
            }
        });
        int fetch_rc = HC_OK;
        std::string fetch_err;
        {
            const size_t piece = std::max<size_t>(1, piece_bytes / sizeof(hc_edge_rec));
            for (size_t at = 0; at < E; at += piece) {
                const size_t k = std::min(piece, E - at);
                fetch_rc = hc_graph_fetch_edges(m_ctx, at, k, edges + at);
                if (fetch_rc != HC_OK) {
                    fetch_err = hc_last_error();
                    abandon.store(true, std::memory_order_release);
                    break;
                }
                arrived.store(at + k, std::memory_order_release);
            }
        }
        adopter.join();
        if (fetch_rc != HC_OK) throw FatalError{fetch_rc, "hc_graph_fetch_edges: " + fetch_err};
        if (adopt_error) std::rethrow_exception(adopt_error);
        lap("fetch of the edges in pieces + adopt behind it");
    }
    if (!tied.empty()) {
        // sortEdges order, lists longer than 16 with fully tied edges: std::sort's order of those is a function of
        // the insertion order — put exactly those lists back into insertion order and let sortEdges' own code sort them
        std::vector<uint32_t> len(reads.size());
        for (size_t r = 0; r < len.size(); r++) len[r] = reads[r]->get_len();
        for (uint32_t v : tied) {
            ArenaList<Edge>& L = overlap_graph->adj_out[v];
            const size_t a = (size_t)out_off[v];
            std::vector<std::pair<uint32_t, Edge>> by_seq;
            for (size_t k = 0; k < L.size(); k++) by_seq.emplace_back(seq[a + k], L[k]);
            std::sort(by_seq.begin(), by_seq.end(), [](const std::pair<uint32_t, Edge>& x, const std::pair<uint32_t, Edge>& y) { return x.first < y.first; });
            for (size_t k = 0; k < L.size(); k++) L[k] = by_seq[k].second;
            overlap_graph->sort_out_list(v, len.data());
        }
        overlap_graph->rebuild_in_lists(program_settings.n_threads);
        lap("tied lists");
    }
    inclusion_count += (unsigned int)gc.inclusion_count;
    dup_count += (unsigned int)gc.dup_count;
    stats.edges_added += gc.n_edges;
    if (program_settings.verbose) {  // totals instead of the reference's per-batch lines
        printf("Number of edges found: %lu\n", (unsigned long)gc.n_edges);
        printf("Number of duplicates: %lu\n", (unsigned long)gc.dup_count);
    }
}

// The same on host threads (HC_RESOLVE=host, or vertex ids beyond the device's 31 bits).
void EdgeCalculator::resolve_on_host() {
    std::vector<size_t> at(m_admitted.size() + 1, 0);
    for (size_t b = 0; b < m_admitted.size(); b++) at[b + 1] = at[b] + m_admitted[b].size();
    const size_t total = at.back();
    InsertCounters ic;
    static_assert(std::is_trivially_copyable<Edge>::value, "Edge lives in a malloc'd array");
    Edge* all = nullptr;
    if (total) {
        const size_t bytes = (total * sizeof(Edge) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign((void**)&all, (size_t)2 << 20, bytes) != 0) throw FatalError{HC_ERR_NOMEM, "construct_edges: out of memory"};
        madvise(all, bytes, MADV_HUGEPAGE);
    }
    try {
        const unsigned T = total < (1u << 16) ? 1u : std::max(1u, std::min<unsigned>(program_settings.n_threads, 16u));
        std::vector<FatalError> errs(T, FatalError{0, ""});
        std::vector<std::thread> th;
        auto build = [&](unsigned t) {  // blocks dealt to the threads in contiguous runs: errors come out in sequence order
            const size_t nb = m_admitted.size();
            for (size_t b = nb * t / T; b < nb * (t + 1) / T; b++)
                for (size_t k = 0; k < m_admitted[b].size(); k++) {
                    try {
                        new ((void*)(all + at[b] + k)) Edge(edge_from_admit(m_admitted[b][k], m_read_info.data(), program_settings.add_duplicates));
                    } catch (const FatalError& e) {
                        errs[t] = e;
                        return;
                    }
                }
        };
        for (unsigned t = 1; t < T; t++) th.emplace_back(build, t);
        build(0);
        for (auto& x : th) x.join();
        for (const FatalError& e : errs)
            if (e.status) throw e;  // the first one in sequence order
        resolve_admitted_edges(*overlap_graph, program_settings, all, total, ic);
    } catch (...) {
        free(all);
        throw;
    }
    free(all);
    inclusion_count += ic.inclusion_count;
    dup_count += ic.dup_count;
    stats.edges_added += ic.edges_added;
    if (program_settings.verbose) {
        printf("Number of edges found: %lu\n", (unsigned long)ic.edges_added);
        printf("Number of duplicates: %u\n", ic.dup_count);
    }
}

static Overlap overlap_of(const hc_line_rec& l) {
    Overlap o;
    o.m_id1 = l.id1;
    o.m_id2 = l.id2;
    o.m_pos1 = l.pos1;
    o.m_pos2 = l.pos2;
    o.m_perc1 = l.perc1;
    o.m_perc2 = l.perc2;
    o.m_len1 = l.len1;
    o.m_len2 = l.len2;
    o.m_ord = (char)l.ord;
    o.m_ori1 = (char)l.ori1;
    o.m_ori2 = (char)l.ori2;
    o.m_type1 = (char)l.type1;
    o.m_type2 = (char)l.type2;
    return o;
}

// finalize_block for a block the device parsed: every row carries the parsed line it came from
void EdgeCalculator::finalize_text_block(const IdIndex& ids, const hc_text_row* rows, uint64_t n_rows, BlockOut& out, unsigned threads) {
    out.admitted.clear();
    out.nonedge_text.clear();
    out.nonedges = out.ambiguous = 0;
    if (n_rows == 0) return;
    struct Piece {
        std::vector<hc_admit_rec> admitted;
        std::string nonedge_text;
        uint64_t nonedges = 0, ambiguous = 0;
        FatalError error{0, ""};
    };
    auto build = [&](uint64_t kb, uint64_t ke, Piece& pc) {
        char linebuf[192];
        pc.admitted.reserve((size_t)(ke - kb));
        for (uint64_t k = kb; k < ke; k++) {
            const hc_result_rec& r = *(const hc_result_rec*)&rows[k].row.x1;
            const hc_line_rec& l = rows[k].line;
            uint32_t cls = HC_RES_CLS(r);
            if (cls == HC_CLS_ERROR) {
                pc.error = FatalError{HC_ERR_DATA, "overlap " + overlap_of(l).get_overlap_line() + " touches an invalid base or quality byte"};
                return;
            }
            if (cls == HC_CLS_NONEDGE) {  // :410-413
                pc.nonedge_text.append(linebuf, overlap_of(l).write_line(linebuf));
                pc.nonedges++;
                continue;
            }
            double score, mismatch_rate;
            if (cls == HC_CLS_AMBIG) pc.ambiguous++;
            const int st = hc_finalize(&m_cs, &r, &score, &mismatch_rate, &cls);  // exp() with the host libm
            if (st != HC_OK) {
                pc.error = FatalError{st, "hc_finalize"};
                return;
            }
            if (cls == HC_CLS_DROP) continue;
            if (cls == HC_CLS_NONEDGE) {
                pc.nonedge_text.append(linebuf, overlap_of(l).write_line(linebuf));
                pc.nonedges++;
                continue;
            }
            hc_admit_rec a;
            if (!ids.find(l.id1, a.read1) || !ids.find(l.id2, a.read2)) {  // the device found them
                pc.error = FatalError{HC_ERR_STATE, "device-parsed line names an unknown read"};
                return;
            }
            a.score = score;
            a.pos1 = l.pos1;
            a.pos2 = l.pos2;
            a.mm = r.mm;
            a.n = HC_RES_N(r);
            a.len1 = l.len1;
            a.len2 = l.len2;
            a.perc = l.perc2 > 0 ? (uint32_t)(0.5 * (l.perc1 + l.perc2)) : l.perc1;  // Overlap::get_perc, src/Overlap.h:203-210
            a.ori1 = l.ori1 == '+';
            a.ori2 = l.ori2 == '+';
            a.ord = l.ord;
            a.pad = 0;
            pc.admitted.push_back(a);
        }
    };
    constexpr unsigned build_cap = 8u;  // more threads per block bought nothing (a round-2 knob, gone)
    unsigned T = program_settings.n_threads > 1 ? std::min<unsigned>(program_settings.n_threads, std::max(1u, build_cap)) : 1;
    if (n_rows < 4096) T = 1;
    // several collectors call this side by side (threads = 1: the pool is one): a block nearly all of whose lines survive — overlaps
    // straight from the finder — then spends 12 ms in one thread's exp(); such a block gets a few threads of its own
    bool own_threads = false;
    if (threads) {
        const unsigned want = n_rows >= 100000 ? std::min(4u, std::max(1u, program_settings.n_threads / 4)) : 1u;
        own_threads = threads == 1 && want > 1;
        T = own_threads ? want : std::min(T, threads);
        // (a caller that names more threads than the cap gets them: the device-lines route, where nearly every line survives and ONE helper finalises)
        if (threads > build_cap && n_rows >= 4096) T = std::min<unsigned>(threads, std::max(1u, program_settings.n_threads));
    }
    std::vector<Piece> pieces(T);
    if (T == 1) {
        build(0, n_rows, pieces[0]);
    } else if (own_threads) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++)
            th.emplace_back([&, t] {
                try {
                    build(n_rows * t / T, n_rows * (t + 1) / T, pieces[t]);
                } catch (const FatalError& e) {
                    pieces[t].error = e;
                } catch (const std::exception& e) {
                    pieces[t].error = FatalError{HC_ERR_NOMEM, e.what()};
                }
            });
        try {
            build(0, n_rows / T, pieces[0]);
        } catch (const FatalError& e) {
            pieces[0].error = e;
        } catch (const std::exception& e) {
            pieces[0].error = FatalError{HC_ERR_NOMEM, e.what()};
        }
        for (auto& x : th) x.join();
    } else {
        if (!m_build_pool || m_build_pool->workers() + 1 < T) m_build_pool.reset(new WorkerPool(T - 1));
        m_build_pool->run(T, [&](unsigned int t) {
            try {
                build(n_rows * t / T, n_rows * (t + 1) / T, pieces[t]);
            } catch (const FatalError& e) {
                pieces[t].error = e;
            } catch (const std::exception& e) {  // nothing may leave a pool thread
                pieces[t].error = FatalError{HC_ERR_NOMEM, e.what()};
            }
        });
    }
    size_t n_adm = 0;
    for (const Piece& pc : pieces) {
        if (pc.error.status) throw pc.error;  // the first one in sequence order
        n_adm += pc.admitted.size();
    }
    out.admitted.reserve(n_adm);
    for (Piece& pc : pieces) {
        out.admitted.insert(out.admitted.end(), pc.admitted.begin(), pc.admitted.end());
        out.nonedge_text += pc.nonedge_text;
        out.nonedges += pc.nonedges;
        out.ambiguous += pc.ambiguous;
    }
}

static hc_line_rec line_rec_of(const Overlap& o) {
    hc_line_rec l;
    memset(&l, 0, sizeof l);
    l.id1 = o.m_id1;
    l.id2 = o.m_id2;
    l.pos1 = o.m_pos1;
    l.pos2 = o.m_pos2;
    l.perc1 = o.m_perc1;
    l.perc2 = o.m_perc2;
    l.len1 = o.m_len1;
    l.len2 = o.m_len2;
    l.ord = (uint8_t)o.m_ord;
    l.ori1 = (uint8_t)o.m_ori1;
    l.ori2 = (uint8_t)o.m_ori2;
    l.type1 = (uint8_t)o.m_type1;
    l.type2 = (uint8_t)o.m_type2;
    return l;
}

// Per-LINE fallback (round 5; the reference reads every line by itself, src/EdgeCalculator.cpp:581-604).  The device's parser reads the
// plain lines of a block and LISTS the others (padding the reference trims at :584, an id with a leading zero — strtoul(.., 0) reads it
// as octal —, --allow_spaced_overlaps input, a line with other than 13 fields, ...): each of those goes through the host's tokeniser +
// Overlap constructor alone (OverlapsParser::classify_line: every message and every exit is the host parser's), the ones that pass are
// scored on the device as one small block, and their rows are spliced into the block's rows at their places in file order.  Until round 5 one
// such line sent its whole 16 MiB block to the host's tokeniser, synchronously, at its place in the order.
void EdgeCalculator::score_odd_lines(const OverlapsParser& parser, const char* block_text, const hc_text_result& tr, OddLines& odd) {
    std::vector<hc_cand_rec> recs;
    std::vector<std::pair<uint32_t, Overlap>> passing;  // (line number in the block, the line)
    for (uint64_t j = 0; j < tr.n_nonplain_listed; j++) {
        const hc_text_nonplain& np = tr.nonplain[j];
        Overlap o;
        hc_cand_rec rec;
        switch (parser.classify_line(block_text + np.begin, np.length, o, rec)) {  // throws what the reference exits on
            case OverlapsParser::LineKind::Malformed: odd.pc.malformed++; break;
            case OverlapsParser::LineKind::Self: odd.pc.self_overlaps++; break;
            case OverlapsParser::LineKind::Silent: odd.pc.silently_dropped++; break;
            case OverlapsParser::LineKind::Rejected:
                odd.pc.prefilter_rejected++;
                odd.rejected.emplace_back(np.line_index, o);
                break;
            case OverlapsParser::LineKind::Pass:
                recs.push_back(rec);
                passing.emplace_back(np.line_index, o);
                break;
        }
    }
    odd.scored = recs.size();
    std::vector<hc_text_row> mine;  // the rows of the passing odd lines, in line order
    if (!recs.empty()) {
        std::lock_guard<std::mutex> g(m_odd_mu);  // one small block for all collectors: these lines are rare
        if (!m_odd_blk || recs.size() > m_odd_blk_cap) {
            hc_block_destroy(m_odd_blk);
            m_odd_blk = nullptr;
            m_odd_blk_cap = std::max<size_t>(recs.size() + recs.size() / 4, 4096);
            check(hc_block_create(m_dev[0].ctx, m_odd_blk_cap, &m_odd_blk), "hc_block_create");
        }
        const hc_gather_row* rows = nullptr;
        uint64_t n_rows = 0;
        check(hc_block_submit(m_odd_blk, recs.data(), recs.size(), 0), "hc_block_submit");
        check(hc_block_wait(m_odd_blk, &rows, &n_rows), "hc_block_wait");
        mine.resize(n_rows);
        for (uint64_t r = 0; r < n_rows; r++) {  // index = position among the passing odd lines -> the line's number in the block
            mine[r].row = rows[r];
            const auto& src = passing[rows[r].index];
            mine[r].row.index = src.first;
            mine[r].line = line_rec_of(src.second);
        }
    }
    // splice: both lists are sorted by line number
    odd.rows.resize(tr.n_rows + mine.size());
    std::merge(tr.rows, tr.rows + tr.n_rows, mine.begin(), mine.end(), odd.rows.begin(),
               [](const hc_text_row& a, const hc_text_row& b) { return a.row.index < b.row.index; });
}

// The file's TEXT sent to the device block by block: the caller's thread copies the next stretch of
// the file — cut behind a newline — into the page-locked buffer of a text block and submits it (split into lines,
// parse, --max_ov, prefilter, id lookup and scoring all happen on the device, hc_textblock_*); the collector thread
// waits for the blocks in file order and runs the serial half on what survived.  Lines that are not plain are read by the host
// ONE BY ONE (score_odd_lines) while the rest of their block stays on the device; a block the device reports as unusual beyond that
// (more such lines than its list holds, an id that is not in the FASTQ input, more lines than it has room for) is tokenised by the host
// parser and scored as a block of records, synchronously, at its place in the order: every error and every malformed-line message
// comes out as from the host-parsed pipeline.
void EdgeCalculator::score_device_parsed(OverlapsParser& parser, std::vector<Overlap>& rejected, ParseCounters& pc) {
    const size_t N = m_dev.size();
    const size_t D = m_text_depth;  // text blocks per device
    const size_t R = D * N + 1;
    const size_t B = m_text_block;
    const double t_setup0 = now_s();
    // the calling thread copies text too: next to the device for the length of the call, then back where it was allowed before
    BoundForNow bound(m_node_cpus);
    for (Device& d : m_dev) {
        if (d.tblk.size() < D) d.tblk.resize(D, nullptr);  // (never shrunk: a resident process's devices may bring more blocks than this file needs)
        for (hc_textblock*& b : d.tblk)
            if (!b) {
                check(hc_textblock_create(d.ctx, B, &b), "hc_textblock_create");
                if (m_odd_line_cap) check(hc_textblock_list_nonplain(b, m_odd_line_cap), "hc_textblock_list_nonplain");
            }
    }
    if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] text blocks ready after %.3f s\n", now_s() - t_setup0);
    struct Slot {
        hc_textblock* tb = nullptr;  // nullptr: the host's block (nothing was submitted)
        size_t begin = 0, end = 0;
        uint64_t first_line = 0, n_lines = 0;
    };
    std::vector<Slot> ring(R);
    std::vector<BlockOut> outs(R);
    std::mutex mu;
    std::condition_variable cv;
    size_t submitted = 0, consumed = 0;
    bool producer_done = false;
    std::atomic<bool> collector_failed{false};
    FatalError collector_error{0, ""};
    double t_collect = 0;
    uint64_t n_host_blocks = 0;
    double tm_wait = 0, tm_final = 0, tm_turn = 0, tm_consume = 0, tm_slot = 0, tm_copy = 0, tm_submit = 0;  // HC_STAGE_TIMING
    // Where the text comes from and who numbers the lines (HC_TEXT_SOURCE):
    //   pread-chain (default): pread on the pool into the block's page-locked buffer, which the host then never looks at —
    //       the cut is found in the file's mapping, and the blocks number their lines through a chain of counters on the
    //       device side (hc_linechain).  Counting the newlines of what it had just copied took the host as long as the copy:
    //       C3 copy 0.16 -> 0.075 s, stage 0.26 -> 0.20 s.
    //   pread: the same copy with the host's newline count and explicit line numbers (round-2 form, kept as a route).
    //   map: no copy at all — the runtime moves the pageable mapping (hc_textblock_submit_from on the mapping): 0.25 s,
    //       the runtime's pageable copy reaches 25 GB/s beside the collectors' HIP calls (40-50 alone:
    //       tools/experiments/register_cost.cpp).
    // HC_TEXT_BUFFER=wc makes the blocks' buffers write-combined (pread fills those 1.4x as fast in isolation,
    // tools/experiments/pread_targets.cpp; no gain in the stage, where the producer then waits for the device).
    const char* text_source = getenv("HC_TEXT_SOURCE");
    const bool from_map = text_source && !strcmp(text_source, "map");
    const bool pread_chain = !text_source || !strcmp(text_source, "pread-chain");
    const bool chained = from_map || pread_chain;
    struct ChainGuard {
        hc_linechain* p = nullptr;
        ~ChainGuard() { hc_linechain_destroy(p); }
    } chain;
    if (chained) check(hc_linechain_create(m_ctx, parser.size() / std::max<size_t>(B / 2, 1) + 4, &chain.p), "hc_linechain_create");
    std::atomic<uint64_t> lines_consumed{0};  // from_map: what the collectors have seen (the device applies --max_ov exactly)
    // Several collectors: collector c takes the blocks k = c, c + C, ...; waiting for the device, putting the rows in
    // order and finalising them (exp() of the admitted ones, the lines of the non-edges) happens side by side for
    // different blocks, the serial half (and everything that touches shared state) strictly in block order.
    unsigned C = std::max(1u, std::min<unsigned>(4u, program_settings.n_threads));
    if (const char* e = getenv("HC_COLLECTORS")) C = std::max(1, atoi(e));
    C = (unsigned)std::min<size_t>(C, D * N);  // at most D * N blocks are in flight
    auto collect = [&](unsigned c) {
        ParsedBatch::RecStorage pinned;
        pinned.ctx = m_ctx;
        pinned.alloc = [](void* ctx, size_t n) -> hc_cand_rec* {
            void* p = nullptr;
            check(hc_host_alloc((hc_ctx*)ctx, &p, n * sizeof(hc_cand_rec)), "hc_host_alloc");
            return (hc_cand_rec*)p;
        };
        pinned.release = [](void* ctx, hc_cand_rec* p) { hc_host_free((hc_ctx*)ctx, p); };
        ParsedBatch host_batch(pinned);
        for (size_t k = c;; k += C) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return submitted > k || producer_done; });
                if (submitted <= k) return;
            }
            const Slot sl = ring[k % R];
            BlockOut& out = outs[k % R];
            hc_text_result tr;
            memset(&tr, 0, sizeof tr);
            tr.needs_host = 1;
            FatalError mine{0, ""};
            OddLines odd;  // the block's lines the device did not read, handled one by one (per-line fallback)
            const double t0 = now_s();
            double t_w = t0;
            try {  // side by side with the other collectors
                if (sl.tb) check(hc_textblock_wait(sl.tb, &tr), "hc_textblock_wait");
                t_w = now_s();
                if (!tr.needs_host && !collector_failed) {
                    if (tr.n_nonplain_listed) {
                        score_odd_lines(parser, parser.data() + sl.begin, tr, odd);
                        finalize_text_block(parser.ids(), odd.rows.data(), odd.rows.size(), out, /*threads=*/1);
                    } else {
                        finalize_text_block(parser.ids(), tr.rows, tr.n_rows, out, /*threads=*/1);
                    }
                }
            } catch (const FatalError& e) {
                mine = e;
            } catch (const std::exception& e) {
                mine = FatalError{HC_ERR_NOMEM, e.what()};
            }
            const double t1 = now_s();
            {  // in block order from here
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return consumed == k; });
            }
            const double t2 = now_s();
            if (!collector_failed) {
                try {
                    if (mine.status) throw mine;
                    if (tr.needs_host) {  // the host's tokeniser + Overlap constructor own this block
                        n_host_blocks++;
                        stats.host_blocks++;
                        parser.parse_range(sl.begin, sl.end, chained ? lines_consumed.load() : sl.first_line, host_batch, rejected, pc,
                                           /*print_malformed=*/true);
                        const size_t n = host_batch.size();
                        const hc_gather_row* rows = nullptr;
                        uint64_t n_rows = 0;
                        if (n) {
                            Device& dev = m_dev[0];
                            if (n > m_block_cap || !dev.blk[0]) {
                                hc_block_destroy(dev.blk[0]);
                                dev.blk[0] = nullptr;
                                m_block_cap = std::max(m_block_cap, n + n / 4);
                                check(hc_block_create(dev.ctx, m_block_cap, &dev.blk[0]), "hc_block_create");
                            }
                            check(hc_block_submit(dev.blk[0], host_batch.recs, n, 0), "hc_block_submit");
                            check(hc_block_wait(dev.blk[0], &rows, &n_rows), "hc_block_wait");
                            stats.scored += n;
                        }
                        finalize_block(host_batch, rows, n_rows, 0, out);
                    } else {
                        stats.device_blocks++;
                        pc.lines_read += tr.lines_read;  // (the device counts the lines it leaves to the host too)
                        pc.self_overlaps += tr.self_overlaps + odd.pc.self_overlaps;
                        pc.silently_dropped += tr.silently_dropped + odd.pc.silently_dropped;
                        pc.prefilter_rejected += tr.prefilter_rejected + odd.pc.prefilter_rejected;
                        pc.malformed += odd.pc.malformed;
                        stats.scored += tr.scored + odd.scored;
                        stats.host_lines += tr.n_nonplain_listed;
                        for (uint64_t m = 0; m < odd.pc.malformed; m++) puts("incorrect overlap; skipping");  // :600, in block order
                        // the prefilter's rejects in file order: the device's (sorted by line) and the odd lines' (sorted by line), merged
                        size_t jo = 0;
                        for (uint64_t j = 0; j < tr.n_rejected; j++) {
                            while (jo < odd.rejected.size() && odd.rejected[jo].first < tr.rejected[j].line_index) rejected.push_back(odd.rejected[jo++].second);
                            rejected.push_back(overlap_of(tr.rejected[j].line));
                        }
                        while (jo < odd.rejected.size()) rejected.push_back(odd.rejected[jo++].second);
                    }
                    consume_block(out);
                    if (sl.tb) lines_consumed += tr.n_lines;  // (a block that never went to the device is the file's last)
                } catch (const FatalError& e) {  // the submitter may be reporting a failure of its own: under mu, the first error wins
                    std::lock_guard<std::mutex> g(mu);
                    if (!collector_failed) {
                        collector_error = e;
                        collector_failed = true;
                    }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!collector_failed) {
                        collector_error = FatalError{HC_ERR_NOMEM, e.what()};
                        collector_failed = true;
                    }
                }
            }
            {
                std::lock_guard<std::mutex> g(mu);
                const double t3 = now_s();
                t_collect += (t1 - t0) / C + (t3 - t2);
                tm_wait += t_w - t0, tm_final += t1 - t_w, tm_turn += t2 - t1, tm_consume += t3 - t2;
                consumed = k + 1;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> collectors;
    for (unsigned c = 0; c < C; c++)
        collectors.emplace_back([&, c] {
            bind_here();
            collect(c);
        });
    auto stop_collector = [&] {
        {
            std::lock_guard<std::mutex> g(mu);
            producer_done = true;
        }
        cv.notify_all();
        for (auto& t : collectors)
            if (t.joinable()) t.join();
    };
    // The HIP calls of a block (a dozen launches, copies and events: 0.1 ms) are issued by a thread of their own, so that
    // the caller's thread can go on copying the next block meanwhile; blocks are submitted and published strictly in order.
    struct Pending {
        size_t k = 0;
        hc_textblock* tb = nullptr;  // nullptr: nothing to submit (the host's block), only publish
        const char* src = nullptr;
        size_t bytes = 0;
        uint64_t line_no = 0;
        size_t chain_k = 0;
        hc_textblock* prev = nullptr;
    };
    std::deque<Pending> pending;
    std::mutex smu;
    std::condition_variable scv;
    bool no_more = false;
    std::thread submitter([&] {
        bind_here();
        for (;;) {
            Pending p;
            {
                std::unique_lock<std::mutex> g(smu);
                scv.wait(g, [&] { return !pending.empty() || no_more; });
                if (pending.empty()) return;
                p = pending.front();
                pending.pop_front();
            }
            if (p.tb && !collector_failed) {
                const double t0 = now_s();
                try {
                    if (chained) check(hc_textblock_submit_from(p.tb, p.src, p.bytes, chain.p, p.chain_k, p.prev, 0), "hc_textblock_submit_from");
                    else check(hc_textblock_submit(p.tb, p.bytes, p.line_no, 0), "hc_textblock_submit");
                } catch (const FatalError& e) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!collector_failed) {
                        collector_error = e;
                        collector_failed = true;
                    }
                }
                const double dt = now_s() - t0;
                tm_submit += dt;
            }
            {
                std::lock_guard<std::mutex> g(mu);
                submitted = p.k + 1;
            }
            cv.notify_all();
        }
    });
    auto hand_over = [&](const Pending& p) {
        {
            std::lock_guard<std::mutex> g(smu);
            pending.push_back(p);
        }
        scv.notify_one();
    };
    auto stop_submitter = [&] {
        {
            std::lock_guard<std::mutex> g(smu);
            no_more = true;
        }
        scv.notify_one();
        if (submitter.joinable()) submitter.join();
    };
    try {
        size_t pos = 0;
        uint64_t line_no = 0;
        const size_t size = parser.size();
        hc_textblock* prev_tb = nullptr;
        size_t chain_k = 0;
        for (size_t k = 0; pos < size && line_no < program_settings.max_overlaps && lines_consumed.load() < program_settings.max_overlaps; k++) {  // `&& i < max_overlaps`, :581
            const double ts = now_s();
            {  // the block object's previous user (block k - D * N) has been consumed
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return consumed + D * N > k; });
                if (collector_failed) break;
            }
            tm_slot += now_s() - ts;
            Slot sl;
            sl.begin = pos;
            sl.first_line = line_no;
            Device& dev = m_dev[k % N];
            hc_textblock* tb = dev.tblk[(k / N) % D];
            const double t0 = now_s();
            size_t end = std::min(size, pos + B);
            uint64_t newlines = 0;
            if (!from_map) {
                char* dst = hc_textblock_buffer(tb);
                if (!dst) throw FatalError{HC_ERR_NOMEM, "construct_edges: no page-locked buffer for a block of text"};
                parser.copy_range(dst, pos, end, newlines, /*count_newlines=*/!pread_chain);
            }
            if (end < size) {  // cut behind the last newline of the stretch
                const char* buf = chained ? parser.data() + pos : hc_textblock_buffer(tb);
                size_t cut = end - pos;
                while (cut > 0 && buf[cut - 1] != '\n') cut--;
                if (cut == 0) {  // one line longer than a block: the host's (its parser has no such limit)
                    sl.end = parser.line_end_at(end);
                    sl.tb = nullptr;
                    // its line count is only known once parsed: everything up to its end goes to the host in one piece,
                    // and with it the rest of the file, so that line numbers stay exact
                    sl.end = size;
                    stats.t_parse += now_s() - t0;
                    ring[k % R] = sl;
                    pos = size;
                    Pending p;
                    p.k = k;
                    hand_over(p);
                    break;
                }
                end = pos + cut;  // the bytes behind the cut are copied again with the next block
            }
            sl.end = end;
            if (!chained) {
                sl.n_lines = newlines;  // of the whole stretch; corrected below when it was cut
                if (end < std::min(size, pos + B)) {
                    const char* buf = hc_textblock_buffer(tb);
                    sl.n_lines -= (uint64_t)std::count(buf + (end - pos), buf + (std::min(size, pos + B) - pos), '\n');
                }
                if (end == size) {  // a last piece without a newline is a line too (std::getline)
                    const char* buf = hc_textblock_buffer(tb);
                    if (end > pos && buf[end - pos - 1] != '\n') sl.n_lines++;
                }
            }
            const double tc = now_s();
            stats.t_parse += tc - t0;
            tm_copy += tc - t0;
            sl.tb = tb;
            ring[k % R] = sl;  // (read by the collectors only once the submitter has published block k)
            Pending p;
            p.k = k;
            p.tb = tb;
            p.src = from_map ? parser.data() + pos : hc_textblock_buffer(tb);
            p.bytes = end - pos;
            p.line_no = line_no;
            p.chain_k = chain_k;
            p.prev = prev_tb;
            hand_over(p);
            if (chained) {
                prev_tb = tb;
                chain_k++;
            }
            pos = end;
            line_no += sl.n_lines;
        }
    } catch (...) {
        stop_submitter();
        stop_collector();
        throw;
    }
    stop_submitter();
    stop_collector();
    if (collector_failed) throw collector_error;
    stats.t_score = t_collect;
    for (Device& dev : m_dev)
        for (hc_textblock* tb : dev.tblk) stats.regrown_blocks += hc_textblock_regrown(tb);
    if (getenv("HC_STAGE_TIMING"))
        fprintf(stderr, "[hc stage] device-parsed pipeline: %lu block(s) went to the host parser; producer: waiting for a block object %.3f s, "
                        "copy %.3f s, submit %.3f s; %u collector(s), summed: waiting for the device + ordering rows %.3f s, finalise %.3f s, "
                        "waiting for their turn %.3f s, serial half %.3f s\n",
                (unsigned long)n_host_blocks, tm_slot, tm_copy, tm_submit, C, tm_wait, tm_final, tm_turn, tm_consume);
}

// The overlaps file's lines as parsed records in the primary device's memory (construct_edges_from_store): block k = lines
// [k * L, (k + 1) * L) with L = what a text block has room for; D blocks in flight on the block objects of the primary device; what comes
// back is consumed strictly in order, as score_device_parsed's collectors do.  No text, no host tokeniser: every line is plain by construction.
void EdgeCalculator::score_device_lines(OverlapsParser& parser, std::vector<Overlap>& rejected, ParseCounters& pc) {
    Device& dev = m_dev[0];
    const size_t D = m_text_depth;
    if (dev.tblk.size() < D) dev.tblk.resize(D, nullptr);
    for (hc_textblock*& b : dev.tblk)
        if (!b) {
            check(hc_textblock_create(dev.ctx, m_text_block, &b), "hc_textblock_create");
            if (m_odd_line_cap) check(hc_textblock_list_nonplain(b, m_odd_line_cap), "hc_textblock_list_nonplain");
        }
    uint64_t L = hc_textblock_max_lines(dev.tblk[0]);
    if (L == 0) throw FatalError{HC_ERR_STATE, "construct_edges: a text block without room for lines"};
    L = std::min<uint64_t>(L, 1u << 18);  // (smaller pieces than a block's capacity: the host's half of piece k runs beside the device's of k + 1)
    const uint64_t n_use = std::min<uint64_t>(m_lines_override_n, program_settings.max_overlaps);  // `&& i < max_overlaps`, :581
    const uint64_t K = (n_use + L - 1) / L;
    auto submit = [&](uint64_t k) {
        const uint64_t lo = k * L, n = std::min(L, n_use - lo);
        check(hc_textblock_submit_lines(dev.tblk[k % D], m_lines_override + lo, n, lo, 0), "hc_textblock_submit_lines");
    };
    const double t0 = now_s();
    // A helper thread waits for the blocks in order, finalises what survived of block k (exp() of the admitted rows, the non-edges' lines:
    // finalize_text_block on the build pool) and puts block k + D in flight; this thread runs the serial half in order (consume_block).
    struct Done {
        BlockOut out;
        hc_text_result tr;
        std::vector<Overlap> rejected;
    };
    const size_t R = 3;
    std::vector<Done> ring(R);
    std::mutex mu;
    std::condition_variable cv;
    uint64_t produced = 0, consumed = 0;
    FatalError failure{0, ""};
    bool failed = false;
    const unsigned threads = std::max(1u, std::min<unsigned>(24u, program_settings.n_threads));
    double tm_wait = 0, tm_final = 0, tm_consume = 0;  // HC_STAGE_TIMING
    std::thread helper([&] {
        bind_here();
        try {
            for (uint64_t k = 0; k < K && k < D; k++) submit(k);
            for (uint64_t k = 0; k < K; k++) {
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return consumed + R > k || failed; });
                    if (failed) return;
                }
                Done& dn = ring[k % R];
                const double tw0 = now_s();
                check(hc_textblock_wait(dev.tblk[k % D], &dn.tr), "hc_textblock_wait");
                const double tw1 = now_s();
                tm_wait += tw1 - tw0;
                if (dn.tr.needs_host)  // an id that is not in the FASTQ input: not from this stage's own reads
                    throw FatalError{HC_ERR_STATE, "construct_edges_from_store: a block of lines the device does not take (unknown read id?)"};
                finalize_text_block(parser.ids(), dn.tr.rows, dn.tr.n_rows, dn.out, threads);
                dn.rejected.clear();
                for (uint64_t j = 0; j < dn.tr.n_rejected; j++) dn.rejected.push_back(overlap_of(dn.tr.rejected[j].line));
                tm_final += now_s() - tw1;
                if (k + D < K) submit(k + D);  // (the block object is free again: its rows and rejects have been copied)
                {
                    std::lock_guard<std::mutex> g(mu);
                    produced = k + 1;
                }
                cv.notify_all();
            }
        } catch (const FatalError& e) {
            std::lock_guard<std::mutex> g(mu);
            failure = e;
            failed = true;
            cv.notify_all();
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> g(mu);
            failure = FatalError{HC_ERR_NOMEM, e.what()};
            failed = true;
            cv.notify_all();
        }
    });
    struct JoinHelper {
        std::thread& t;
        std::mutex& mu;
        std::condition_variable& cv;
        bool& failed;
        ~JoinHelper() {
            {
                std::lock_guard<std::mutex> g(mu);
                failed = true;  // (an exception on this side: the helper stops waiting)
            }
            cv.notify_all();
            if (t.joinable()) t.join();
        }
    };
    {
        JoinHelper join{helper, mu, cv, failed};
        for (uint64_t k = 0; k < K; k++) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return produced > k || (failed && failure.status); });
                if (produced <= k) throw failure;
            }
            Done& dn = ring[k % R];
            stats.device_blocks++;
            pc.lines_read += dn.tr.lines_read;
            pc.self_overlaps += dn.tr.self_overlaps;
            pc.silently_dropped += dn.tr.silently_dropped;
            pc.prefilter_rejected += dn.tr.prefilter_rejected;
            stats.scored += dn.tr.scored;
            for (Overlap& o : dn.rejected) rejected.push_back(o);
            const double tc0 = now_s();
            consume_block(dn.out);
            tm_consume += now_s() - tc0;
            {
                std::lock_guard<std::mutex> g(mu);
                consumed = k + 1;
            }
            cv.notify_all();
        }
        helper.join();
        std::lock_guard<std::mutex> g(mu);
        if (failure.status) throw failure;
    }
    stats.t_score = now_s() - t0;
    for (hc_textblock* tb : dev.tblk) stats.regrown_blocks += hc_textblock_regrown(tb);
    if (getenv("HC_STAGE_TIMING"))
        fprintf(stderr, "[hc stage] device-lines pipeline: %lu piece(s) of up to %lu lines; helper: waiting for the device %.3f s, finalise %.3f s; serial half %.3f s; "
                        "all %.3f s\n", (unsigned long)K, (unsigned long)L, tm_wait, tm_final, tm_consume, now_s() - t0);
}

// The file tokenised on the host's threads (HC_PARSE=host; also what a block the device's parser does not read goes
// through).
void EdgeCalculator::score_host_parsed(OverlapsParser& parser, std::vector<Overlap>& rejected, ParseCounters& pc) {
    size_t overlaps_per_vec = 250000;  // the reference batches 1,000,000 (:571); batch boundaries do not influence the result
    if (const char* e = getenv("HC_STAGE_BLOCK")) overlaps_per_vec = std::max<size_t>(1, (size_t)strtoull(e, nullptr, 10));  // experiment knob
    // The pipeline.  The caller's thread tokenises block k (on the parser's worker threads) and submits it to device
    // k mod N; the collector thread waits for the blocks in file order, finalises what the device kept of them and
    // runs the serial half.  Every stage consumes the blocks strictly in file order, so the graph, the counters and
    // nonedge_overlaps.txt are those of the sequential loop.  Up to two blocks per device are in flight.
    const size_t N = m_dev.size();
    const size_t R = 2 * N + 1;  // parsed blocks alive at once: two per device in flight + the one being parsed
    ParsedBatch::RecStorage pinned;  // the parser writes its records where the device's DMA reads them
    pinned.ctx = m_ctx;
    pinned.alloc = [](void* ctx, size_t n) -> hc_cand_rec* {
        void* p = nullptr;
        check(hc_host_alloc((hc_ctx*)ctx, &p, n * sizeof(hc_cand_rec)), "hc_host_alloc");
        return (hc_cand_rec*)p;
    };
    pinned.release = [](void* ctx, hc_cand_rec* p) { hc_host_free((hc_ctx*)ctx, p); };
    struct Slot {
        ParsedBatch batch;
        uint64_t base = 0;
        hc_block* block = nullptr;  // nullptr: nothing was submitted (an empty block)
        explicit Slot(const ParsedBatch::RecStorage& st) : batch(st) {}
    };
    std::vector<std::unique_ptr<Slot>> ring;
    for (size_t r = 0; r < R; r++) ring.emplace_back(new Slot(pinned));

    std::mutex mu;
    std::condition_variable cv;
    size_t submitted = 0, consumed = 0;  // blocks
    bool producer_done = false;
    std::atomic<bool> collector_failed{false};
    FatalError collector_error{0, ""};
    double t_collect = 0;
    std::thread collector([&] {
        BlockOut out;
        for (size_t k = 0;; k++) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return submitted > k || producer_done; });
                if (submitted <= k) return;
            }
            Slot& sl = *ring[k % R];
            if (!collector_failed) {
                try {
                    const double t0 = now_s();
                    const hc_gather_row* rows = nullptr;
                    uint64_t n_rows = 0;
                    if (sl.block) check(hc_block_wait(sl.block, &rows, &n_rows), "hc_block_wait");
                    finalize_block(sl.batch, rows, n_rows, sl.base, out);
                    t_collect += now_s() - t0;
                    consume_block(out);
                } catch (const FatalError& e) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!collector_failed) {
                        collector_error = e;
                        collector_failed = true;
                    }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!collector_failed) {
                        collector_error = FatalError{HC_ERR_NOMEM, e.what()};
                        collector_failed = true;
                    }
                }
            } else if (sl.block) {  // drain: the block object must not stay in flight
                const hc_gather_row* rows = nullptr;
                uint64_t n_rows = 0;
                (void)hc_block_wait(sl.block, &rows, &n_rows);
            }
            {
                std::lock_guard<std::mutex> g(mu);
                consumed = k + 1;
            }
            cv.notify_all();
        }
    });
    auto stop_collector = [&] {
        {
            std::lock_guard<std::mutex> g(mu);
            producer_done = true;
        }
        cv.notify_all();
        if (collector.joinable()) collector.join();
    };
    uint64_t base = 0;
    try {
        for (size_t k = 0;; k++) {
            {  // the slot's previous block (k - R) has been consumed
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return consumed + R > k; });
                if (collector_failed) break;
            }
            Slot& sl = *ring[k % R];
            const double t0 = now_s();
            const bool more = parser.next_batch(sl.batch, overlaps_per_vec, rejected, pc, /*print_malformed=*/true);
            stats.t_parse += now_s() - t0;
            if (!more) break;
            sl.base = base;
            sl.block = nullptr;
            const size_t n = sl.batch.size();
            if (n) {  // :636-644
                Device& dev = m_dev[k % N];
                const size_t j = (k / N) % 2;
                {  // the block object's previous user (block k - 2N) has been waited for
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return consumed + 2 * N > k; });
                }
                if (n > m_block_cap) {  // the first block sizes the objects; a later, larger one re-creates them all (never in flight then: see below)
                    {
                        std::unique_lock<std::mutex> g(mu);
                        cv.wait(g, [&] { return consumed == k; });
                    }
                    for (Device& d : m_dev)
                        for (hc_block*& b : d.blk) {
                            hc_block_destroy(b);
                            b = nullptr;
                        }
                    m_block_cap = n + n / 4;
                }
                if (!dev.blk[j]) check(hc_block_create(dev.ctx, m_block_cap, &dev.blk[j]), "hc_block_create");
                check(hc_block_submit(dev.blk[j], sl.batch.recs, n, base), "hc_block_submit");
                sl.block = dev.blk[j];
                stats.scored += n;
                base += n;
            }
            {
                std::lock_guard<std::mutex> g(mu);
                submitted = k + 1;
            }
            cv.notify_all();
        }
    } catch (...) {
        stop_collector();
        throw;
    }
    stop_collector();
    if (collector_failed) throw collector_error;
    stats.t_score = t_collect;
}

// src/EdgeCalculator.cpp:561-666
void EdgeCalculator::run_stage(bool then_sort) {
    const double t_stage0 = now_s();
    auto stage_lap = [&](const char* what) {
        if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] %s at %.3f s\n", what, now_s() - t_stage0);
    };
    collect_read_info();  // vertex ids may have been assigned since the last call
    stage_lap("read info collected");
    stats = Stats();
    // An empty graph (every pipeline call) takes the bulk path: the admitted candidates are collected in sequence
    // order and resolved at once after the last block.  A graph that already holds edges, or HC_INSERT_MODE=serial,
    // takes the per-edge insert of the reference's serial half.
    m_collect = !m_serial_insert && overlap_graph->getEdgeCount() == 0 &&
                EdgeSlotIndex::representable(overlap_graph->adj_out.size(), overlap_graph->adj_out.size());
    m_admitted.clear();
    m_device_resolve = m_collect && !m_host_resolve && !program_settings.add_duplicates;  // vertices by orientation: host route
    for (const ReadInfo& x : m_read_info)
        if (x.vertex_set && x.vertex >= ((node_id_t)1 << 31)) m_device_resolve = false;
    if (m_device_resolve) check(hc_graph_begin(m_ctx), "hc_graph_begin");
    struct AppenderGuard {  // (an exception on the way: the thread is joined before m_admitted goes)
        EdgeCalculator* self;
        ~AppenderGuard() { self->finish_appender(false); }
    } appender_guard{this};
    if (m_device_resolve) start_appender();
    std::remove("nonedge_overlaps.txt");  // :566 — in the cwd, whatever --output says (kept as is)
    std::vector<Overlap> rejected;
    // (closed by the clean-up thread: unmapping the 4 GB file with the worker threads alive takes 20 ms)
    if (m_lines_override && !m_text_override) m_text_override = std::make_shared<std::string>();  // (the parser only lends its id index then)
    std::unique_ptr<OverlapsParser> parser_owner(m_text_override
                                                     ? new OverlapsParser(m_text_override, program_settings, *fastq_storage, m_pool.get())
                                                     : new OverlapsParser(program_settings.overlaps_file, program_settings, *fastq_storage, m_pool.get()));
    struct CloseLater {
        std::unique_ptr<OverlapsParser>& p;
        EdgeCalculator* self;
        ~CloseLater() { self->defer_cleanup([q = p.release()] { delete q; }); }
    } close_later{parser_owner, this};
    OverlapsParser& parser = *parser_owner;
    stage_lap("overlaps file open");
    if (!parser.is_open()) throw FatalError{HC_ERR_IO, "Unable to open overlaps file"};  // :662-665
    if (program_settings.verbose) puts("reading overlaps file... ");
    ParseCounters pc;
    if (m_lines_override) score_device_lines(parser, rejected, pc);
    else if (m_host_parse) score_host_parsed(parser, rejected, pc);
    else score_device_parsed(parser, rejected, pc);
    finish_appender(true);
    stage_lap("all blocks scored and consumed");
    bool sorted_already = false;
    if (m_collect) {
        const double tr = now_s();
        if (m_device_resolve) {
            resolve_on_device(then_sort);
            sorted_already = then_sort;
        } else {
            resolve_on_host();
        }
        {  // 180 MB of admitted records at C3: freed off the caller's clock
            auto* old = new std::vector<std::vector<hc_admit_rec>>(std::move(m_admitted));
            m_admitted.clear();
            defer_cleanup([old] { delete old; });
        }
        if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] resolve total %.3f s\n", now_s() - tr);
        stage_lap("graph resolved");
        m_collect = false;
        stats.t_insert += now_s() - tr;
    }
    if (program_settings.add_duplicates) {  // :650-652
        unsigned int built = 0, doubles = 0;
        overlap_graph->addEquivalentEdges(&built, &doubles);
        if (program_settings.verbose) {
            printf("Number of equivalent edges built: %u\n", built);
            printf("Number of duplicates: %u\n", doubles);
        }
    }
    if (then_sort && !sorted_already) {  // the serial / host-resolved paths: sortEdges as its own pass
        std::vector<uint32_t> len(fastq_storage->m_read_vec.size());
        for (size_t r = 0; r < len.size(); r++) len[r] = fastq_storage->m_read_vec[r]->get_len();
        overlap_graph->sortEdges(len.data(), program_settings.n_threads);
    }
    stats.lines_read = pc.lines_read;
    stats.malformed = pc.malformed;
    stats.self_overlaps = pc.self_overlaps;
    stats.prefilter_rejected = pc.prefilter_rejected;
    stats.silently_dropped = pc.silently_dropped;
    if (program_settings.verbose) {  // :646-649
        printf("Number of self-overlapping reads: %u\n", self_overlap_count);
        printf("Number of inclusion edges: %u\n", inclusion_count);
    }
    const double t0 = now_s();
    FILE* fo = fopen((program_settings.output_dir + "nonedge_overlaps.txt").c_str(), "a");  // :654-660
    if (fo) {
        char linebuf[192];
        std::string buf;
        for (const Overlap& o : rejected) buf.append(linebuf, o.write_line(linebuf));
        fwrite(buf.data(), 1, buf.size(), fo);
        fclose(fo);
    }
    stats.t_write += now_s() - t0;
    stage_lap("construct_edges body done (destructors follow)");
}

void EdgeCalculator::construct_edges_from_reads(double err_rate, uint32_t min_overlap, uint32_t find_flags, bool then_sort, uint64_t* n_found,
                                                uint64_t* n_lines) {
    const double t0 = now_s();
    uint64_t found = 0, lines = 0;
    // Overlaps that come straight from the finder mostly survive the scoring (all of them at err_rate 0): the first text blocks get
    // row buffers for lines of 32 bytes now, beside the finder's kernels, instead of each growing its own inside its first wait
    // (page-locking 68 MB a block; a block with still more rows, and the later blocks of a long text, grow on demand as ever)
    int grow_rc = HC_OK;
    std::thread grower([&] {
        bind_here();
        for (Device& d : m_dev)
            for (size_t k = 0; k < d.tblk.size() && k < 6; k++)
                if (d.tblk[k] && grow_rc == HC_OK) grow_rc = hc_textblock_reserve_rows(d.tblk[k], m_text_block / 32);
    });
    struct Join {
        std::thread& t;
        ~Join() {
            if (t.joinable()) t.join();
        }
    } join_grower{grower};
    check(hc_find_overlaps(m_ctx, err_rate, min_overlap, find_flags, nullptr, 0, &found), "hc_find_overlaps");
    grower.join();
    check(grow_rc, "hc_textblock_reserve_rows");
    const double t1 = now_s();
    auto text = std::make_shared<std::string>();
    check(hc_found_to_overlaps_text(m_ctx, fastq_storage->m_readcount_single, fastq_storage->m_readcount_paired, *text, &lines),
          "hc_found_to_overlaps");
    const double t2 = now_s();
    if (n_found) *n_found = found;
    if (n_lines) *n_lines = lines;
    struct Reset {
        std::shared_ptr<const std::string>& p;
        ~Reset() { p.reset(); }
    } reset{m_text_override};
    m_text_override = text;
    run_stage(then_sort);
    if (getenv("HC_STAGE_TIMING"))
        fprintf(stderr, "[hc stage] reads -> graph: find %.3f s (%lu SFO records), ingest %.3f s (%lu lines, %zu bytes of text in memory), construct %.3f s\n",
                t1 - t0, (unsigned long)found, t2 - t1, (unsigned long)lines, text->size(), now_s() - t2);
}

void EdgeCalculator::construct_edges_from_store(double err_rate, uint32_t min_overlap, uint32_t find_flags, bool then_sort, uint64_t* n_found,
                                                uint64_t* n_lines, int* device_route) {
    const double t0 = now_s();
    uint64_t found = 0, lines = 0;
    int grow_rc = HC_OK;
    std::thread grower([&] {  // row buffers for lines that mostly survive, beside the finder's kernels (as construct_edges_from_reads)
        bind_here();
        Device& d = m_dev[0];
        for (size_t k = 0; k < d.tblk.size() && k < 6; k++)
            if (d.tblk[k] && grow_rc == HC_OK) grow_rc = hc_textblock_reserve_rows(d.tblk[k], m_text_block / 32);
    });
    struct Join {
        std::thread& t;
        ~Join() {
            if (t.joinable()) t.join();
        }
    } join_grower{grower};
    check(hc_find_overlaps(m_ctx, err_rate, min_overlap, find_flags, nullptr, 0, &found), "hc_find_overlaps");
    grower.join();
    check(grow_rc, "hc_textblock_reserve_rows");
    const double t1 = now_s();
    if (!run_stage_from_found(then_sort, &lines)) {  // the host's matcher owns the script's errors
        if (device_route) *device_route = 0;
        construct_edges_from_reads(err_rate, min_overlap, find_flags, then_sort, n_found, n_lines);
        return;
    }
    if (device_route) *device_route = 1;
    if (n_found) *n_found = found;
    if (n_lines) *n_lines = lines;
    if (getenv("HC_STAGE_TIMING"))
        fprintf(stderr, "[hc stage] reads -> graph on the device: find %.3f s (%lu SFO records), ingest + construct %.3f s (%lu lines, none of them text)\n",
                t1 - t0, (unsigned long)found, now_s() - t1, (unsigned long)lines);
}

bool EdgeCalculator::run_stage_from_found(bool then_sort, uint64_t* n_lines) {
    const hc_line_rec* d_lines = nullptr;
    uint64_t lines = 0;
    const double t0 = now_s();
    const int rc = hc_found_to_lines_device(m_ctx, fastq_storage->m_readcount_single, fastq_storage->m_readcount_paired, &d_lines, &lines);
    if (rc == HC_ERR_NOT_ON_DEVICE) return false;
    check(rc, "hc_found_to_lines_device");
    if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] SFO ingest on the device: %.3f s (%lu lines)\n", now_s() - t0, (unsigned long)lines);
    if (n_lines) *n_lines = lines;
    struct Reset {
        EdgeCalculator* self;
        ~Reset() {
            self->m_lines_override = nullptr;
            self->m_lines_override_n = 0;
            self->m_text_override.reset();
        }
    } reset{this};
    static const hc_line_rec none{};
    m_lines_override = lines ? d_lines : &none;  // (no lines at all: an empty file — the stage still runs, over nothing)
    m_lines_override_n = lines;
    run_stage(then_sort);
    return true;
}

void EdgeCalculator::construct_edges_from_sfo(const std::string& sfo_path, bool then_sort, uint64_t* n_records, uint64_t* n_lines, int* device_route) {
    // the file mapped read-only (a pipe or another unmappable input is read whole)
    const int fd = open(sfo_path.c_str(), O_RDONLY);
    if (fd < 0) throw FatalError{HC_ERR_IO, "cannot open " + sfo_path};
    struct stat stt;
    std::string owned;
    const char* text = nullptr;
    size_t bytes = 0;
    void* map = nullptr;
    if (fstat(fd, &stt) == 0 && S_ISREG(stt.st_mode) && stt.st_size > 0) {
        map = mmap(nullptr, (size_t)stt.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map == MAP_FAILED) map = nullptr;
    }
    if (map) {
        text = (const char*)map;
        bytes = (size_t)stt.st_size;
    } else {
        char buf[1 << 16];
        ssize_t k;
        while ((k = read(fd, buf, sizeof buf)) > 0) owned.append(buf, (size_t)k);
        text = owned.data();
        bytes = owned.size();
    }
    struct Unmap {
        void* p;
        size_t n;
        int fd;
        ~Unmap() {
            if (p) munmap(p, n);
            close(fd);
        }
    } unmap{map, bytes, fd};
    const long ns = (long)fastq_storage->m_readcount_single, np = (long)fastq_storage->m_readcount_paired;
    uint64_t lines = 0;
    {
        // a canonical file (what the tool writes): its text is read on the device, 64 MiB at a time (hc_set_found_from_sfo_text; HC_SFO_PARSE=host:
        // on the host's threads, round 6's first form, kept as a route), and the records take the finder's place
        const double tp0 = now_s();
        uint64_t n_rec = 0;
        int rc = HC_ERR_NOT_ON_DEVICE;
        const char* route = getenv("HC_SFO_PARSE");
        if (route && !strcmp(route, "host")) {
            std::vector<hc_sfo_rec, DefaultInitAllocator<hc_sfo_rec>> recs;
            if (sfo_text_to_records(text, bytes, recs)) {
                n_rec = recs.size();
                rc = hc_set_found_records(m_ctx, recs.data(), recs.size());
                auto spent = std::make_shared<std::vector<hc_sfo_rec, DefaultInitAllocator<hc_sfo_rec>>>(std::move(recs));  // 2 GB at config 3's size:
                defer_cleanup([spent]() mutable { spent.reset(); });                                                     // given back behind the caller's back
            }
        } else if (!getenv("HC_SFO_TEXT_GENERAL")) {
            rc = hc_set_found_from_sfo_text(m_ctx, text, bytes, &n_rec);
        }
        if (rc != HC_ERR_NOT_ON_DEVICE) {
            check(rc, "hc_set_found_from_sfo_text");
            if (n_records) *n_records = n_rec;
            if (getenv("HC_STAGE_TIMING")) fprintf(stderr, "[hc stage] SFO file: %zu bytes -> %lu records on the device in %.3f s\n", bytes, (unsigned long)n_rec, now_s() - tp0);
            if (run_stage_from_found(then_sort, &lines)) {
                if (device_route) *device_route = 1;
                if (n_lines) *n_lines = lines;
                return;
            }
        } else if (n_records) {
            *n_records = 0;  // (not counted: the general path reads the file line by line)
        }
    }
    // any other file, or an input the device does not decide: the host's ingest, its text in memory, the text blocks
    auto out = std::make_shared<std::string>(sfo_to_overlaps(text, bytes, ns, np, lines));
    if (device_route) *device_route = 0;
    if (n_lines) *n_lines = lines;
    struct Reset {
        std::shared_ptr<const std::string>& p;
        ~Reset() { p.reset(); }
    } reset{m_text_override};
    m_text_override = out;
    run_stage(then_sort);
}

}  // namespace hc
