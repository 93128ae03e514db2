// bam_filter.cpp -- the step after `predict` (include/chimeralm_feed.h): drop the reads classified as chimera artifacts from
// the BAM, then coordinate-sort and index the result.
//
// Mirrors /root/reference/chimeralm/__main__.py:99-153 (filter_bam_by_predcition), which does this through pysam / samtools:
//   * every record whose query_name has prediction 1 is skipped -- all of them: primary, secondary and supplementary records
//     share the name (:131-134) -- everything else is copied unchanged into <bam>.filtered.bam with the input's header (:126-127);
//   * `pysam.sort` into <...>.sorted.bam and `pysam.index` (:147-153).
// BAM / BAI layouts: SAM/BAM specification sections 4.2 and 5.2 (binning index: 6-level bins of 2^29 .. 2^14 bases,
// bin = reg2bin(beg, end); per reference the bins with their chunks of virtual file offsets, a linear index of the smallest
// offset per 16 kbp window, a metadata pseudo-bin 37450, and the count of unplaced reads at the end).
// The sort is an in-memory stable sort on (reference id with unmapped last, position, strand), samtools' coordinate order.
#include "chimeralm_feed.h"
#include "chimeralm_hip.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <numeric>
#include <string>
#include <unordered_set>
#include <vector>

#include "bgzf.h"

namespace {

using clmbgzf::le16;
using clmbgzf::le32;

thread_local std::string g_err;

int bam_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// worker threads for the BGZF members of input and output (independent deflate streams): the host's cores less two, at most 8;
// CLM_BAM_THREADS overrides (tests compare 1 against several)
int bam_threads() {
    if (const char* e = std::getenv("CLM_BAM_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 256) return v;
    }
    const int t = (int)std::thread::hardware_concurrency() - 2;
    return t < 1 ? 1 : t > 8 ? 8 : t;
}

// header bytes (magic .. last reference) as one blob; n_ref and the reference lengths on the side
struct Header {
    std::vector<uint8_t> blob;
    std::vector<int32_t> ref_len;
    size_t text_off = 0, text_len = 0;
};

int read_header(clmbgzf::ParallelReader& rd, Header& h) {
    auto take = [&](size_t n) -> int {
        const int rc = rd.need(n);
        if (rc <= 0) return -1;
        h.blob.insert(h.blob.end(), rd.data(), rd.data() + n);
        rd.advance(n);
        return 0;
    };
    if (take(8) || std::memcmp(h.blob.data(), "BAM\1", 4) != 0) return -1;
    const int32_t l_text = le32(h.blob.data() + 4);
    if (l_text < 0 || take((size_t)l_text + 4)) return -1;
    h.text_off = 8;
    h.text_len = (size_t)l_text;
    const int32_t n_ref = le32(h.blob.data() + 8 + l_text);
    if (n_ref < 0) return -1;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (take(4)) return -1;
        const int32_t l_name = le32(h.blob.data() + h.blob.size() - 4);
        if (l_name < 0 || take((size_t)l_name + 4)) return -1;
        h.ref_len.push_back(le32(h.blob.data() + h.blob.size() - 4));
    }
    return 0;
}

// next alignment record incl. its 4-byte length prefix: 1 = ok (ptr/len valid until the next call), 0 = end, -1 = error
int next_record(clmbgzf::ParallelReader& rd, const uint8_t*& ptr, size_t& len) {
    int rc = rd.need(4);
    if (rc <= 0) return rc;
    const int32_t bs = le32(rd.data());
    if (bs < 32) {
        rd.err = rd.path + ": corrupt BAM record";
        return -1;
    }
    rc = rd.need((size_t)bs + 4);
    if (rc <= 0) {
        if (rd.err.empty()) rd.err = rd.path + ": BAM stream ends inside a record";
        return -1;
    }
    ptr = rd.data();
    len = (size_t)bs + 4;
    rd.advance(len);
    return 1;
}

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}

// reference bases covered by the alignment (CIGAR ops M, D, N, =, X), at least 1
int64_t ref_span(const uint8_t* rec /* after the length prefix */) {
    const unsigned l_name = rec[8], n_cigar = le16(rec + 12);
    const uint8_t* cg = rec + 32 + l_name;
    int64_t span = 0;
    for (unsigned i = 0; i < n_cigar; ++i) {
        const uint32_t v = (uint32_t)le32(cg + 4 * i);
        const unsigned op = v & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;
    }
    return span > 0 ? span : 1;
}

void put32(std::vector<uint8_t>& o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
void put64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(v >> (8 * i))); }

}  // namespace

extern "C" {

const char* clm_bam_last_error(void) { return g_err.c_str(); }

int clm_bam_filter(const char* in_bam, const char* out_bam, const char* const* drop_names, int64_t n_drop, int64_t* kept,
                   int64_t* dropped) {
    if (!in_bam || !out_bam || (n_drop > 0 && !drop_names) || n_drop < 0) return bam_fail(CLM_E_INVALID, "clm_bam_filter: bad argument");
    std::unordered_set<std::string> drop;
    drop.reserve((size_t)n_drop * 2 + 16);
    for (int64_t i = 0; i < n_drop; ++i) drop.insert(drop_names[i]);
    clmbgzf::ParallelReader rd;
    if (!rd.open(in_bam, bam_threads())) return bam_fail(CLM_E_INVALID, rd.err);
    Header hd;
    if (read_header(rd, hd)) return bam_fail(CLM_E_INVALID, rd.err.empty() ? std::string(in_bam) + ": not a BAM file" : rd.err);
    clmbgzf::ParallelWriter wr;
    if (!wr.open(out_bam, bam_threads())) return bam_fail(CLM_E_INVALID, wr.err);
    if (!wr.write(hd.blob.data(), hd.blob.size())) return bam_fail(CLM_E_INVALID, wr.err);
    int64_t nk = 0, nd = 0;
    const uint8_t* rec;
    size_t len;
    int rc;
    while ((rc = next_record(rd, rec, len)) == 1) {
        const unsigned l_name = rec[4 + 8];
        const std::string name(reinterpret_cast<const char*>(rec + 4 + 32), l_name ? strnlen(reinterpret_cast<const char*>(rec + 4 + 32), l_name) : 0);
        if (drop.count(name)) {
            ++nd;
            continue;
        }
        if (!wr.write(rec, len)) return bam_fail(CLM_E_INVALID, wr.err);
        ++nk;
    }
    if (rc < 0) {
        std::remove(out_bam);                                   // like the reference (:140-144): no partial output left behind
        return bam_fail(CLM_E_INVALID, rd.err);
    }
    if (!wr.finish()) return bam_fail(CLM_E_INVALID, wr.err);
    if (kept) *kept = nk;
    if (dropped) *dropped = nd;
    return CLM_OK;
}

int clm_bam_sort_index(const char* in_bam, const char* out_sorted_bam, const char* out_bai, int64_t* n_records) {
    if (!in_bam || !out_sorted_bam) return bam_fail(CLM_E_INVALID, "clm_bam_sort_index: bad argument");
    clmbgzf::ParallelReader rd;
    if (!rd.open(in_bam, bam_threads())) return bam_fail(CLM_E_INVALID, rd.err);
    Header hd;
    if (read_header(rd, hd)) return bam_fail(CLM_E_INVALID, rd.err.empty() ? std::string(in_bam) + ": not a BAM file" : rd.err);
    // ---- all records in memory (the reference's samtools sort spills to disk; an external merge is future work)
    std::vector<uint8_t> pool;
    std::vector<size_t> off, lens;
    std::vector<uint64_t> key;
    const uint8_t* rec;
    size_t len;
    int rc;
    while ((rc = next_record(rd, rec, len)) == 1) {
        off.push_back(pool.size());
        lens.push_back(len);
        pool.insert(pool.end(), rec, rec + len);
        const int32_t tid = le32(rec + 4), pos = le32(rec + 8);
        const uint16_t flag = le16(rec + 4 + 14);
        const uint64_t t = tid < 0 ? 0xffffffffull : (uint64_t)(uint32_t)tid;
        key.push_back((t << 32) | ((uint64_t)(uint32_t)(pos + 1) << 1) | ((flag & 0x10) ? 1u : 0u));
    }
    if (rc < 0) return bam_fail(CLM_E_INVALID, rd.err);
    std::vector<size_t> order(off.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    // ---- header with @HD ... SO:coordinate
    std::string text(reinterpret_cast<const char*>(hd.blob.data() + hd.text_off), hd.text_len);
    while (!text.empty() && text.back() == '\0') text.pop_back();
    if (text.rfind("@HD", 0) == 0) {
        const size_t eol = text.find('\n');
        std::string line = text.substr(0, eol == std::string::npos ? text.size() : eol);
        const size_t so = line.find("\tSO:");
        if (so != std::string::npos) {
            const size_t end = line.find('\t', so + 1);
            line.replace(so, (end == std::string::npos ? line.size() : end) - so, "\tSO:coordinate");
        } else {
            line += "\tSO:coordinate";
        }
        text = line + (eol == std::string::npos ? "\n" : text.substr(eol));
    } else {
        text = "@HD\tVN:1.6\tSO:coordinate\n" + text;
    }
    std::vector<uint8_t> nh(hd.blob.begin(), hd.blob.begin() + 4);
    put32(nh, (uint32_t)text.size());
    nh.insert(nh.end(), text.begin(), text.end());
    nh.insert(nh.end(), hd.blob.begin() + (long)(hd.text_off + hd.text_len), hd.blob.end());
    clmbgzf::ParallelWriter wr;                                // voffset() is logical until finish(): resolved below
    if (!wr.open(out_sorted_bam, bam_threads())) return bam_fail(CLM_E_INVALID, wr.err);
    if (!wr.write(nh.data(), nh.size()) || !wr.flush_block()) return bam_fail(CLM_E_INVALID, wr.err);
    // ---- write in order, collecting the index
    struct RefIdx {
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> lin;
        uint64_t beg = 0, end = 0, n_mapped = 0, n_unmapped = 0;
        bool any = false;
    };
    std::vector<RefIdx> idx(hd.ref_len.size());
    uint64_t n_no_coor = 0;
    for (size_t oi : order) {
        const uint8_t* r = pool.data() + off[oi];
        if (!wr.align_block(lens[oi])) return bam_fail(CLM_E_INVALID, wr.err);
        const uint64_t v0 = wr.voffset();
        if (!wr.write(r, lens[oi])) return bam_fail(CLM_E_INVALID, wr.err);
        const uint64_t v1 = wr.voffset();
        const int32_t tid = le32(r + 4), pos = le32(r + 8);
        const uint16_t flag = le16(r + 4 + 14);
        if (tid < 0 || (size_t)tid >= idx.size() || pos < 0) {
            ++n_no_coor;
            continue;
        }
        RefIdx& x = idx[(size_t)tid];
        const int64_t beg = pos, end = (flag & 4) ? pos + 1 : pos + ref_span(r + 4);
        auto& chunks = x.bins[(uint32_t)reg2bin(beg, end)];
        if (!chunks.empty() && chunks.back().second == v0) chunks.back().second = v1;   // adjacent records of a bin: one chunk
        else chunks.emplace_back(v0, v1);
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (x.lin.size() <= w1) x.lin.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w)
            if (x.lin[w] == 0) x.lin[w] = v0;
        if (!x.any) x.beg = v0, x.any = true;
        x.end = v1;
        ((flag & 4) ? x.n_unmapped : x.n_mapped) += 1;
    }
    if (!wr.finish()) return bam_fail(CLM_E_INVALID, wr.err);
    // ---- BAI
    std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
    put32(bai, (uint32_t)idx.size());
    for (RefIdx& x : idx) {
        put32(bai, (uint32_t)(x.bins.size() + (x.any ? 1 : 0)));
        for (auto& kv : x.bins) {
            put32(bai, kv.first);
            put32(bai, (uint32_t)kv.second.size());
            for (auto& c : kv.second) put64(bai, wr.resolve(c.first)), put64(bai, wr.resolve(c.second));
        }
        if (x.any) {                                            // metadata pseudo-bin
            put32(bai, 37450);
            put32(bai, 2);
            put64(bai, wr.resolve(x.beg)), put64(bai, wr.resolve(x.end)), put64(bai, x.n_mapped), put64(bai, x.n_unmapped);
        }
        for (size_t w = 1; w < x.lin.size(); ++w)
            if (x.lin[w] == 0) x.lin[w] = x.lin[w - 1];         // windows no read starts in inherit the previous offset
        put32(bai, (uint32_t)x.lin.size());
        for (uint64_t v : x.lin) put64(bai, wr.resolve(v));
    }
    put64(bai, n_no_coor);
    const std::string bai_path = out_bai ? std::string(out_bai) : std::string(out_sorted_bam) + ".bai";
    FILE* f = std::fopen(bai_path.c_str(), "wb");
    if (!f || std::fwrite(bai.data(), 1, bai.size(), f) != bai.size() || std::fclose(f) != 0)
        return bam_fail(CLM_E_INVALID, bai_path + ": cannot write the index");
    if (n_records) *n_records = (int64_t)order.size();
    return CLM_OK;
}

}  // extern "C"
