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
// The sort is a stable sort on (reference id with unplaced last, position, strand), samtools' coordinate order: in memory while
// the records fit the budget (CLM_SORT_MEM_MB, default a quarter of the host's RAM clamped to 256 MiB .. 16 GiB), otherwise sorted
// runs are spilled to temporary BGZF files next to the output and merged k-way (ties broken by run number = input order, so
// the result is the same stable order whatever the budget) -- `samtools sort` does the same with its -m budget.
// Input may also be SAM text (reference: file_mode "r" for any suffix but .bam, :127): records are encoded to BAM here.
// Reads WITHOUT a reference placement (refID < 0): the reference iterates `bam_file.fetch()` (:131), which for BAM walks the
// index reference by reference and therefore never yields them (and needs the .bai); for SAM text it yields every record.
// The filter reproduces exactly that output -- unplaced records of a BAM are left out and counted -- without needing an index.
#include "chimeralm_feed.h"
#include "chimeralm_hip.h"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <numeric>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <unistd.h>

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

void put16(std::vector<uint8_t>& o, uint32_t v) { o.push_back((uint8_t)v), o.push_back((uint8_t)(v >> 8)); }
void put32(std::vector<uint8_t>& o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
void put64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(v >> (8 * i))); }

// sort key: (reference id, unplaced last | position + 1 | reverse strand)
uint64_t sort_key(const uint8_t* rec /* at the length prefix */) {
    const int32_t tid = le32(rec + 4), pos = le32(rec + 8);
    const uint16_t flag = le16(rec + 4 + 14);
    const uint64_t t = tid < 0 ? 0xffffffffull : (uint64_t)(uint32_t)tid;
    return (t << 32) | ((uint64_t)(uint32_t)(pos + 1) << 1) | ((flag & 0x10) ? 1u : 0u);
}

size_t sort_budget_bytes() {
    if (const char* e = std::getenv("CLM_SORT_MEM_MB")) {
        const long v = std::atol(e);
        if (v >= 1) return (size_t)v << 20;
    }
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    size_t q = (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz / 4 : (size_t)1 << 30;
    const size_t lo = (size_t)256 << 20, hi = (size_t)16 << 30;
    return q < lo ? lo : q > hi ? hi : q;
}

// ---- SAM text -> BAM (SAM specification sections 1.3, 1.4 and 4.2) ------------------------------------------------------
struct SamInput {
    std::ifstream in;
    std::string path, err, line;
    std::vector<uint8_t> header_blob;                     // BAM header: magic, l_text, text, n_ref, references
    std::unordered_map<std::string, int32_t> ref_id;
    std::string pending;                                  // first alignment line, read while scanning the header
    bool has_pending = false;
    int64_t line_no = 0;

    bool open(const std::string& p) {
        path = p;
        in.open(p);
        if (!in) {
            err = p + ": cannot open";
            return false;
        }
        std::string text;
        std::vector<std::pair<std::string, int32_t>> refs;
        while (std::getline(in, line)) {
            ++line_no;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (line[0] != '@') {
                pending = line;
                has_pending = true;
                break;
            }
            text += line;
            text += '\n';
            if (line.rfind("@SQ", 0) == 0) {
                std::string sn;
                long ln = -1;
                size_t b = 0;
                while (b < line.size()) {
                    size_t e = line.find('\t', b);
                    if (e == std::string::npos) e = line.size();
                    if (line.compare(b, 3, "SN:") == 0) sn = line.substr(b + 3, e - b - 3);
                    if (line.compare(b, 3, "LN:") == 0) ln = std::atol(line.c_str() + b + 3);
                    b = e + 1;
                }
                if (sn.empty() || ln < 0) {
                    err = p + ": @SQ line without SN / LN";
                    return false;
                }
                ref_id[sn] = (int32_t)refs.size();
                refs.emplace_back(sn, (int32_t)ln);
            }
        }
        header_blob = {'B', 'A', 'M', 1};
        put32(header_blob, (uint32_t)text.size());
        header_blob.insert(header_blob.end(), text.begin(), text.end());
        put32(header_blob, (uint32_t)refs.size());
        for (auto& r : refs) {
            put32(header_blob, (uint32_t)r.first.size() + 1);
            header_blob.insert(header_blob.end(), r.first.begin(), r.first.end());
            header_blob.push_back(0);
            put32(header_blob, (uint32_t)r.second);
        }
        return true;
    }

    int fail(const std::string& what) {                     // malformed line: message with its number, -1 for next()
        err = path + ": line " + std::to_string(line_no) + ": " + what;
        return -1;
    }

    // one alignment line -> BAM record with its length prefix in `out`; 1 = ok, 0 = end of file, -1 = error
    int next(std::vector<uint8_t>& out) {
        for (;;) {
            if (has_pending) {
                line.swap(pending);
                has_pending = false;
            } else {
                if (!std::getline(in, line)) return 0;
                ++line_no;
                if (!line.empty() && line.back() == '\r') line.pop_back();
            }
            if (!line.empty()) break;
        }
        std::vector<std::pair<const char*, size_t>> f;
        for (size_t b = 0; b <= line.size();) {
            size_t e = line.find('\t', b);
            if (e == std::string::npos) e = line.size();
            f.emplace_back(line.data() + b, e - b);
            b = e + 1;
        }
        if (f.size() < 11) return fail("fewer than 11 fields");
        auto str = [&](int i) { return std::string(f[(size_t)i].first, f[(size_t)i].second); };
        auto refid = [&](const std::string& n, int32_t same, int32_t& o) -> bool {
            if (n == "*") { o = -1; return true; }
            if (n == "=") { o = same; return true; }
            auto it = ref_id.find(n);
            if (it == ref_id.end()) return false;
            o = it->second;
            return true;
        };
        const std::string qname = str(0), rname = str(2), cigar = str(5), rnext = str(6), seq = str(9), qual = str(10);
        const long flag = std::atol(str(1).c_str()), pos = std::atol(str(3).c_str()) - 1, mapq = std::atol(str(4).c_str()),
                   pnext = std::atol(str(7).c_str()) - 1, tlen = std::atol(str(8).c_str());
        int32_t tid = -1, ntid = -1;
        if (!refid(rname, -1, tid)) return fail("unknown reference " + rname);
        if (!refid(rnext, tid, ntid)) return fail("unknown mate reference " + rnext);
        if (qname.size() > 254) return fail("read name longer than 254");
        std::vector<uint32_t> cg;
        if (cigar != "*") {
            static const char* OPS = "MIDNSHP=X";
            size_t i = 0;
            while (i < cigar.size()) {
                uint64_t n = 0;
                size_t j = i;
                while (j < cigar.size() && cigar[j] >= '0' && cigar[j] <= '9') n = n * 10 + (uint64_t)(cigar[j++] - '0');
                const char* op = j < cigar.size() ? std::strchr(OPS, cigar[j]) : nullptr;
                if (j == i || !op || !*op || n >= (1u << 28)) return fail("bad CIGAR " + cigar);
                cg.push_back((uint32_t)(n << 4) | (uint32_t)(op - OPS));
                i = j + 1;
            }
        }
        const size_t l_seq = seq == "*" ? 0 : seq.size();
        if (qual != "*" && qual.size() != l_seq) return fail("SEQ and QUAL differ in length");
        std::vector<uint8_t> body;
        put32(body, (uint32_t)tid);
        put32(body, (uint32_t)(int32_t)pos);
        body.push_back((uint8_t)(qname.size() + 1));
        body.push_back((uint8_t)mapq);
        int64_t span = 0;
        for (uint32_t v : cg) {
            const unsigned op = v & 15;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;
        }
        const int64_t end = ((flag & 4) || span <= 0) ? pos + 1 : pos + span;
        put16(body, (uint32_t)reg2bin(pos, end));
        put16(body, (uint32_t)cg.size());
        put16(body, (uint32_t)flag);
        put32(body, (uint32_t)l_seq);
        put32(body, (uint32_t)ntid);
        put32(body, (uint32_t)(int32_t)pnext);
        put32(body, (uint32_t)(int32_t)tlen);
        body.insert(body.end(), qname.begin(), qname.end());
        body.push_back(0);
        for (uint32_t v : cg) put32(body, v);
        static const char* CODES = "=ACMGRSVTWYHKDBN";
        for (size_t i = 0; i < l_seq; i += 2) {
            auto code = [&](char ch) -> unsigned {
                const char* q = std::strchr(CODES, (char)std::toupper((unsigned char)ch));
                return (q && *q) ? (unsigned)(q - CODES) : 15u;
            };
            body.push_back((uint8_t)((code(seq[i]) << 4) | (i + 1 < l_seq ? code(seq[i + 1]) : 0u)));
        }
        for (size_t i = 0; i < l_seq; ++i) body.push_back(qual == "*" ? 0xff : (uint8_t)(qual[i] - 33));
        for (size_t k = 11; k < f.size(); ++k) {                                   // TAG:TYPE:VALUE
            const char* t = f[k].first;
            const size_t n = f[k].second;
            if (n == 0) continue;
            if (n < 5 || t[2] != ':' || t[4] != ':') return fail("bad optional field");
            const std::string val(t + 5, n - 5);
            body.push_back((uint8_t)t[0]);
            body.push_back((uint8_t)t[1]);
            switch (t[3]) {
                case 'A': body.push_back('A'); body.push_back(val.empty() ? 0 : (uint8_t)val[0]); break;
                case 'i': {                                                        // smallest type that holds it, as htslib
                    const long long x = std::atoll(val.c_str());
                    if (x < 0) {
                        if (x >= -128) body.push_back('c'), body.push_back((uint8_t)(int8_t)x);
                        else if (x >= -32768) body.push_back('s'), put16(body, (uint32_t)(int32_t)x);
                        else body.push_back('i'), put32(body, (uint32_t)(int32_t)x);
                    } else {
                        if (x <= 255) body.push_back('C'), body.push_back((uint8_t)x);
                        else if (x <= 65535) body.push_back('S'), put16(body, (uint32_t)x);
                        else body.push_back('I'), put32(body, (uint32_t)x);
                    }
                    break;
                }
                case 'f': {
                    const float v = std::strtof(val.c_str(), nullptr);
                    uint32_t u;
                    std::memcpy(&u, &v, 4);
                    body.push_back('f');
                    put32(body, u);
                    break;
                }
                case 'Z': case 'H':
                    body.push_back((uint8_t)t[3]);
                    body.insert(body.end(), val.begin(), val.end());
                    body.push_back(0);
                    break;
                case 'B': {
                    if (val.empty()) return fail("empty B array");
                    const char sub = val[0];
                    std::vector<std::string> items;
                    for (size_t b = 1; b < val.size();) {
                        if (val[b] == ',') ++b;
                        size_t e = val.find(',', b);
                        if (e == std::string::npos) e = val.size();
                        if (e > b) items.push_back(val.substr(b, e - b));
                        b = e;
                    }
                    body.push_back('B');
                    body.push_back((uint8_t)sub);
                    put32(body, (uint32_t)items.size());
                    for (auto& it : items) {
                        if (sub == 'f') {
                            const float v = std::strtof(it.c_str(), nullptr);
                            uint32_t u;
                            std::memcpy(&u, &v, 4);
                            put32(body, u);
                        } else {
                            const long long x = std::atoll(it.c_str());
                            if (sub == 'c' || sub == 'C') body.push_back((uint8_t)x);
                            else if (sub == 's' || sub == 'S') put16(body, (uint32_t)x);
                            else if (sub == 'i' || sub == 'I') put32(body, (uint32_t)x);
                            else return fail("bad B array subtype");
                        }
                    }
                    break;
                }
                default: return fail(std::string("unknown optional field type ") + t[3]);
            }
        }
        out.clear();
        put32(out, (uint32_t)body.size());
        out.insert(out.end(), body.begin(), body.end());
        return 1;
    }
};

// ---- coordinate-sorted output with its index, fed record by record in final order ---------------------------------------
struct SortedWriter {
    struct RefIdx {
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> lin;
        uint64_t beg = 0, end = 0, n_mapped = 0, n_unmapped = 0;
        bool any = false;
    };
    clmbgzf::ParallelWriter wr;                                // voffset() is logical until finish(): resolved in write_bai
    std::vector<RefIdx> idx;
    uint64_t n_no_coor = 0, n_records = 0;

    bool open(const std::string& path, const std::vector<uint8_t>& header, size_t n_ref) {
        idx.resize(n_ref);
        return wr.open(path, bam_threads()) && wr.write(header.data(), header.size()) && wr.flush_block();
    }
    bool put(const uint8_t* r, size_t len) {
        if (!wr.align_block(len)) return false;
        const uint64_t v0 = wr.voffset();
        if (!wr.write(r, len)) return false;
        const uint64_t v1 = wr.voffset();
        ++n_records;
        const int32_t tid = le32(r + 4), pos = le32(r + 8);
        const uint16_t flag = le16(r + 4 + 14);
        if (tid < 0 || (size_t)tid >= idx.size() || pos < 0) {
            ++n_no_coor;
            return true;
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
        return true;
    }
    bool write_bai(const std::string& bai_path) {
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
        FILE* f = std::fopen(bai_path.c_str(), "wb");
        return f && std::fwrite(bai.data(), 1, bai.size(), f) == bai.size() && std::fclose(f) == 0;
    }
};

// one sorted run on disk: the records (length-prefixed, as in a BAM) in BGZF members at zlib level 1, no header
struct RunReader {
    clmbgzf::Reader rd;
    const uint8_t* rec = nullptr;
    size_t len = 0;
    uint64_t key = 0;
    // 1 = a record is current, 0 = run exhausted, -1 = error
    int advance() {
        int rc = rd.need(4);
        if (rc <= 0) return rc;
        const int32_t bs = le32(rd.data());
        rc = rd.need((size_t)bs + 4);
        if (rc <= 0) return -1;
        rec = rd.data();
        len = (size_t)bs + 4;
        key = sort_key(rec);
        rd.advance(len);
        return 1;
    }
};

}  // namespace

extern "C" {

const char* clm_bam_last_error(void) { return g_err.c_str(); }

int clm_bam_filter_ex(const char* in_path, const char* out_bam, const char* const* drop_names, int64_t n_drop, int flags,
                    int64_t* kept, int64_t* dropped, int64_t* unplaced) {
    if (!in_path || !out_bam || (n_drop > 0 && !drop_names) || n_drop < 0) return bam_fail(CLM_E_INVALID, "clm_bam_filter: bad argument");
    const bool sam = (flags & CLM_BAM_INPUT_SAM) != 0;
    const bool keep_unplaced = sam || (flags & CLM_BAM_KEEP_UNPLACED) != 0;     // pysam fetch(): every record of a SAM file
    std::unordered_set<std::string> drop;
    drop.reserve((size_t)n_drop * 2 + 16);
    for (int64_t i = 0; i < n_drop; ++i) drop.insert(drop_names[i]);
    clmbgzf::ParallelReader rd;
    SamInput si;
    Header hd;
    if (sam) {
        if (!si.open(in_path)) return bam_fail(CLM_E_INVALID, si.err);
        hd.blob = si.header_blob;
    } else {
        if (!rd.open(in_path, bam_threads())) return bam_fail(CLM_E_INVALID, rd.err);
        if (read_header(rd, hd)) return bam_fail(CLM_E_INVALID, rd.err.empty() ? std::string(in_path) + ": not a BAM file" : rd.err);
    }
    clmbgzf::ParallelWriter wr;
    if (!wr.open(out_bam, bam_threads())) return bam_fail(CLM_E_INVALID, wr.err);
    if (!wr.write(hd.blob.data(), hd.blob.size())) return bam_fail(CLM_E_INVALID, wr.err);
    int64_t nk = 0, nd = 0, nu = 0;
    const uint8_t* rec = nullptr;
    size_t len = 0;
    std::vector<uint8_t> enc;
    int rc;
    for (;;) {
        if (sam) {
            rc = si.next(enc);
            rec = enc.data();
            len = enc.size();
        } else {
            rc = next_record(rd, rec, len);
        }
        if (rc != 1) break;
        const unsigned l_name = rec[4 + 8];
        const std::string name(reinterpret_cast<const char*>(rec + 4 + 32), l_name ? strnlen(reinterpret_cast<const char*>(rec + 4 + 32), l_name) : 0);
        if (!keep_unplaced && le32(rec + 4) < 0) {               // no reference placement: an index walk never reaches it
            ++nu;
            continue;
        }
        if (drop.count(name)) {
            ++nd;
            continue;
        }
        if (!wr.write(rec, len)) return bam_fail(CLM_E_INVALID, wr.err);
        ++nk;
    }
    if (rc < 0) {
        std::remove(out_bam);                                   // like the reference (:140-144): no partial output left behind
        return bam_fail(CLM_E_INVALID, sam ? si.err : rd.err);
    }
    if (!wr.finish()) return bam_fail(CLM_E_INVALID, wr.err);
    if (kept) *kept = nk;
    if (dropped) *dropped = nd;
    if (unplaced) *unplaced = nu;
    return CLM_OK;
}

int clm_bam_filter(const char* in_bam, const char* out_bam, const char* const* drop_names, int64_t n_drop, int64_t* kept,
                   int64_t* dropped) {
    return clm_bam_filter_ex(in_bam, out_bam, drop_names, n_drop, 0, kept, dropped, nullptr);
}

int clm_bam_sort_index(const char* in_bam, const char* out_sorted_bam, const char* out_bai, int64_t* n_records) {
    if (!in_bam || !out_sorted_bam) return bam_fail(CLM_E_INVALID, "clm_bam_sort_index: bad argument");
    clmbgzf::ParallelReader rd;
    if (!rd.open(in_bam, bam_threads())) return bam_fail(CLM_E_INVALID, rd.err);
    Header hd;
    if (read_header(rd, hd)) return bam_fail(CLM_E_INVALID, rd.err.empty() ? std::string(in_bam) + ": not a BAM file" : rd.err);
    // ---- records are pooled up to the memory budget; a full pool is sorted and spilled as one run
    const size_t budget = sort_budget_bytes();
    std::vector<uint8_t> pool;
    std::vector<size_t> off, lens;
    std::vector<uint64_t> key;
    std::vector<size_t> order;
    std::vector<std::string> runs;
    auto cleanup = [&] { for (auto& r : runs) std::remove(r.c_str()); };
    auto sort_pool = [&] {
        order.resize(off.size());
        std::iota(order.begin(), order.end(), (size_t)0);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    };
    auto spill = [&]() -> bool {
        sort_pool();
        const std::string path = std::string(out_sorted_bam) + ".tmp." + std::to_string(runs.size()) + ".run";
        clmbgzf::ParallelWriter w;
        w.level = 1;
        runs.push_back(path);
        if (!w.open(path, bam_threads())) return false;
        for (size_t oi : order)
            if (!w.write(pool.data() + off[oi], lens[oi])) return false;
        if (!w.finish()) return false;
        pool.clear(), off.clear(), lens.clear(), key.clear();
        return true;
    };
    const uint8_t* rec;
    size_t len;
    int rc;
    while ((rc = next_record(rd, rec, len)) == 1) {
        if (!pool.empty() && pool.size() + len + 24 * (off.size() + 1) > budget && !spill()) {
            cleanup();
            return bam_fail(CLM_E_INVALID, std::string(out_sorted_bam) + ": cannot write a temporary sort run");
        }
        off.push_back(pool.size());
        lens.push_back(len);
        pool.insert(pool.end(), rec, rec + len);
        key.push_back(sort_key(rec));
    }
    if (rc < 0) {
        cleanup();
        return bam_fail(CLM_E_INVALID, rd.err);
    }
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
    SortedWriter sw;
    if (!sw.open(out_sorted_bam, nh, hd.ref_len.size())) {
        cleanup();
        return bam_fail(CLM_E_INVALID, sw.wr.err);
    }
    bool ok = true;
    if (runs.empty()) {                                          // everything fitted: no temporary file at all
        sort_pool();
        for (size_t oi : order)
            if (!(ok = sw.put(pool.data() + off[oi], lens[oi]))) break;
    } else {
        if (!off.empty() && !spill()) {
            cleanup();
            return bam_fail(CLM_E_INVALID, std::string(out_sorted_bam) + ": cannot write a temporary sort run");
        }
        pool.shrink_to_fit();
        // k-way merge; equal keys leave in run order = input order (each run is itself stable)
        std::vector<std::unique_ptr<RunReader>> rr;
        using Item = std::pair<uint64_t, size_t>;
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
        for (size_t i = 0; i < runs.size() && ok; ++i) {
            rr.emplace_back(new RunReader());
            if (!rr[i]->rd.open(runs[i])) ok = false;
            else {
                const int a = rr[i]->advance();
                if (a < 0) ok = false;
                if (a == 1) heap.emplace(rr[i]->key, i);
            }
        }
        while (ok && !heap.empty()) {
            const size_t i = heap.top().second;
            heap.pop();
            if (!(ok = sw.put(rr[i]->rec, rr[i]->len))) break;
            const int a = rr[i]->advance();
            if (a < 0) ok = false;
            if (a == 1) heap.emplace(rr[i]->key, i);
        }
        rr.clear();
        cleanup();
        if (!ok && sw.wr.err.empty()) {
            std::remove(out_sorted_bam);                         // no partial output left behind, as clm_bam_filter_ex
            return bam_fail(CLM_E_INVALID, std::string(out_sorted_bam) + ": a temporary sort run could not be read back");
        }
    }
    if (!ok || !sw.wr.finish()) {
        std::remove(out_sorted_bam);                             // (unlinked while still open on the failure paths: fine on POSIX)
        return bam_fail(CLM_E_INVALID, sw.wr.err);
    }
    const std::string bai_path = out_bai ? std::string(out_bai) : std::string(out_sorted_bam) + ".bai";
    if (!sw.write_bai(bai_path)) {
        std::remove(out_sorted_bam);
        std::remove(bai_path.c_str());
        return bam_fail(CLM_E_INVALID, bai_path + ": cannot write the index");
    }
    if (n_records) *n_records = (int64_t)sw.n_records;
    return CLM_OK;
}

}  // extern "C"
