// bgzf.h -- BGZF (blocked gzip) stream reader / writer shared by the BAM feeder and the BAM filter (host C++, zlib).
// Format: SAM/BAM specification section 4.1 -- a series of gzip members (<= 64 KiB of payload each) whose extra field carries
// the member size in a "BC" subfield; a BAM file ends with an empty member.  A "virtual file offset" of a byte of the inflated
// stream is (file offset of its member << 16) | offset inside the member's payload.
#pragma once
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace clmbgzf {

// One member's framing: reads the gzip header, the BC subfield and the compressed bytes + trailer into `zin`.
// 1 = ok (cdata bytes of deflate data, then CRC32 and ISIZE), 0 = clean end of file, -1 = error (err set)
inline int read_member(FILE* fp, const std::string& path, std::vector<uint8_t>& zin, size_t& cdata, uint32_t& crc,
                       uint32_t& isize, std::string& err) {
    uint8_t head[12];
    const size_t got = std::fread(head, 1, sizeof(head), fp);
    if (got == 0) return 0;
    if (got != sizeof(head) || head[0] != 31 || head[1] != 139 || head[2] != 8 || !(head[3] & 4)) {
        err = path + ": not a BGZF block (bad gzip member header)";
        return -1;
    }
    const unsigned xlen = head[10] | (head[11] << 8);
    std::vector<uint8_t> extra(xlen);
    if (std::fread(extra.data(), 1, xlen, fp) != xlen) {
        err = path + ": truncated BGZF extra field";
        return -1;
    }
    int bsize = -1;
    for (size_t p = 0; p + 4 <= xlen;) {
        const unsigned slen = extra[p + 2] | (extra[p + 3] << 8);
        if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = extra[p + 4] | (extra[p + 5] << 8);
        p += 4 + slen;
    }
    const long cd = (long)bsize + 1 - 12 - (long)xlen - 8;
    if (bsize < 0 || cd < 0) {
        err = path + ": BGZF block without a valid BC subfield";
        return -1;
    }
    cdata = (size_t)cd;
    zin.resize(cdata + 8);
    if (std::fread(zin.data(), 1, zin.size(), fp) != zin.size()) {
        err = path + ": truncated BGZF block";
        return -1;
    }
    const uint8_t* tail = zin.data() + cdata;
    crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    if (isize > 65536) {
        err = path + ": BGZF block claims more than 64 KiB of payload";
        return -1;
    }
    return 1;
}

// Inflates one member's `cdata` bytes into out[0..isize) and checks the CRC; zs is (re)used across calls.
inline bool inflate_member(z_stream& zs, bool& zs_ready, const std::string& path, const uint8_t* zin, size_t cdata, uint32_t crc,
                           uint32_t isize, uint8_t* out, std::string& err) {
    if (!zs_ready) {
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) {
            err = "zlib inflateInit2 failed";
            return false;
        }
        zs_ready = true;
    } else {
        inflateReset(&zs);
    }
    zs.next_in = const_cast<uint8_t*>(zin);
    zs.avail_in = (uInt)cdata;
    zs.next_out = out;
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END || zs.avail_out != 0) {
        err = path + ": corrupt BGZF block (inflate failed)";
        return false;
    }
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out, isize) != crc) {
        err = path + ": corrupt BGZF block (CRC mismatch)";
        return false;
    }
    return true;
}

struct Reader {
    std::string path, err;
    FILE* fp = nullptr;
    std::vector<uint8_t> zin, buf;   // compressed member; inflated bytes, consumed up to `pos`
    size_t pos = 0;
    z_stream zs{};
    bool zs_ready = false;

    bool open(const std::string& p) {
        path = p;
        fp = std::fopen(p.c_str(), "rb");
        if (!fp) err = p + ": cannot open";
        return fp != nullptr;
    }
    ~Reader() {
        if (zs_ready) inflateEnd(&zs);
        if (fp) std::fclose(fp);
    }
    const uint8_t* data() const { return buf.data() + pos; }
    void advance(size_t n) { pos += n; }

    // appends the payload of the next member to buf: 1 = ok, 0 = clean end of file, -1 = error (err set)
    int next_block() {
        size_t cdata = 0;
        uint32_t crc = 0, isize = 0;
        const int rc = read_member(fp, path, zin, cdata, crc, isize, err);
        if (rc <= 0) return rc;
        if (isize == 0) return 1;   // empty member (the EOF marker)
        if (pos > 0 && pos == buf.size()) {                    // compact the consumed prefix before growing
            buf.clear();
            pos = 0;
        } else if (pos > (1u << 20)) {
            buf.erase(buf.begin(), buf.begin() + (long)pos);
            pos = 0;
        }
        const size_t old = buf.size();
        buf.resize(old + isize);
        return inflate_member(zs, zs_ready, path, zin.data(), cdata, crc, isize, buf.data() + old, err) ? 1 : -1;
    }

    // makes n inflated bytes available at data(): 1 = ok, 0 = clean EOF before the first byte, -1 = error
    int need(size_t n) {
        while (buf.size() - pos < n) {
            const bool empty = buf.size() == pos;
            const int rc = next_block();
            if (rc < 0) return -1;
            if (rc == 0) {
                if (empty) return 0;
                err = path + ": BAM stream ends inside a record";
                return -1;
            }
        }
        return 1;
    }
};

// The same stream interface with the inflate work spread over `threads` workers: BGZF members are independent deflate streams,
// so an I/O thread frames them into a ring of jobs, the workers inflate and CRC-check them concurrently and need() appends
// their payloads to buf in file order.  (One zlib inflate runs at 100-250 MB/s: a single decoder thread feeds about one GPU.)
struct ParallelReader {
    std::string path, err;
    std::vector<uint8_t> buf;
    size_t pos = 0;

    bool open(const std::string& p, int threads) {
        path = p;
        fp = std::fopen(p.c_str(), "rb");
        if (!fp) {
            err = p + ": cannot open";
            return false;
        }
        if (threads < 1) threads = 1;
        jobs.resize((size_t)threads * 4);
        io = std::thread([this] { io_loop(); });
        for (int i = 0; i < threads; ++i) workers.emplace_back([this] { work_loop(); });
        return true;
    }
    ~ParallelReader() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_io.notify_all();
        cv_work.notify_all();
        cv_cons.notify_all();
        if (io.joinable()) io.join();
        for (auto& w : workers)
            if (w.joinable()) w.join();
        if (fp) std::fclose(fp);
    }
    const uint8_t* data() const { return buf.data() + pos; }
    void advance(size_t n) { pos += n; }

    // makes n inflated bytes available at data(): 1 = ok, 0 = clean EOF before the first byte, -1 = error
    int need(size_t n) {
        while (buf.size() - pos < n) {
            Job& j = jobs[next_take % jobs.size()];
            int st;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_cons.wait(lk, [&] { return j.state == DONE || j.state == FAILED || j.state == END; });
                st = j.state;
            }
            if (st == END) {
                if (buf.size() == pos) return 0;
                err = path + ": BAM stream ends inside a record";
                return -1;
            }
            if (st == FAILED) {
                err = j.err;
                return -1;
            }
            if (pos > 0 && pos == buf.size()) {                // compact the consumed prefix before growing
                buf.clear();
                pos = 0;
            } else if (pos > (1u << 20)) {
                buf.erase(buf.begin(), buf.begin() + (long)pos);
                pos = 0;
            }
            buf.insert(buf.end(), j.out.begin(), j.out.begin() + j.isize);
            {
                std::lock_guard<std::mutex> lk(mu);
                j.state = EMPTY;
            }
            cv_io.notify_one();
            ++next_take;
        }
        return 1;
    }

  private:
    enum { EMPTY, QUEUED, DONE, FAILED, END };
    struct Job {
        std::vector<uint8_t> zin, out;
        size_t cdata = 0;
        uint32_t crc = 0, isize = 0;
        int state = EMPTY;
        std::string err;
    };
    FILE* fp = nullptr;
    std::vector<Job> jobs;
    std::mutex mu;
    std::condition_variable cv_io, cv_work, cv_cons;
    std::deque<size_t> work;
    uint64_t next_take = 0;
    bool stop = false;
    std::thread io;
    std::vector<std::thread> workers;

    void io_loop() {
        for (uint64_t seq = 0;; ++seq) {
            Job& j = jobs[seq % jobs.size()];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_io.wait(lk, [&] { return stop || j.state == EMPTY; });
                if (stop) return;
            }
            const int rc = read_member(fp, path, j.zin, j.cdata, j.crc, j.isize, j.err);   // the job is ours while EMPTY
            std::lock_guard<std::mutex> lk(mu);
            if (rc == 0) j.state = END;
            else if (rc < 0) j.state = FAILED;
            else if (j.isize == 0) j.state = DONE;             // empty member (the EOF marker): nothing to inflate
            else {
                j.state = QUEUED;
                work.push_back(seq % jobs.size());
                cv_work.notify_one();
                continue;
            }
            cv_cons.notify_all();
            if (rc <= 0) return;
        }
    }
    void work_loop() {
        z_stream zs;
        bool zs_ready = false;
        for (;;) {
            size_t idx;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !work.empty(); });
                if (stop) break;
                idx = work.front();
                work.pop_front();
            }
            Job& j = jobs[idx];
            if (j.out.size() < j.isize) j.out.resize(65536);
            const bool ok = inflate_member(zs, zs_ready, path, j.zin.data(), j.cdata, j.crc, j.isize, j.out.data(), j.err);
            {
                std::lock_guard<std::mutex> lk(mu);
                j.state = ok ? DONE : FAILED;
            }
            cv_cons.notify_all();
        }
        if (zs_ready) inflateEnd(&zs);
    }
};

// One complete BGZF member (header, deflate data at level 6, CRC32, ISIZE) for `n` <= 0xff00 payload bytes; returns its
// length in zout or 0 on error.
inline size_t deflate_member(const uint8_t* payload, size_t n, std::vector<uint8_t>& zout, const std::string& path,
                             std::string& err, int level = 6) {
    z_stream z;
    std::memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
        err = "zlib deflateInit2 failed";
        return 0;
    }
    zout.resize(deflateBound(&z, (uLong)n) + 64);
    z.next_in = const_cast<uint8_t*>(payload);
    z.avail_in = (uInt)n;
    z.next_out = zout.data() + 18;
    z.avail_out = (uInt)(zout.size() - 18);
    const int rc = deflate(&z, Z_FINISH);
    const size_t clen = z.total_out;
    deflateEnd(&z);
    if (rc != Z_STREAM_END || clen + 26 > 65536) {
        err = path + ": BGZF block does not fit 64 KiB after deflate";
        return 0;
    }
    const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0,
                              (uint8_t)((clen + 25) & 255), (uint8_t)((clen + 25) >> 8)};
    std::memcpy(zout.data(), head, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), payload, (uInt)n), isz = (uint32_t)n;
    uint8_t* t = zout.data() + 18 + clen;
    for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i)), t[4 + i] = (uint8_t)(isz >> (8 * i));
    return clen + 26;
}

struct Writer {
    std::string path, err;
    FILE* fp = nullptr;
    std::vector<uint8_t> pend, zout;
    uint64_t file_off = 0;          // file offset at which the member holding `pend` will start
    static constexpr size_t BLOCK = 0xff00;

    bool open(const std::string& p) {
        path = p;
        fp = std::fopen(p.c_str(), "wb");
        if (!fp) err = p + ": cannot create";
        pend.reserve(BLOCK);
        return fp != nullptr;
    }
    ~Writer() {
        if (fp) std::fclose(fp);
    }
    uint64_t voffset() const { return (file_off << 16) | (uint64_t)pend.size(); }   // of the next byte written

    bool flush_block() {
        const size_t mlen = deflate_member(pend.data(), pend.size(), zout, path, err);
        if (!mlen) return false;
        if (std::fwrite(zout.data(), 1, mlen, fp) != mlen) {
            err = path + ": write failed";
            return false;
        }
        file_off += mlen;
        pend.clear();
        return true;
    }
    bool write(const uint8_t* p, size_t n) {
        while (n) {
            const size_t room = BLOCK - pend.size(), k = n < room ? n : room;
            pend.insert(pend.end(), p, p + k);
            p += k;
            n -= k;
            if (pend.size() == BLOCK && !flush_block()) return false;
        }
        return true;
    }
    // start a new member so that the next write begins a block (records then never straddle it unless they exceed 64 KiB)
    bool align_block(size_t upcoming) {
        if (!pend.empty() && pend.size() + upcoming > BLOCK) return flush_block();
        return true;
    }
    bool finish() {                                            // last data block + the empty EOF member
        if (!pend.empty() && !flush_block()) return false;
        if (!flush_block()) return false;
        const bool ok = std::fclose(fp) == 0;
        fp = nullptr;
        if (!ok) err = path + ": close failed";
        return ok;
    }
};

// Writer with the deflate work spread over `threads` workers (the members are independent): write() fills a block, full blocks
// go to a ring of jobs, workers compress them, an output thread writes them in order.  Compressed sizes are not known when a
// record is written, so voffset() is LOGICAL -- (sequence number of the block << 16) | offset inside it -- and resolve() turns
// it into the real virtual file offset once finish() has returned.  (One zlib deflate at level 6 runs at ~25 MB/s.)
struct ParallelWriter {
    std::string path, err;
    int level = 6;                  // zlib level of the members (set before open; 1 for temporary sort runs)
    static constexpr size_t BLOCK = 0xff00;

    bool open(const std::string& p, int threads) {
        path = p;
        fp = std::fopen(p.c_str(), "wb");
        if (!fp) {
            err = p + ": cannot create";
            return false;
        }
        if (threads < 1) threads = 1;
        jobs.resize((size_t)threads * 4);
        pend.reserve(BLOCK);
        out = std::thread([this] { out_loop(); });
        for (int i = 0; i < threads; ++i) workers.emplace_back([this] { work_loop(); });
        return true;
    }
    ~ParallelWriter() { shutdown(); }

    uint64_t voffset() const { return (seq << 16) | (uint64_t)pend.size(); }   // logical, of the next byte written
    uint64_t resolve(uint64_t v) const { return (block_off[(size_t)(v >> 16)] << 16) | (v & 0xffff); }   // after finish()

    bool flush_block() { return submit(); }
    bool write(const uint8_t* p, size_t n) {
        while (n) {
            const size_t room = BLOCK - pend.size(), k = n < room ? n : room;
            pend.insert(pend.end(), p, p + k);
            p += k;
            n -= k;
            if (pend.size() == BLOCK && !submit()) return false;
        }
        return true;
    }
    bool align_block(size_t upcoming) {
        if (!pend.empty() && pend.size() + upcoming > BLOCK) return submit();
        return true;
    }
    bool finish() {                                            // last data block + the empty EOF member, then drain
        if (!pend.empty() && !submit()) return false;
        if (!submit()) return false;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_main.wait(lk, [&] { return failed || written == seq; });
        }
        const bool bad = failed;
        shutdown();
        if (bad) return false;
        block_off.push_back(file_off);                         // where a further block would start (voffsets taken at the end)
        const bool ok = std::fclose(fp) == 0;
        fp = nullptr;
        if (!ok) err = path + ": close failed";
        return ok;
    }

  private:
    enum { EMPTY, QUEUED, DONE };
    struct Job {
        std::vector<uint8_t> in, zout;
        size_t mlen = 0;
        int state = EMPTY;
    };
    FILE* fp = nullptr;
    std::vector<uint8_t> pend;
    std::vector<Job> jobs;
    std::vector<uint64_t> block_off;       // file offset of block i, filled by the output thread in order
    uint64_t seq = 0, written = 0, file_off = 0;
    std::mutex mu;
    std::condition_variable cv_main, cv_work, cv_out;
    std::deque<size_t> work;
    bool stop = false, failed = false;
    std::thread out;
    std::vector<std::thread> workers;

    bool submit() {                        // hands `pend` (possibly empty: the EOF member) to the workers as block `seq`
        Job& j = jobs[seq % jobs.size()];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_main.wait(lk, [&] { return failed || j.state == EMPTY; });
            if (failed) return false;
        }
        j.in.swap(pend);
        pend.clear();
        pend.reserve(BLOCK);
        {
            std::lock_guard<std::mutex> lk(mu);
            j.state = QUEUED;
            work.push_back(seq % jobs.size());
        }
        cv_work.notify_one();
        ++seq;
        return true;
    }
    void work_loop() {
        for (;;) {
            size_t idx;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !work.empty(); });
                if (stop) return;
                idx = work.front();
                work.pop_front();
            }
            Job& j = jobs[idx];
            std::string e;
            j.mlen = deflate_member(j.in.data(), j.in.size(), j.zout, path, e, level);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!j.mlen) {
                    failed = true;
                    err = e;
                }
                j.state = DONE;
            }
            cv_out.notify_one();
            cv_main.notify_all();
        }
    }
    void out_loop() {
        for (;;) {
            Job& j = jobs[written % jobs.size()];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_out.wait(lk, [&] { return stop || failed || j.state == DONE; });
                if (stop || failed) return;
            }
            const bool ok = std::fwrite(j.zout.data(), 1, j.mlen, fp) == j.mlen;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!ok) {
                    failed = true;
                    err = path + ": write failed";
                } else {
                    block_off.push_back(file_off);
                    file_off += j.mlen;
                    ++written;
                }
                j.state = EMPTY;
            }
            cv_main.notify_all();
            if (!ok) return;
        }
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        cv_out.notify_all();
        cv_main.notify_all();
        if (out.joinable()) out.join();
        for (auto& w : workers)
            if (w.joinable()) w.join();
        workers.clear();
    }
};

inline int32_t le32(const uint8_t* p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace clmbgzf
