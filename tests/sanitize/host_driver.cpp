// Driver for the sanitizer builds of the host-side C++ (tests/test_sanitizers.py): the BAM feeder (BGZF inflate workers, decoder
// thread, slot ring), the filter (BAM and SAM text input) and the coordinate sort with spilled runs + BAI, through the C ABI of
// include/chimeralm_feed.h.  Built from csrc/bam_feeder.cpp + csrc/bam_filter.cpp with -fsanitize=address,undefined and, again,
// with -fsanitize=thread; any report makes the run fail (halt_on_error).
//   host_driver <reads.bam> <reads.sam> <workdir>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "chimeralm_feed.h"
#include "chimeralm_hip.h"

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, clm_bam_last_error()); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string bam = argv[1], sam = argv[2], dir = argv[3];
    // ---- feeder: several inflate threads, small ring, batches released out of order of arrival
    std::vector<std::string> names;
    for (int threads : {1, 3}) {
        clm_feeder_config cfg;
        CHECK(clm_feeder_default_config(&cfg) == 0);
        cfg.batch_size = 7;
        cfg.slots = 2;
        cfg.pinned = 0;
        cfg.inflate_threads = threads;
        cfg.max_tokens = 5000;
        clm_feeder* f = nullptr;
        if (clm_feeder_open(bam.c_str(), &cfg, &f) != 0) {
            std::fprintf(stderr, "feeder open: %s\n", clm_feeder_last_error(nullptr));
            return 1;
        }
        clm_feed_batch b;
        long reads = 0, sum = 0;
        int rc;
        while ((rc = clm_feeder_next(f, &b)) == 1) {
            const unsigned char* ids = static_cast<const unsigned char*>(b.ids);
            for (int r = 0; r < b.n_reads; ++r)
                for (int t = 0; t < b.n_tokens; ++t) sum += ids[(size_t)r * b.row_stride + t];   // touch every byte of the slot
            if (threads == 1) {
                const signed char* nm = static_cast<const signed char*>(b.names);
                for (int r = 0; r < b.n_reads; ++r) names.emplace_back(reinterpret_cast<const char*>(nm + (size_t)r * 256 + 1), (size_t)(unsigned char)nm[(size_t)r * 256]);
            }
            reads += b.n_reads;
            CHECK(clm_feeder_release(f, b.slot) == 0);
        }
        CHECK(rc == 0);
        CHECK(reads == 100 && sum > 0);
        CHECK(clm_feeder_close(f) == 0);
    }
    // ---- filter: BAM in, SAM in, every second read dropped
    std::vector<const char*> drop;
    for (size_t i = 0; i < names.size(); i += 2) drop.push_back(names[i].c_str());
    int64_t kept = 0, dropped = 0, unplaced = 0, kept2 = 0, dropped2 = 0;
    const std::string f1 = dir + "/a.filtered.bam", f2 = dir + "/b.filtered.bam";
    CHECK(clm_bam_filter_ex(bam.c_str(), f1.c_str(), drop.data(), (int64_t)drop.size(), 0, &kept, &dropped, &unplaced) == 0);
    CHECK(clm_bam_filter_ex(sam.c_str(), f2.c_str(), drop.data(), (int64_t)drop.size(), CLM_BAM_INPUT_SAM, &kept2, &dropped2, &unplaced) == 0);
    CHECK(kept == kept2 && dropped == dropped2 && dropped > 0);
    // ---- sort + index: in memory, then with a 1 MiB budget (spilled runs, k-way merge)
    int64_t n1 = 0, n2 = 0;
    const std::string s1 = dir + "/a.sorted.bam", s2 = dir + "/b.sorted.bam";
    setenv("CLM_SORT_MEM_MB", "4096", 1);
    CHECK(clm_bam_sort_index(f1.c_str(), s1.c_str(), nullptr, &n1) == 0);
    setenv("CLM_SORT_MEM_MB", "1", 1);
    CHECK(clm_bam_sort_index(bam.c_str(), s2.c_str(), nullptr, &n2) == 0);
    CHECK(n1 == kept && n2 >= n1);
    // ---- error paths must not leak or race either
    CHECK(clm_bam_sort_index((dir + "/missing.bam").c_str(), s2.c_str(), nullptr, &n2) != 0);
    CHECK(clm_bam_filter_ex(f1.c_str(), (dir + "/c.bam").c_str(), nullptr, 0, CLM_BAM_INPUT_SAM, &kept, &dropped, &unplaced) != 0);   // a BAM read as SAM text
    std::printf("sanitizer driver OK: %ld names, kept %lld, sorted %lld / %lld\n", (long)names.size(), (long long)kept2, (long long)n1, (long long)n2);
    return 0;
}
