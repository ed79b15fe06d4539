// tools/asan_facade_harness.cpp -- the hs_* facade's host half under AddressSanitizer + UBSan:
// every input line (and every group of four) is compiled, serialised, deserialised (intact, with
// one byte damaged, truncated), confirmed over random blocks with naive literal hits, and measured
// by hs_expression_info. The device runtime is replaced by tools/asan_stubs.cpp.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tools/asan_facade_harness.cpp \
//       tools/asan_stubs.cpp hyperscan_amd/csrc/{hs_pattern,hs_facade,compile}.cpp -lpthread -o /tmp/fh
// Round 1: 55 000 lines (27 042 databases compiled), no report.
#include "../include/hs_gpu.h"
#include "../include/hsgpu.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
static int on_event(unsigned long long, unsigned, unsigned long long, unsigned long long, unsigned, void *ctx) {
    return ++*(unsigned long long *)ctx > 5000; /* also exercises per-block termination */
}
int main(int argc, char **argv) {
    std::ifstream in(argv[1], std::ios::binary);
    std::string line;
    std::mt19937 rng(7);
    const char alpha[] = "abcABCXYxy01 _-\nfoobarcabXY";
    size_t n = 0, ok = 0, events = 0;
    std::vector<std::string> batch;
    auto run = [&](const std::vector<std::string> &exprs) {
        std::vector<const char *> ptr;
        std::vector<unsigned> flags, ids;
        for (size_t i = 0; i < exprs.size(); i++) {
            ptr.push_back(exprs[i].c_str());
            unsigned f = (unsigned)(rng() % 8);
            if (rng() % 5 == 0) f |= HS_FLAG_SOM_LEFTMOST;
            else if (rng() % 5 == 0) f |= HS_FLAG_SINGLEMATCH;
            flags.push_back(f);
            ids.push_back((unsigned)(i % 3));
        }
        hs_database_t *db = nullptr;
        hs_compile_error_t *err = nullptr;
        if (hs_compile_multi(ptr.data(), flags.data(), ids.data(), (unsigned)ptr.size(), HS_MODE_BLOCK, nullptr, &db, &err) != HS_SUCCESS) {
            hs_free_compile_error(err);
            return;
        }
        ok++;
        char *blob = nullptr;
        size_t blen = 0;
        if (hs_serialize_database(db, &blob, &blen) == HS_SUCCESS) {
            hs_database_t *db2 = nullptr;
            if (hs_deserialize_database(blob, blen, &db2) == HS_SUCCESS) hs_free_database(db2);
            if (blen > 16) { /* damaged copies must be refused, not crash */
                std::string bad(blob, blen);
                bad[rng() % blen] ^= (char)(1 + rng() % 255);
                hs_database_t *db3 = nullptr;
                if (hs_deserialize_database(bad.data(), bad.size(), &db3) == HS_SUCCESS) hs_free_database(db3);
                size_t cut = rng() % blen;
                if (hs_deserialize_database(blob, cut, &db3) == HS_SUCCESS) hs_free_database(db3);
            }
            free(blob);
        }
        /* blocks + naive literal hits */
        std::string data;
        std::vector<unsigned long long> off{0};
        for (int b = 0; b < 6; b++) {
            size_t len = rng() % 120;
            for (size_t k = 0; k < len; k++) data.push_back(alpha[rng() % (sizeof(alpha) - 1)]);
            off.push_back(data.size());
        }
        if (data.empty()) data.push_back('x');
        std::vector<hsgpu_match_t> recs;
        for (unsigned li = 0;; li++) {
            const char *bytes; size_t len; int nocase; unsigned id;
            if (hs_database_literal(db, li, &bytes, &len, &nocase, &id) != HS_SUCCESS) break;
            for (size_t b = 0; b + 1 < off.size(); b++)
                for (size_t e = off[b] + len; e <= off[b + 1]; e++) {
                    bool eq = true;
                    for (size_t k = 0; k < len && eq; k++) {
                        unsigned char x = data[e - len + k], y = bytes[k];
                        if (nocase) { if (x >= 'a' && x <= 'z') x -= 32; if (y >= 'a' && y <= 'z') y -= 32; }
                        eq = x == y;
                    }
                    if (eq) recs.push_back(hsgpu_match_t{(uint32_t)b, (uint32_t)(e - 1 - off[b]), li, li});
                }
        }
        std::sort(recs.begin(), recs.end(), [](const hsgpu_match_t &a, const hsgpu_match_t &b) {
            return a.block != b.block ? a.block < b.block : a.end != b.end ? a.end < b.end : a.lit < b.lit; });
        unsigned long long cnt = 0;
        hs_confirm_batch(db, data.data(), off.data(), off.size() - 1, recs.data(), recs.size(), on_event, &cnt);
        events += cnt;
        hs_expr_info_t *info = nullptr;
        if (hs_expression_info(exprs[0].c_str(), flags[0], &info, &err) == HS_SUCCESS) free(info);
        else hs_free_compile_error(err);
        hs_free_database(db);
    };
    while (std::getline(in, line)) {
        n++;
        run({line});
        batch.push_back(line);
        if (batch.size() == 4) { run(batch); batch.clear(); }
    }
    printf("%zu lines: %zu databases compiled, %zu events\n", n, ok, events);
    return 0;
}
