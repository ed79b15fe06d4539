// tools/asan_pattern_harness.cpp -- the pattern compiler and the host automata under
// AddressSanitizer + UBSan: every line of the input file is compiled (grammar strings, mutated
// grammar strings and plain garbage from tests/fuzz_patterns.py's generator); what compiles is
// run forwards and backwards over random data. No GPU, no library: only hs_pattern.cpp.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tools/asan_pattern_harness.cpp \
//       hyperscan_amd/csrc/hs_pattern.cpp -o /tmp/asan_harness && /tmp/asan_harness patterns.txt
// Round 1: 55 000 lines (26 112 compiled, 28 888 refused), then 45 000 more with HS_FLAG_UTF8 on a
// third of them and non-ASCII characters spliced in: no report.
#include "../include/hs_gpu.h"
#include "../hyperscan_amd/csrc/hs_pattern.h"
#include <cstdio>
#include <fstream>
#include <iostream>
#include <random>
using namespace hsf;
int main(int argc, char **argv) {
    std::ifstream in(argv[1], std::ios::binary);
    std::string line;
    std::mt19937 rng(1);
    size_t ok = 0, bad = 0, n = 0;
    std::string data(300, 'a');
    const char alpha[] = "abcXY01 _-\nfoobar";
    while (std::getline(in, line)) {
        n++;
        unsigned flags = (unsigned)(rng() % 8) | ((rng() % 4 == 0) ? HS_FLAG_SOM_LEFTMOST : 0) | ((rng() % 3 == 0) ? HS_FLAG_UTF8 : 0);
        try {
            std::vector<Pattern> bs = parse_pattern(line, flags, 1);
            for (Pattern &p : bs) finish_pattern(p);
            ok++;
            for (auto &c : data) c = alpha[rng() % (sizeof(alpha) - 1)];
            const unsigned char *buf = (const unsigned char *)data.data();
            for (Pattern &p : bs) {
                for (size_t pos = 0; pos <= data.size(); pos += 7) {
                    size_t cnt = 0;
                    if (p.general) TailNfa::run_general(p.g, buf, data.size(), pos, [&](size_t) { return ++cnt < 50; });
                    else if (!p.tail.empty()) TailNfa::run64(p, buf, data.size(), pos, [&](size_t) { return ++cnt < 50; });
                    size_t from = 0;
                    if (p.has_pre) TailNfa::run_reverse(p, buf, data.size(), pos, (rng() & 1) != 0, from);
                }
                unsigned long long lo, hi; bool inf;
                raw_widths(p, lo, hi, inf);
            }
        } catch (const ParseError &) {
            bad++;
        }
    }
    printf("%zu lines: %zu compiled, %zu refused\n", n, ok, bad);
    return 0;
}
