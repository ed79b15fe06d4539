// tools/asan_table_harness.cpp -- the literal-table compiler under AddressSanitizer + UBSan: random literal sets of
// 1 .. 12 000 literals (lengths 1-8, caseless, masks, noruns, duplicate ids, every build flag combination incl. the
// pair filter and the key gate) are compiled, validated, serialised, deserialised (intact, with bytes damaged,
// truncated). The device runtime is replaced by tools/asan_stubs.cpp.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tools/asan_table_harness.cpp tools/asan_stubs.cpp \
//       hyperscan_amd/csrc/compile.cpp -lpthread -o /tmp/th && /tmp/th 400
#include "../include/hsgpu.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937 rng(11);
    const unsigned flag_pool[] = {0, 1, 2, 2 | 4, 1 | 4, 16, 16 | 2 | 8, 32, 64, 64 | 2 | 32 | 4, 512, 512 | 64, 128, 256, 1024, 1024 | 32, 2048, 4096, 4096 | 2};
    size_t built = 0, refused = 0, loaded = 0;
    for (int r = 0; r < rounds; r++) {
        const size_t n = r % 17 == 0 ? 3000 + rng() % 9000 : 1 + rng() % (r % 5 == 0 ? 1500 : 120);
        std::vector<std::string> bytes(n), msk(n), cmp(n);
        std::vector<hsgpu_lit_t> lits(n);
        const bool texty = rng() & 1;
        const bool sloppy = rng() % 10 == 0; /* one set in ten carries masks that contradict their literals: refused */
        for (size_t i = 0; i < n; i++) {
            const unsigned len = (rng() % 16 == 0) ? rng() % 3 + 1 : rng() % 8 + 1;
            for (unsigned j = 0; j < len; j++) bytes[i].push_back(texty ? "abcdefgHIJ01 /-"[rng() % 15] : (char)(rng() & 0xff));
            memset(&lits[i], 0, sizeof(lits[i]));
            lits[i].s = (const uint8_t *)bytes[i].data();
            lits[i].len = len;
            lits[i].id = rng() % 7 == 0 ? (uint32_t)(rng() % 5) : (uint32_t)i;
            lits[i].nocase = rng() % 3 == 0;
            lits[i].noruns = rng() % 11 == 0;
            lits[i].groups = rng() % 9 == 0 ? (1ull << (rng() % 64)) : ~0ull;
            if (rng() % 13 == 0) { /* a mask over the last bytes, consistent with the literal more often than not */
                const unsigned ml = rng() % 8 + 1;
                for (unsigned j = 0; j < ml; j++) {
                    const unsigned char m = rng() % 4 ? 0xff : (unsigned char)rng();
                    unsigned char c = j < len && (!sloppy || rng() % 5) ? (unsigned char)bytes[i][len - 1 - j] : (unsigned char)rng();
                    if (lits[i].nocase && ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) c &= 0xdf;
                    msk[i].insert(msk[i].begin(), (char)m);
                    cmp[i].insert(cmp[i].begin(), (char)(c & m));
                }
                lits[i].msk = (const uint8_t *)msk[i].data();
                lits[i].cmp = (const uint8_t *)cmp[i].data();
                lits[i].msk_len = ml;
            }
        }
        const unsigned flags = flag_pool[rng() % (sizeof(flag_pool) / sizeof(flag_pool[0]))];
        const char *why = nullptr;
        hsgpu_hwlm_t *t = nullptr;
        if (hsgpu_hwlm_build(lits.data(), n, flags, &t) != HSGPU_SUCCESS) {
            refused++;
            why = hsgpu_last_error();
            if (r < 12) printf("  set %d (n=%zu, flags %u) refused: %s\n", r, n, flags, why);
            continue;
        }
        built++;
        hsgpu_hwlm_info_t info;
        hsgpu_hwlm_get_info(t, &info);
        size_t len = 0;
        hsgpu_hwlm_serialize(t, nullptr, 0, &len);
        std::vector<unsigned char> blob(len);
        hsgpu_hwlm_serialize(t, blob.data(), blob.size(), &len);
        hsgpu_hwlm_t *u = nullptr;
        if (hsgpu_hwlm_deserialize(blob.data(), len, &u) == HSGPU_SUCCESS) loaded++, hsgpu_hwlm_free(u);
        else printf("  set %d (n=%zu, flags %u -> table flags %u): intact image refused: %s\n", r, n, flags, info.flags, hsgpu_last_error());
        for (int k = 0; k < 6; k++) { /* damaged and truncated images must be refused or load cleanly, never crash */
            std::vector<unsigned char> bad(blob);
            if (k < 4) bad[rng() % bad.size()] ^= (unsigned char)(1u << (rng() % 8));
            else bad.resize(rng() % bad.size());
            u = nullptr;
            if (hsgpu_hwlm_deserialize(bad.data(), bad.size(), &u) == HSGPU_SUCCESS) hsgpu_hwlm_free(u);
        }
        hsgpu_hwlm_free(t);
    }
    printf("%d sets: %zu tables built, %zu refused, %zu reloaded\n", rounds, built, refused, loaded);
    return 0;
}
