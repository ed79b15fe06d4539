/*
 * simplegrep -- the reference's examples/simplegrep.c use case against include/hs_gpu.h:
 * compile ONE pattern for block mode, read a whole file, scan it with hs_scan and print
 * "Match for pattern "<p>" at offset <to>" per match (examples/simplegrep.c:77-81,147-220).
 * Written from scratch against the same public API; the scan runs on the GPU.
 *
 *   usage: simplegrep <pattern> <input file>
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hs_gpu.h"

static int on_match(unsigned int id, unsigned long long from, unsigned long long to, unsigned int flags, void *ctx) {
    (void)id;
    (void)from;
    (void)flags;
    printf("Match for pattern \"%s\" at offset %llu\n", (const char *)ctx, to);
    return 0;
}

static char *read_file(const char *path, unsigned int *len) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "ERROR: unable to open file \"%s\": %s\n", path, strerror(errno));
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0 || (unsigned long)n > 0xffffffffUL) {
        fprintf(stderr, "ERROR: file too large for one block (hs_scan length is 32 bits)\n");
        fclose(f);
        return NULL;
    }
    char *buf = malloc(n ? (size_t)n : 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "ERROR: unable to read file\n");
        fclose(f);
        free(buf);
        return NULL;
    }
    fclose(f);
    *len = (unsigned int)n;
    return buf;
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "Usage: %s <pattern> <input file>\n", argv[0]);
        return -1;
    }
    const char *pattern = argv[1];
    hs_database_t *db = NULL;
    hs_compile_error_t *err = NULL;
    if (hs_compile(pattern, HS_FLAG_DOTALL, HS_MODE_BLOCK, NULL, &db, &err) != HS_SUCCESS) {
        fprintf(stderr, "ERROR: Unable to compile pattern \"%s\": %s\n", pattern, err ? err->message : "?");
        hs_free_compile_error(err);
        return -1;
    }
    unsigned int len = 0;
    char *data = read_file(argv[2], &len);
    if (!data) {
        hs_free_database(db);
        return -1;
    }
    hs_scratch_t *scratch = NULL;
    if (hs_alloc_scratch(db, &scratch) != HS_SUCCESS) {
        fprintf(stderr, "ERROR: Unable to allocate scratch space. Exiting.\n");
        free(data);
        hs_free_database(db);
        return -1;
    }
    printf("Scanning %u bytes with Hyperscan\n", len);
    if (hs_scan(db, data, len, 0, scratch, on_match, (void *)pattern) != HS_SUCCESS) {
        fprintf(stderr, "ERROR: Unable to scan input buffer. Exiting.\n");
        hs_free_scratch(scratch);
        free(data);
        hs_free_database(db);
        return -1;
    }
    hs_free_scratch(scratch);
    free(data);
    hs_free_database(db);
    return 0;
}
