/* internal.h -- host-side objects behind the opaque C-ABI handles. */
#ifndef HSGPU_INTERNAL_H
#define HSGPU_INTERNAL_H

#include "../../include/hsgpu.h"
#include "table.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>

struct hsgpu_hwlm {
    std::vector<uint8_t> blob; /* HsgpuTableHeader + sections */
    /* lazily uploaded, one copy per device; freed with the table */
    std::mutex mu;
    std::map<int, void *> dev_blob;

    const HsgpuTableHeader *hdr() const { return (const HsgpuTableHeader *)blob.data(); }
    const HsgpuDevLit *lits() const { return (const HsgpuDevLit *)(blob.data() + hdr()->off_lits); }
};

void hsgpu_set_error(const char *fmt, ...);

/* compile.cpp */
int hsgpu_compile_table(const hsgpu_lit_t *lits, size_t n, unsigned flags, std::vector<uint8_t> &blob);
int hsgpu_validate_blob(const void *buf, size_t len);
uint32_t hsgpu_blob_checksum(const uint8_t *blob, size_t len);
int hsgpu_table_agrees(const hsgpu_hwlm *t, const hsgpu_lit_t *lits, size_t n);

/* devmem.hip: every device allocation of the library (hipMalloc, or guard-page ranges in the out-of-bounds tests) */
int hsgpu_dev_alloc(void **p, size_t bytes);
void hsgpu_dev_free(void *p);
int hsgpu_dev_guard_mode();

/* runtime.hip */
void hsgpu_release_device_copies(hsgpu_hwlm *t);
int hsgpu_host_is_pinned(const void *p);

#endif
