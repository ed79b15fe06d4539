// stand-ins for the device runtime (runtime.hip, class_scan.hip): the host-only library under ASan
#include "../include/hsgpu.h"
#include "../hyperscan_amd/csrc/internal.h"
extern "C" {
int hsgpu_scratch_alloc(hsgpu_scratch_t **s, int) { if (s) *s = nullptr; hsgpu_set_error("no device (ASan host build)"); return HSGPU_UNKNOWN_ERROR; }
void hsgpu_scratch_free(hsgpu_scratch_t *) {}
int hsgpu_hwlm_exec_batch(const hsgpu_hwlm_t *, hsgpu_scratch_t *, const uint8_t *, const uint64_t *, size_t, size_t, hsgpu_match_t *, size_t, size_t *) { return HSGPU_UNKNOWN_ERROR; }
}
void hsgpu_release_device_copies(hsgpu_hwlm *) {}
